// Measurement helper for bench.py (not part of the product library): the shader clock the GPU is
// actually running at, sampled on the caller's stream right behind a solve launch.
//   s_memtime     counts shader-clock cycles (profiles/ubench/mfma_valu_overlap.result.txt: 32.01 ticks
//                 per v_mfma_f32_16x16x4_f32 = its 32-cycle issue interval)
//   s_memrealtime counts the constant reference clock (hipDeviceAttributeWallClockRate, 100 MHz on gfx950)
// One wave spins for `ref_ticks` reference ticks and reports both deltas; MHz = d_shader / d_ref * ref_MHz.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC clockprobe.hip -o libclockprobe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(64) clockprobe_kernel(int64_t ref_ticks, int64_t* out) {
    const int64_t r0 = (int64_t)__builtin_amdgcn_s_memrealtime();
    const int64_t c0 = (int64_t)__builtin_amdgcn_s_memtime();
    int64_t r1 = r0;
    while (r1 - r0 < ref_ticks) {
        __builtin_amdgcn_s_sleep(8);
        r1 = (int64_t)__builtin_amdgcn_s_memrealtime();
    }
    const int64_t c1 = (int64_t)__builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

extern "C" {
// Enqueue one probe on `stream`; out2 = device pointer to two int64 {shader cycles, reference ticks}.
int clockprobe_launch(int64_t ref_ticks, int64_t* out2, void* stream) {
    hipLaunchKernelGGL(clockprobe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ref_ticks, out2);
    return (int)hipGetLastError();
}
// Reference clock in kHz (hipDeviceAttributeWallClockRate); <= 0 if unknown.
int clockprobe_ref_khz(int device) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess) return -1;
    return khz;
}
}
