// Microbenchmark 4: issue cost of packed-f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against
// plain v_fma_f32, alone and interleaved with v_mfma_f32_16x16x4_f32 (one wave per SIMD).
//   hipcc -O3 --offload-arch=gfx950 pk_f32_cost.hip -o pk_f32_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: NV plain v_fma_f32; 1: NV v_pk_fma_f32; 2: NV v_pk_mul_f32; 3: NV v_pk_add_f32
template <int KIND, int NV, int NM>
__global__ void __launch_bounds__(256) k(int iters, long long* cyc, float* sink) {
    f32x4 acc[4] = {};
    float v[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    f32x2 w[8];
    for (int i = 0; i < 8; ++i) w[i] = f32x2{0.1f * i, 0.2f * i};
    const float a = 1e-3f * threadIdx.x, b = 0.5f;
    const f32x2 a2 = {a, a + 1.f}, b2 = {b, b * 0.5f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(w[i & 7]) : "v"(a2), "v"(b2));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(w[i & 7]) : "v"(a2));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(w[i & 7]) : "v"(a2));
        }
    }
    asm volatile("s_nop 7\n s_nop 7");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0]; for (int i = 0; i < 8; ++i) s += v[i] + w[i][0] + w[i][1];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NV, int NM>
void run(const char* label) {
    const int iters = 1000, nblk = 256;
    long long* cyc; float* sink;
    (void)hipMalloc(&cyc, sizeof(long long) * nblk * 4); (void)hipMalloc(&sink, sizeof(float) * nblk * 256);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, NV, NM>), dim3(nblk), dim3(256), 0, 0, iters, cyc, sink);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(nblk * 4);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * nblk * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s NM=%2d NV=%2d : %8.1f ticks/iter\n", label, NM, NV, (double)h[h.size() / 2] / iters);
    (void)hipFree(cyc); (void)hipFree(sink);
}

int main() {
    run<0, 64, 0>("64 v_fma_f32 alone");
    run<1, 64, 0>("64 v_pk_fma_f32 alone (= 128 fma)");
    run<2, 64, 0>("64 v_pk_mul_f32 alone");
    run<3, 64, 0>("64 v_pk_add_f32 alone");
    run<0, 0, 16>("16 MFMA alone");
    run<0, 64, 16>("16 MFMA then 64 v_fma_f32");
    run<1, 32, 16>("16 MFMA then 32 v_pk_fma_f32 (same flops)");
    run<2, 32, 16>("16 MFMA then 32 v_pk_mul_f32");
    run<1, 64, 16>("16 MFMA then 64 v_pk_fma_f32");
    return 0;
}
