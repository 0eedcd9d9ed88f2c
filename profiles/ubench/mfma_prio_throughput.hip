// Microbenchmark 3: does an ASYMMETRIC wave priority let co-resident waves hide each other's VALU phases
// behind v_mfma_f32_16x16x4_f32?  Phase-structured waves as in mfma_phase_overlap.hip ([NM MFMAs] then
// [NV v_fma + NE v_exp]); every wave runs until a deadline and counts its iterations, so the figure is
// SIMD throughput (what a kernel with a dynamic tile queue would see), not the slowest wave.
//   PRIO 0: none   1: waves 0-3 (one per SIMD) at s_setprio 3 for the whole run
//   2: every wave raises its priority for its VALU phase   3: ... for its MFMA phase
//   hipcc -O3 --offload-arch=gfx950 mfma_prio_throughput.hip -o mfma_prio_throughput
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NE, int PRIO>
__global__ void __launch_bounds__(1024) k(long long deadline, int* iters_out, float* sink) {
    f32x4 acc[4] = {};
    float v[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    const float a = 1e-3f * threadIdx.x, b = 0.5f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (PRIO == 1 && wave < 4) asm volatile("s_setprio 3");
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    int it = 0;
    // stagger the start of the second wave of each SIMD by half an iteration (PRIO 4)
    for (;; ++it) {
        if (PRIO == 3) asm volatile("s_setprio 3");
#pragma unroll
        for (int i = 0; i < NM; ++i)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
        if (PRIO == 3) asm volatile("s_setprio 0");
        if (PRIO == 2) asm volatile("s_setprio 3");
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
        if (PRIO == 2) asm volatile("s_setprio 0");
        if (__builtin_amdgcn_s_memtime() - t0 > deadline) break;
    }
    if ((threadIdx.x & 63) == 0) iters_out[blockIdx.x * 16 + wave] = it + 1;
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0]; for (int i = 0; i < 8; ++i) s += v[i];
    sink[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int NM, int NV, int NE, int PRIO>
void run(int nthreads, const char* label) {
    const int nblk = 256;
    const long long deadline = 3000000;
    int* it; float* sink;
    (void)hipMalloc(&it, sizeof(int) * nblk * 16); (void)hipMalloc(&sink, sizeof(float) * nblk * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NM, NV, NE, PRIO>), dim3(nblk), dim3(nthreads), 0, 0, deadline, it, sink);
    (void)hipDeviceSynchronize();
    const int wpb = nthreads / 64;
    std::vector<int> h(nblk * 16);
    (void)hipMemcpy(h.data(), it, sizeof(int) * nblk * 16, hipMemcpyDeviceToHost);
    double tot = 0, hi = 0, lo = 0;
    for (int bI = 0; bI < nblk; ++bI) for (int w = 0; w < wpb; ++w) { tot += h[bI * 16 + w]; (w < 4 ? hi : lo) += h[bI * 16 + w]; }
    const double per_simd = tot / (nblk * 4.0);
    printf("%-40s waves/SIMD=%d NM=%d NV=%d NE=%d prio=%d: %.1f iters/SIMD (waves0-3 %.1f, others %.1f)  MFMA-busy %.1f%%\n", label,
           wpb / 4, NM, NV, NE, PRIO, per_simd, hi / (nblk * 4.0), lo / (nblk * 4.0), 100.0 * per_simd * NM * 32.0 / deadline);
    (void)hipFree(it); (void)hipFree(sink);
}

int main() {
    for (int nt : {256, 512, 1024}) {
        run<64, 96, 32, 0>(nt, "layer 64 MFMA | 96 fma + 32 exp");
        run<64, 96, 32, 1>(nt, "  static: waves 0-3 high");
        run<64, 96, 32, 2>(nt, "  VALU phase high");
        run<64, 96, 32, 3>(nt, "  MFMA phase high");
    }
    // finer interleave: 16 MFMA | 24 fma + 8 exp (same ratio)
    for (int nt : {256, 512}) {
        run<16, 24, 8, 0>(nt, "fine 16 MFMA | 24 fma + 8 exp");
        run<16, 24, 8, 1>(nt, "  static: waves 0-3 high");
        run<16, 24, 8, 2>(nt, "  VALU phase high");
    }
    return 0;
}
