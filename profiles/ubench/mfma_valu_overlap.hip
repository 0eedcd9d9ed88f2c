// Microbenchmark (evidence for DESIGN.md §4): does v_mfma_f32_16x16x4_f32 overlap with f32 VALU /
// transcendental work on gfx950 — inside one wave, and between the two waves sharing a SIMD?
//   hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// body: 16 MFMAs on 4 independent accumulators; after each MFMA, KF v_fma and KE v_exp (independent chains)
template <int KF, int KE, int MF>
__device__ __forceinline__ void body(f32x4 (&acc)[4], float (&v)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (MF) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < KF; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < KE; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(k + 4) & 7]));
    }
}

// mode 0: every wave runs body<KF,KE,1>.  mode 1: waves 0-3 MFMA-only, waves 4-7 VALU-only (KF,KE per slot)
template <int KF, int KE>
__global__ void __launch_bounds__(512) k(int iters, int mode, long long* cyc, float* sink) {
    f32x4 acc[4] = {};
    float v[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    const float a = 1e-3f * threadIdx.x, b = 0.5f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
        for (int it = 0; it < iters; ++it) body<KF, KE, 1>(acc, v, a, b);
    } else if (wave < 4) {
        for (int it = 0; it < iters; ++it) body<0, 0, 1>(acc, v, a, b);
    } else {
        for (int it = 0; it < iters; ++it) body<KF, KE, 0>(acc, v, a, b);
    }
    asm volatile("s_nop 7\n s_nop 7");
    long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1]; for (int i = 0; i < 8; ++i) s += v[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KF, int KE>
void run(int nthreads, int mode, const char* label) {
    const int iters = 2000, nblk = 256;
    long long* cyc; float* sink;
    hipMalloc(&cyc, sizeof(long long) * nblk * 8); hipMalloc(&sink, sizeof(float) * nblk * 512);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KF, KE>), dim3(nblk), dim3(nthreads), 0, 0, iters, mode, cyc, sink);
    hipDeviceSynchronize();
    std::vector<long long> h(nblk * 8);
    hipMemcpy(h.data(), cyc, sizeof(long long) * nblk * (nthreads / 64), hipMemcpyDeviceToHost);
    const int wpb = nthreads / 64;
    // s_memtime ticks at 100 MHz * ? -> report raw ticks per MFMA slot; also per-role medians
    std::vector<long long> lo, hi;
    for (int bI = 0; bI < nblk; ++bI) for (int w = 0; w < wpb; ++w) (w < 4 ? lo : hi).push_back(h[bI * wpb + w]);
    std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end());
    const double per = 1.0 / (iters * 16.0);
    printf("%-46s nt=%4d mode=%d KF=%d KE=%d  waves0-3: %.2f ticks/slot", label, nthreads, mode, KF, KE, lo[lo.size() / 2] * per);
    if (!hi.empty()) printf("   waves4-7: %.2f ticks/slot", hi[hi.size() / 2] * per);
    printf("\n");
    hipFree(cyc); hipFree(sink);
}

int main() {
    printf("slot = one MFMA (+KF v_fma +KE v_exp).  s_memtime ticks (shader clock).\n");
    run<0, 0>(256, 0, "1 wave/SIMD, MFMA only");
    run<1, 0>(256, 0, "1 wave/SIMD, MFMA + 1 fma");
    run<2, 0>(256, 0, "1 wave/SIMD, MFMA + 2 fma");
    run<4, 0>(256, 0, "1 wave/SIMD, MFMA + 4 fma");
    run<6, 0>(256, 0, "1 wave/SIMD, MFMA + 6 fma");
    run<8, 0>(256, 0, "1 wave/SIMD, MFMA + 8 fma");
    run<0, 1>(256, 0, "1 wave/SIMD, MFMA + 1 exp");
    run<0, 2>(256, 0, "1 wave/SIMD, MFMA + 2 exp");
    run<0, 4>(256, 0, "1 wave/SIMD, MFMA + 4 exp");
    run<2, 1>(256, 0, "1 wave/SIMD, MFMA + 2 fma + 1 exp");
    run<0, 0>(512, 0, "2 waves/SIMD, both MFMA only");
    run<2, 0>(512, 0, "2 waves/SIMD, both MFMA + 2 fma");
    run<4, 0>(512, 0, "2 waves/SIMD, both MFMA + 4 fma");
    run<4, 0>(512, 1, "2 waves/SIMD, A: MFMA only | B: 4 fma/slot");
    run<8, 0>(512, 1, "2 waves/SIMD, A: MFMA only | B: 8 fma/slot");
    run<0, 2>(512, 1, "2 waves/SIMD, A: MFMA only | B: 2 exp/slot");
    run<0, 4>(512, 1, "2 waves/SIMD, A: MFMA only | B: 4 exp/slot");
    return 0;
}
