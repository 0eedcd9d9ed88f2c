// Microbenchmark 2: phase-structured waves.  Each wave repeats [NM MFMAs (4 accumulators)] then
// [NV v_fma + NE v_exp] — the shape of one MLP layer (products, then activation).  How much of the
// VALU phase do 1 / 2 / 3 / 4 co-resident waves per SIMD hide?  Also: cost of ds_read_b128 in the
// MFMA stream.   hipcc -O3 --offload-arch=gfx950 mfma_phase_overlap.hip -o mfma_phase_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NE, int LDS_EVERY, int PRIO>
__global__ void __launch_bounds__(1024) k(int iters, long long* cyc, float* sink) {
    __shared__ f32x4 lds[1024];
    lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    f32x4 acc[4] = {};
    f32x4 w = {0.1f, 0.2f, 0.3f, 0.4f};
    float v[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    const float a = 1e-3f * threadIdx.x, b = 0.5f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned addr = (threadIdx.x & 63) * 16;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (PRIO) asm volatile("s_setprio 1");
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (LDS_EVERY > 0 && (i % LDS_EVERY) == 0)
                asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(addr));
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
        }
        if (LDS_EVERY > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (PRIO) asm volatile("s_setprio 0");
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
    }
    asm volatile("s_nop 7\n s_nop 7");
    long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    float s = w[0]; for (int i = 0; i < 4; ++i) s += acc[i][0]; for (int i = 0; i < 8; ++i) s += v[i];
    sink[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int NM, int NV, int NE, int LDS_EVERY, int PRIO = 0>
void run(int nthreads, const char* label) {
    const int iters = 500, nblk = 256;
    long long* cyc; float* sink;
    (void)hipMalloc(&cyc, sizeof(long long) * nblk * 16); (void)hipMalloc(&sink, sizeof(float) * nblk * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NM, NV, NE, LDS_EVERY, PRIO>), dim3(nblk), dim3(nthreads), 0, 0, iters, cyc, sink);
    (void)hipDeviceSynchronize();
    const int wpb = nthreads / 64;
    std::vector<long long> h(nblk * 16);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * nblk * 16, hipMemcpyDeviceToHost);
    long long mx = 0;   // slowest wave = SIMD completion time
    for (int bI = 0; bI < nblk; ++bI) for (int w = 0; w < wpb; ++w) mx = std::max(mx, h[bI * 16 + w]);
    const double waves_per_simd = wpb / 4.0;
    const double mfma_cycles = 32.0 * NM * iters * waves_per_simd;
    printf("%-44s waves/SIMD=%.0f  NM=%3d NV=%3d NE=%2d lds/%d : %8.0f ticks/iter/wave  MFMA-busy %.1f%%\n", label,
           waves_per_simd, NM, NV, NE, LDS_EVERY, (double)mx / iters, 100.0 * mfma_cycles / mx);
    (void)hipFree(cyc); (void)hipFree(sink);
}

int main() {
    // one 64x64 layer for a 16-sample tile: 64 MFMAs; activation: 16 elements x (6 fma-class + 2 transcendental)
    for (int nt : {256, 512, 768, 1024}) {
        run<64, 0, 0, 0>(nt, "MFMA only");
        run<64, 96, 32, 0>(nt, "layer: 64 MFMA | 96 fma + 32 exp");
        run<64, 96, 32, 4>(nt, "layer + ds_read_b128 per 4 MFMA");
        run<64, 48, 16, 4>(nt, "half VALU + ds_read per 4");
        run<64, 192, 64, 4>(nt, "double VALU + ds_read per 4");
        run<64, 96, 32, 4, 1>(nt, "layer + lds, setprio 1 around MFMA phase");
        run<64, 48, 16, 4, 1>(nt, "half VALU + lds, setprio around MFMA");
        run<64, 192, 64, 4, 1>(nt, "double VALU + lds, setprio around MFMA");
    }
    run<64, 0, 0, 1>(256, "ds_read_b128 per MFMA");
    run<64, 0, 0, 2>(256, "ds_read_b128 per 2 MFMA");
    run<64, 0, 0, 4>(256, "ds_read_b128 per 4 MFMA");
    return 0;
}
