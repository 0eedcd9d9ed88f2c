// Microbenchmark 3: would a 3-way bf16 split (6 bf16 MFMAs per product, f32-equivalent accuracy)
// beat the exact-f32 MFMA layer?  One 64x64 layer for a 16-sample tile:
//   f32   : 64 v_mfma_f32_16x16x4_f32   + 16 ds_read_b128 + activation (96 fma-class + 32 trans)
//   bf16x6: 48 v_mfma_f32_16x16x32_bf16 + 24 ds_read_b128 + activation + split (NSPLIT extra VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV, int NE>
__global__ void __launch_bounds__(1024) k(int iters, long long* cyc, float* sink) {
    __shared__ f32x4 lds[1024];
    lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    f32x4 acc[4] = {};
    f32x4 w = {0.1f, 0.2f, 0.3f, 0.4f};
    bf16x8 wa = {1, 2, 3, 4, 5, 6, 7, 8}, wb = {1, 1, 2, 2, 3, 3, 4, 4};
    float v[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    const float a = 1e-3f * threadIdx.x, b = 0.5f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned addr = (threadIdx.x & 63) * 16;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                if ((i & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(addr));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                if ((i & 1) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(addr));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(wa), "v"(wb));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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

template <int MODE, int NV, int NE>
void run(int nthreads, const char* label) {
    const int iters = 500, nblk = 256;
    long long* cyc; float* sink;
    (void)hipMalloc(&cyc, sizeof(long long) * nblk * 16); (void)hipMalloc(&sink, sizeof(float) * nblk * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, NV, NE>), dim3(nblk), dim3(nthreads), 0, 0, iters, cyc, sink);
    (void)hipDeviceSynchronize();
    const int wpb = nthreads / 64;
    std::vector<long long> h(nblk * 16);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * nblk * 16, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int bI = 0; bI < nblk; ++bI) for (int w = 0; w < wpb; ++w) mx = std::max(mx, h[bI * 16 + w]);
    printf("%-40s waves/SIMD=%d NV=%3d NE=%2d : %7.0f ticks per layer-tile (SIMD time / tiles)\n", label, wpb / 4, NV, NE,
           (double)mx / iters / (wpb / 4.0));
    (void)hipFree(cyc); (void)hipFree(sink);
}

int main() {
    for (int nt : {256, 512, 1024}) {
        run<0, 96, 32>(nt, "f32 layer (64 mfma + act)");
        run<1, 0, 0>(nt, "bf16x6 MFMAs only (48)");
        run<1, 96, 32>(nt, "bf16x6 + act only");
        run<1, 184, 32>(nt, "bf16x6 + act + split (88 extra valu)");
        run<1, 240, 32>(nt, "bf16x6 + act + split (144 extra valu)");
    }
    return 0;
}
