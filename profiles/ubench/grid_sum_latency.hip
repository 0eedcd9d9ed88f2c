// Latency of one grid-wide sum of three doubles (the step controller of the one-launch adaptive solves meets in 2 - 3 of them per
// step) for G resident workgroups of 256 threads:
//   A  partials into slots, release fetch-add on a counter, poll the counter, acquire fence, read the slots   (grid_sum3, round 2)
//   B  every partial travels as two 64-bit words {half of the double, round tag}, written and polled with relaxed agent-scope
//      atomics: no counter, no release / acquire fence (no L2 write-back / invalidate), one load round trip per poll
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../continuousnormalizingflows.jl_amd/csrc grid_sum_latency.hip -o grid_sum_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct Q { double* slots; unsigned* counter; unsigned long long* words; int* abort_flag; };

__device__ __forceinline__ bool sum_a(const Q& q, unsigned& round, int wave, int nact, int lane, double& v0, double& v1, double& v2) {
    __shared__ double part[3][16];
    __shared__ double tot[3];
    __shared__ int okf;
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
    if (lane == 0) { part[0][wave] = v0; part[1][wave] = v1; part[2][wave] = v2; }
    __syncthreads();
    if (wave == 0) {
        const unsigned nb = gridDim.x;
        double* sl = q.slots + (size_t)(round & 1u) * 3u * nb;
        int ok = 1;
        if (lane == 0) {
            double a0 = 0, a1 = 0, a2 = 0;
            for (int w = 0; w < nact; ++w) { a0 += part[0][w]; a1 += part[1][w]; a2 += part[2][w]; }
            sl[blockIdx.x] = a0; sl[nb + blockIdx.x] = a1; sl[2 * nb + blockIdx.x] = a2;
            __hip_atomic_fetch_add(q.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (round + 1u) * nb;
            unsigned polls = 0;
            while (__hip_atomic_load(q.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > (1u << 20)) { ok = 0; break; }
            }
        }
        ok = __builtin_amdgcn_readfirstlane(ok);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double a0 = 0, a1 = 0, a2 = 0;
        for (unsigned i = lane; i < nb; i += 64) { a0 += sl[i]; a1 += sl[nb + i]; a2 += sl[2 * nb + i]; }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) { tot[0] = a0; tot[1] = a1; tot[2] = a2; okf = ok; }
    }
    __syncthreads();
    v0 = tot[0]; v1 = tot[1]; v2 = tot[2];
    ++round;
    return okf != 0;
}

// words: [2 parities][nb workgroups][6]: word 2 k + h = {32-bit half h of value k (low bits), tag (high bits)}
__device__ __forceinline__ bool sum_b(const Q& q, unsigned& round, int wave, int nact, int lane, double& v0, double& v1, double& v2) {
    __shared__ double part[3][16];
    __shared__ double tot[3];
    __shared__ int okf;
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
    if (lane == 0) { part[0][wave] = v0; part[1][wave] = v1; part[2][wave] = v2; }
    __syncthreads();
    if (wave == 0) {
        const unsigned nb = gridDim.x;
        const unsigned long long tag = (unsigned long long)(round + 1u) << 32;
        unsigned long long* wd = q.words + (size_t)(round & 1u) * 6u * nb;
        if (lane < 6) {   // lane 2 k + h publishes half h of value k
            const int k = lane >> 1, h = lane & 1;
            double a = 0;
            for (int w = 0; w < nact; ++w) a += part[k][w];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(a);
            const unsigned long long half = h ? (bits >> 32) : (bits & 0xffffffffull);
            __hip_atomic_store(wd + (size_t)blockIdx.x * 6 + lane, tag | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // every lane polls the words i = lane, lane + 64, ... of the 6 nb until all carry this round's tag
        int ok = 1;
        double a[3] = {0, 0, 0};
        const unsigned nw = 6u * nb;
        unsigned polls = 0;
        // pass 1: wait for all words; pass 2 (below) sums in workgroup order from the values this lane holds
        unsigned long long mine[4];   // nb <= 42 here (4 x 64 words)
        for (int j = 0; j < 4; ++j) mine[j] = tag;
        bool all;
        do {
            all = true;
            for (int j = 0; j < 4; ++j) {
                const unsigned i = lane + 64u * j;
                if (i < nw) {
                    mine[j] = __hip_atomic_load(wd + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    all = all && (mine[j] >> 32) == (tag >> 32);
                }
            }
            all = __all(all);
            if (!all && ++polls > (1u << 20)) { ok = 0; break; }
        } while (!all);
        // value k of workgroup b: words 6 b + 2 k (+1); gather through LDS
        __shared__ unsigned halves[4 * 64];
        for (int j = 0; j < 4; ++j) halves[lane + 64 * j] = (unsigned)(mine[j] & 0xffffffffull);
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        if (lane < 3) {
            double s = 0;
            for (unsigned b = 0; b < nb; ++b) {
                const unsigned lo = halves[6 * b + 2 * lane], hi = halves[6 * b + 2 * lane + 1];
                s += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            }
            tot[lane] = s;
        }
        if (lane == 0) okf = ok;
        (void)a;
    }
    __syncthreads();
    v0 = tot[0]; v1 = tot[1]; v2 = tot[2];
    ++round;
    return okf != 0;
}

template <int MODE>
__global__ void __launch_bounds__(256) bench_kernel(Q q, int rounds, double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned round = 0;
    double acc = 0;
    for (int r = 0; r < rounds; ++r) {
        double v0 = 1.0 + lane * 1e-3 + blockIdx.x, v1 = 2.0, v2 = (double)r;
        bool ok = MODE == 0 ? sum_a(q, round, wave, 4, lane, v0, v1, v2) : sum_b(q, round, wave, 4, lane, v0, v1, v2);
        if (!ok) break;
        acc += v0 + v1 + v2;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main() {
    Q q;
    const int GMAX = 42, R = 2000;
    hipMalloc(&q.slots, 6 * GMAX * sizeof(double)); hipMalloc(&q.counter, 64); hipMalloc(&q.words, 12 * GMAX * sizeof(unsigned long long));
    hipMalloc(&q.abort_flag, 64);
    double* out; hipMalloc(&out, GMAX * sizeof(double));
    double host[GMAX];
    for (int G : {1, 2, 4, 8, 16, 32, 42}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(q.counter, 0, 64); hipMemset(q.words, 0, 12 * GMAX * sizeof(unsigned long long)); hipMemset(q.slots, 0, 6 * GMAX * sizeof(double));
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(bench_kernel<0>, dim3(G), dim3(256), 0, 0, q, R, out);
                else hipLaunchKernelGGL(bench_kernel<1>, dim3(G), dim3(256), 0, 0, q, R, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            hipMemcpy(host, out, G * sizeof(double), hipMemcpyDeviceToHost);
            printf("G = %2d  %s : %.2f us per sum   (check %.6e)\n", G, mode == 0 ? "A counter + fences" : "B tagged words     ", best * 1e3 / R, host[0]);
        }
    }
    return 0;
}
