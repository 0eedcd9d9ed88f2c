// cnf_lgemm.hip — the products of the layer-wise path (cnf_layered.hip) as hand-written gfx950 kernels.
//
// The layer-wise path serves every Dense chain the fused kernels do not (wide layers such as BASELINE config 4's gradient,
// unequal widths, mixed activations, many probes): one dynamics evaluation / one reverse-sweep stage is a sequence of
// [features x B] products with elementwise work between them (src/core/icnf.jl:517-559, src/core/utils.jl:150-170 unfused, as the
// reference's Lux + Zygote path is).  Round 1 sent the products to rocBLAS and ran the elementwise pieces as separate launches;
// here both kinds of product are v_mfma_f32_16x16x4_f32 kernels with the elementwise work fused into their epilogues:
//
//   lg_gemm   Out[M x B] = A[M x K] In[K x B]  - A a weight matrix (or its transpose, or Q) pre-packed once per parameter set
//             into MFMA operand order, In / Out column-major with one column per sample (the ABI's layout).  A 256-thread
//             workgroup owns 64 samples: their input columns are staged once into LDS as the B-operand image (16 B per lane,
//             conflict-free), wave w computes every fourth 16-row tile for all four 16-sample tiles, weight fragments stream
//             from L2 (16 B per lane, each feeding 16 MFMAs).  Epilogues: plain store; activation (h and act' in one pass, plus
//             the ones row the next product's bias rides on); product with another [M x B] array (the pullback's .* act').
//   lg_wgrad  C[M x Nc] += X[M x B] Y[Nc x B]^T (weight cotangents: samples on the MFMA K axis).  A workgroup owns 64 rows of C
//             and one chunk of samples; wave w keeps one 16-row strip of C - up to 17 accumulator tiles - in registers for its
//             whole chunk and adds it to the chunk's slab once, so the slab traffic is (#chunks x |C|) per call instead of per
//             256 columns.  Slabs are summed in a fixed order at the end (no atomics: bit-reproducible).
//
// Contraction order is a fixed function of the shapes, so results are deterministic and independent of B's partition into
// workgroups (each output element is one fmaf chain over k in image order).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "cnf_mfma_dev.h"

namespace cnf {

// ---- A-operand image: [mt][kq][lane = 16 g + i][j] = A(16 mt + i, 16 kq + 4 g + j), A(m, k) = src[m * sm + k * sk] ----
// (natural row order, so accumulator register r of lane group g is output row 16 mt + 4 g + r: four consecutive rows per
// lane = one 16-byte store; the k order inside a 16-group is permuted, which a contraction does not see as long as the B
// operand uses the same map: component j of lane group g <-> k = 16 kq + 4 g + j, four consecutive input rows = one 16-byte load)
__global__ void lg_pack_kernel(const float* __restrict__ src, long long sm, long long sk, int M, int K, int KQ,
                               float* __restrict__ img, long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int j = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const long long tile = e >> 8;
    const int kq = (int)(tile % KQ), mt = (int)(tile / KQ);
    const int row = 16 * mt + (lane & 15), k = 16 * kq + 4 * (lane >> 4) + j;
    img[e] = (row < M && k < K) ? src[(long long)row * sm + (long long)k * sk] : 0.f;
}

hipError_t lg_pack_image(const float* src, long long sm, long long sk, int M, int K, float* img, hipStream_t st) {
    const int MT = (M + 15) / 16, KQ = (K + 15) / 16;
    const long long n = (long long)MT * KQ * 256;
    hipLaunchKernelGGL(lg_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, sm, sk, M, K, KQ, img, n);
    return hipGetLastError();
}
size_t lg_image_floats(int M, int K) { return (size_t)((M + 15) / 16) * ((K + 15) / 16) * 256; }

// epilogues (x = the product):
//   PLAIN   out = x
//   ACT     out = act(x), dout = act'(x), ones row behind out
//   MUL     out = x .* e
//   MUL2    out = x, dout = x .* e                                   (pullback: v_l and delta_l = v_l .* act'_l)
//   BOTTOM  out = x .* e, dout (+)= x .* e2  (first: =)              (reverse of the pullback: vbar_l, acc2_l)
//   SBAR    out = x .* e + e2 .* act''  (act'' from e = act' and the activations a3)   (reverse of the forward chain)
enum { LG_EPI_PLAIN = 0, LG_EPI_ACT = 1, LG_EPI_MUL = 2, LG_EPI_MUL2 = 3, LG_EPI_BOTTOM = 4, LG_EPI_SBAR = 5 };

// four consecutive floats of a column whose stride (H + 1 for the arrays that carry a ones row) is not a multiple of four:
// one 16-byte access with 4-byte alignment (gfx950 global memory accesses need dword alignment only)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct LgGemmArgs {
    const float* img;      // A image
    const float* in;       // [K x N], column stride ldb
    float* out;            // [M x N], column stride ldc
    const float* e;        // MUL / MUL2 / BOTTOM / SBAR: [M x N] factor (act'), column stride lde
    float* dout;           // ACT: act'; MUL2: x .* e; BOTTOM: acc2;  [M x N], column stride ldd
    const float* e2;       // BOTTOM: v; SBAR: acc2;  [M x N], column stride lde
    const float* a3;       // SBAR: the layer's activations [M x N], column stride ld3 (tanh: act'' = -2 a act')
    long long N;
    int M, K, MT, KQ, ldb, ldc, lde, ldd, ld3, act, first;
    int mts;               // M-tiles per workgroup row (blockIdx.y picks the group): wide outputs are split over several workgroups
    int spw;               // pipelined kernel: 32-sample sub-panels per workgroup
};

// MTW: 16-row tiles per wave (MT <= 4 MTW); NQ: 16-sample tiles per workgroup (each weight fragment feeds NQ sample tiles)
template <int MTW, int NQ, int EPI>
__global__ void __launch_bounds__(256)
lg_gemm_kernel(LgGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* bimg = reinterpret_cast<f32x4*>(smem);                 // [KQ][NQ sample tiles][64 lanes]
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long s0 = (long long)blockIdx.x * (16 * NQ);
    // stage the input columns: lane (g, n) holds rows 16 kq + 4 g .. + 3 of column s0 + 16 q + n; wave w stages tiles w, w + 4, ...
#pragma unroll
    for (int qq = 0; qq < NQ / 4; ++qq) {
        const int q = wave + 4 * qq;
        const long long col = s0 + 16 * q + n;
        const bool cv = col < a.N;
        const float* src = a.in + (cv ? col : 0) * (long long)a.ldb;
        for (int kq = 0; kq < a.KQ; ++kq) {
            const int k0 = 16 * kq + 4 * g;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cv) {
                if (k0 + 3 < a.K) {
                    const f32x4u t = *reinterpret_cast<const f32x4u*>(src + k0);
                    v = f32x4{t[0], t[1], t[2], t[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k0 + j < a.K) v[j] = src[k0 + j];
                }
            }
            bimg[(kq * NQ + q) * 64 + lane] = v;
        }
    }
    __syncthreads();
    const f32x4* __restrict__ A = reinterpret_cast<const f32x4*>(a.img) + lane;
    f32x4 acc[MTW][NQ];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // M-tiles of this wave: wave, wave + 4, ...; tiles past MT read a clamped (valid) image tile and are never stored
    const int mt_lo = blockIdx.y * a.mts;
    const int mt_hi = mt_lo + a.mts < a.MT ? mt_lo + a.mts : a.MT;
    int mts[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) { const int t = mt_lo + wave + 4 * m; mts[m] = t < mt_hi ? t : mt_hi - 1; }
    f32x4 a0[MTW], a1[MTW], b0[NQ], b1[NQ];
#pragma unroll
    for (int m = 0; m < MTW; ++m) a0[m] = A[((long long)mts[m] * a.KQ + 0) * 64];
#pragma unroll
    for (int q = 0; q < NQ; ++q) b0[q] = bimg[(0 * NQ + q) * 64 + lane];
    // (prefetch loads unconditional, k-group index clamped at the end: see lg_gemm2_kernel)
#pragma clang loop unroll(disable)
    for (int kq = 0; kq < a.KQ; kq += 2) {
        const int k1 = kq + 1 < a.KQ ? kq + 1 : a.KQ - 1, k2 = kq + 2 < a.KQ ? kq + 2 : a.KQ - 1;
#pragma unroll
        for (int m = 0; m < MTW; ++m) a1[m] = A[((long long)mts[m] * a.KQ + k1) * 64];
#pragma unroll
        for (int q = 0; q < NQ; ++q) b1[q] = bimg[(k1 * NQ + q) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[m][q] = mfma4(a0[m][j], b0[q][j], acc[m][q]);
#pragma unroll
        for (int m = 0; m < MTW; ++m) a0[m] = A[((long long)mts[m] * a.KQ + k2) * 64];
#pragma unroll
        for (int q = 0; q < NQ; ++q) b0[q] = bimg[(k2 * NQ + q) * 64 + lane];
        if (kq + 1 < a.KQ) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[m][q] = mfma4(a1[m][j], b1[q][j], acc[m][q]);
        }
    }
    // epilogue: lane (g, n) of tile (mt, q) holds rows 16 mt + 4 g .. + 3 of column s0 + 16 q + n
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int mt = mt_lo + wave + 4 * m;
        if (mt >= mt_hi) continue;
        const int r0 = 16 * mt + 4 * g;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const long long col = s0 + 16 * q + n;
            if (col >= a.N) continue;
            f32x4 v = acc[m][q], dv = {0.f, 0.f, 0.f, 0.f};
            const bool full = r0 + 3 < a.M;
            if (EPI == LG_EPI_ACT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { float dd; v[r] = act_fwd_rt(a.act, v[r], dd); dv[r] = dd; }
            }
            if (EPI == LG_EPI_MUL || EPI == LG_EPI_MUL2 || EPI == LG_EPI_BOTTOM || EPI == LG_EPI_SBAR) {
                // elementwise operands of this lane's four rows (row-guarded scalar loads at the ragged edge)
                auto load4 = [&](const float* base, int ld) -> f32x4 {
                    const float* p = base + col * (long long)ld + r0;
                    if (full) { const f32x4u t = *reinterpret_cast<const f32x4u*>(p); return f32x4{t[0], t[1], t[2], t[3]}; }
                    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r0 + r < a.M) t[r] = p[r];
                    return t;
                };
                const f32x4 ev = load4(a.e, a.lde);
                if (EPI == LG_EPI_MUL) v *= ev;
                if (EPI == LG_EPI_MUL2) dv = v * ev;
                if (EPI == LG_EPI_BOTTOM) {
                    const f32x4 vv = load4(a.e2, a.lde);
                    f32x4 prev = {0.f, 0.f, 0.f, 0.f};
                    if (!a.first) prev = load4(a.dout, a.ldd);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dv[r] = a.first ? v[r] * vv[r] : fmaf(v[r], vv[r], prev[r]);
                    v *= ev;
                }
                if (EPI == LG_EPI_SBAR) {
                    const f32x4 c2 = load4(a.e2, a.lde);
                    f32x4 e2v = {0.f, 0.f, 0.f, 0.f};
                    if (a.act == CNF_ACT_TANH) {
                        const f32x4 av = load4(a.a3, a.ld3);
#pragma unroll
                        for (int r = 0; r < 4; ++r) e2v[r] = -2.f * av[r] * ev[r];
                    } else if (a.act == CNF_ACT_SOFTPLUS) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) e2v[r] = ev[r] * (1.f - ev[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(c2[r], e2v[r], v[r] * ev[r]);
                }
            }
            float* op = a.out + col * (long long)a.ldc + r0;
            if (full) {
                *reinterpret_cast<f32x4u*>(op) = f32x4u{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r0 + r < a.M) op[r] = v[r];
            }
            if (EPI == LG_EPI_ACT || EPI == LG_EPI_MUL2 || EPI == LG_EPI_BOTTOM) {
                float* dp = a.dout + col * (long long)a.ldd + r0;
                if (full) {
                    *reinterpret_cast<f32x4u*>(dp) = f32x4u{dv[0], dv[1], dv[2], dv[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r0 + r < a.M) dp[r] = dv[r];
                }
                // the ones row behind the activations (row M of a column of ldc = M + 1 floats: the next product's bias rides on
                // it), written by the lane that owns the last valid row
                if (EPI == LG_EPI_ACT && a.ldc > a.M && r0 <= a.M - 1 && a.M - 1 < r0 + 4) a.out[col * (long long)a.ldc + a.M] = 1.f;
            }
        }
    }
}

// ---- pipelined variant: a workgroup walks over several 32-sample sub-panels; while the MFMAs of one sub-panel run, the input
// columns of the next are in flight (global -> registers, parked in the other LDS buffer after the epilogue).  The fused
// products of the reverse sweep move ~130 MB for 4.3 GFLOP at cfg4's shapes - HBM time ~ MFMA time - so the one-panel kernel
// above (stage, then multiply, then epilogue, one workgroup per CU) runs at the SUM of the two; this one at their maximum.
// 68 KB of LDS (K <= 272) and <= 256 registers: two workgroups per CU, whose epilogue stalls overlap the other's MFMAs.
template <int MTW, int EPI>
__device__ __forceinline__ void lg_epilogue_tile(const LgGemmArgs& a, f32x4 v, int r0, long long col) {
    f32x4 dv = {0.f, 0.f, 0.f, 0.f};
    const bool full = r0 + 3 < a.M;
    if (EPI == LG_EPI_ACT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { float dd; v[r] = act_fwd_rt(a.act, v[r], dd); dv[r] = dd; }
    }
    if (EPI == LG_EPI_MUL || EPI == LG_EPI_MUL2 || EPI == LG_EPI_BOTTOM || EPI == LG_EPI_SBAR) {
        auto load4 = [&](const float* base, int ld) -> f32x4 {
            const float* p = base + col * (long long)ld + r0;
            if (full) { const f32x4u t = *reinterpret_cast<const f32x4u*>(p); return f32x4{t[0], t[1], t[2], t[3]}; }
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r0 + r < a.M) t[r] = p[r];
            return t;
        };
        const f32x4 ev = load4(a.e, a.lde);
        if (EPI == LG_EPI_MUL) v *= ev;
        if (EPI == LG_EPI_MUL2) dv = v * ev;
        if (EPI == LG_EPI_BOTTOM) {
            const f32x4 vv = load4(a.e2, a.lde);
            f32x4 prev = {0.f, 0.f, 0.f, 0.f};
            if (!a.first) prev = load4(a.dout, a.ldd);
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[r] = a.first ? v[r] * vv[r] : fmaf(v[r], vv[r], prev[r]);
            v *= ev;
        }
        if (EPI == LG_EPI_SBAR) {
            const f32x4 c2 = load4(a.e2, a.lde);
            f32x4 e2v = {0.f, 0.f, 0.f, 0.f};
            if (a.act == CNF_ACT_TANH) {
                const f32x4 av = load4(a.a3, a.ld3);
#pragma unroll
                for (int r = 0; r < 4; ++r) e2v[r] = -2.f * av[r] * ev[r];
            } else if (a.act == CNF_ACT_SOFTPLUS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) e2v[r] = ev[r] * (1.f - ev[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(c2[r], e2v[r], v[r] * ev[r]);
        }
    }
    float* op = a.out + col * (long long)a.ldc + r0;
    if (full) {
        *reinterpret_cast<f32x4u*>(op) = f32x4u{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r0 + r < a.M) op[r] = v[r];
    }
    if (EPI == LG_EPI_ACT || EPI == LG_EPI_MUL2 || EPI == LG_EPI_BOTTOM) {
        float* dp = a.dout + col * (long long)a.ldd + r0;
        if (full) {
            *reinterpret_cast<f32x4u*>(dp) = f32x4u{dv[0], dv[1], dv[2], dv[3]};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r0 + r < a.M) dp[r] = dv[r];
        }
        if (EPI == LG_EPI_ACT && a.ldc > a.M && r0 <= a.M - 1 && a.M - 1 < r0 + 4) a.out[col * (long long)a.ldc + a.M] = 1.f;
    }
}

constexpr int LG2_KQ_MAX = 17;                        // K <= 272: 2 buffers x 17 x 2 KB = 68 KB of LDS, two workgroups per CU
constexpr int LG2_KQ_WIDE = 32;                       // K <= 512: up to 128 KB, one workgroup of eight waves per CU (KQM = 32 instances)

// NW waves per workgroup (4 or 8): with 8, a wave owns half as many row tiles, needs ~110 registers, and four waves share a
// SIMD (two workgroups per CU) - twice the memory latency covered per SIMD
// KQM: the most k-groups of 16 an instance stages (sizes the staging registers; the LDS need follows the call's own KQ)
template <int MTW, int EPI, int NW, int KQM = LG2_KQ_MAX>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 2, NW / 2)))
lg_gemm2_kernel(LgGemmArgs a) {
    constexpr int LG2_UNITS = (2 * KQM + NW - 1) / NW;   // staging units (1 KB: one (kq, q) tile) per wave and sub-panel
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* bimg = reinterpret_cast<f32x4*>(smem);                 // [2 buffers][KQ][2 sample tiles][64 lanes]
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KQ = a.KQ, nunits = 2 * KQ, bufsz = nunits * 64;
    const long long nsub = (a.N + 31) / 32;
    const long long sp0 = (long long)blockIdx.x * a.spw;
    const int cnt = (int)(sp0 + a.spw <= nsub ? a.spw : nsub - sp0);
    f32x4 stage[LG2_UNITS];
    // unit u = 2 kq + q: lane (g, n) holds rows 16 kq + 4 g .. + 3 of column 32 sp + 16 q + n; wave w takes units w, w + 4, ...
    auto fetch = [&](long long sp) {
#pragma unroll
        for (int i = 0; i < LG2_UNITS; ++i) {
            const int u = wave + NW * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < nunits) {
                const int kq = u >> 1, q = u & 1, k0 = 16 * kq + 4 * g;
                const long long col = sp * 32 + 16 * q + n;
                if (col < a.N) {
                    const float* src = a.in + col * (long long)a.ldb + k0;
                    if (k0 + 3 < a.K) {
                        const f32x4u t = *reinterpret_cast<const f32x4u*>(src);
                        v = f32x4{t[0], t[1], t[2], t[3]};
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (k0 + j < a.K) v[j] = src[j];
                    }
                }
            }
            stage[i] = v;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LG2_UNITS; ++i) {
            const int u = wave + NW * i;
            if (u < nunits) bimg[buf * bufsz + u * 64 + lane] = stage[i];
        }
    };
    // weight fragments: buffer loads, the fragment's image offset wave-uniform in an SGPR, the lane's 16-byte slot in a VGPR
    // that never changes (no address VALU in the loop)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.img), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    auto ldA = [&](int mt, int kq) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, lane16, (mt * KQ + kq) * 1024, 0));
    };
    const int mt_lo = blockIdx.y * a.mts;
    const int mt_hi = mt_lo + a.mts < a.MT ? mt_lo + a.mts : a.MT;
    int mts[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) { const int t = mt_lo + wave + NW * m; mts[m] = t < mt_hi ? t : mt_hi - 1; }
    if (cnt <= 0) return;
    fetch(sp0);
    park(0);
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int it = 0; it < cnt; ++it) {
        const long long sp = sp0 + it;
        const bool more = it + 1 < cnt;
        if (more) fetch(sp + 1);
        const f32x4* bb = bimg + (it & 1) * bufsz + lane;
        f32x4 acc[MTW][2];
#pragma unroll
        for (int m = 0; m < MTW; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 a0[MTW], a1[MTW], b0[2], b1[2];
#pragma unroll
        for (int m = 0; m < MTW; ++m) a0[m] = ldA(mts[m], 0);
        b0[0] = bb[0]; b0[1] = bb[64];
        // The prefetch loads are UNCONDITIONAL (k-group index clamped at the end): a load inside an `if` makes a control-flow
        // join at which the compiler's wait-count insertion assumes the worst and waits for everything outstanding
        // (`s_waitcnt vmcnt(0) lgkmcnt(0)` right after the prefetch was issued - measured: no overlap at all, 57 us per product)
#pragma clang loop unroll(disable)
        for (int kq = 0; kq < KQ; kq += 2) {
            const int k1 = kq + 1 < KQ ? kq + 1 : KQ - 1, k2 = kq + 2 < KQ ? kq + 2 : KQ - 1;
#pragma unroll
            for (int m = 0; m < MTW; ++m) a1[m] = ldA(mts[m], k1);
            b1[0] = bb[(2 * k1) * 64]; b1[1] = bb[(2 * k1 + 1) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    acc[m][0] = mfma4(a0[m][j], b0[0][j], acc[m][0]);
                    acc[m][1] = mfma4(a0[m][j], b0[1][j], acc[m][1]);
                }
#pragma unroll
            for (int m = 0; m < MTW; ++m) a0[m] = ldA(mts[m], k2);
            b0[0] = bb[(2 * k2) * 64]; b0[1] = bb[(2 * k2 + 1) * 64];
            if (kq + 1 < KQ) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MTW; ++m) {
                        acc[m][0] = mfma4(a1[m][j], b1[0][j], acc[m][0]);
                        acc[m][1] = mfma4(a1[m][j], b1[1][j], acc[m][1]);
                    }
            }
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int mt = mt_lo + wave + NW * m;
            if (mt >= mt_hi) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const long long col = sp * 32 + 16 * q + n;
                if (col < a.N) lg_epilogue_tile<MTW, EPI>(a, acc[m][q], 16 * mt + 4 * g, col);
            }
        }
        if (more) park((it + 1) & 1);
        __syncthreads();
    }
}

template <int MTW, int EPI, int NW = 4, int KQM = LG2_KQ_MAX>
static hipError_t lg_gemm2_launch(const LgGemmArgs& a, hipStream_t st) {
    const int lds = 2 * a.KQ * 2 * 64 * 16;
    auto kern = lg_gemm2_kernel<MTW, EPI, NW, KQM>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (lds > 64 * 1024 && !once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KQM * 2 * 64 * 16 > 80 * 1024 ? 2 * KQM * 2 * 64 * 16 : 80 * 1024);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    const long long nsub = (a.N + 31) / 32;
    hipLaunchKernelGGL(kern, dim3((unsigned)((nsub + a.spw - 1) / a.spw), (unsigned)((a.MT + a.mts - 1) / a.mts)), dim3(64 * NW), lds, st, a);
    return hipGetLastError();
}

template <int MTW, int NQ, int EPI>
static hipError_t lg_gemm_launch(const LgGemmArgs& a, hipStream_t st) {
    const int lds = a.KQ * NQ * 64 * 16;
    auto kern = lg_gemm_kernel<MTW, NQ, EPI>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (lds > 64 * 1024 && !once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.N + 16 * NQ - 1) / (16 * NQ)), (unsigned)((a.MT + a.mts - 1) / a.mts)), dim3(256), lds, st, a);
    return hipGetLastError();
}

// largest shapes one launch covers: 32 row tiles (M <= 512), the 64-sample B panel in 160 KB of LDS (K <= 640)
bool lg_gemm_supported(int M, int K) { return M >= 1 && M <= 512 && K >= 1 && ((K + 15) / 16) * 4096 <= 160 * 1024; }

template <int EPI>
static hipError_t lg_gemm_dispatch(LgGemmArgs a, hipStream_t st) {
    // Outputs wider than 16 row tiles are split evenly over several workgroups per sample panel (at most 16 tiles = 4 per wave
    // each); 128-sample panels (every weight fragment feeds 8 sample tiles: half the L2 traffic of the images) when the panel
    // fits LDS and there are enough samples to fill the chip that way
    const int splits = (a.MT + 15) / 16;
    a.mts = (a.MT + splits - 1) / splits;
    const int variant = tuning().lg_gemm;
    if (variant == 2 && a.KQ <= LG2_KQ_MAX) {
        // enough sub-panels per workgroup for the pipeline to pay (>= 4 where the batch allows), at least ~2 workgroups per CU
        const int spw_env = tuning().lg_spw;
        const long long nsub = (a.N + 31) / 32;
        long long spw = nsub * splits / 512;
        if (spw < 1) spw = 1;
        if (spw > 8) spw = 8;
        if (spw_env > 0) spw = spw_env;
        a.spw = (int)spw;
        const int nw = tuning().lg_nw;
        if (nw == 8 && a.mts > 8) return lg_gemm2_launch<2, EPI, 8>(a, st);   // 9..16 row tiles over 8 waves
        if (a.mts <= 4) return lg_gemm2_launch<1, EPI>(a, st);
        if (a.mts <= 8) return lg_gemm2_launch<2, EPI>(a, st);
        return lg_gemm2_launch<4, EPI>(a, st);
    }
    // K = 273 .. 512 (the default architecture from nvariables = 33 on, where it runs layer-wise): the same pipeline, one workgroup
    // of eight waves per CU (the 392 x 392 products of nvariables = 48: 193 - 214 us on lg_gemm_kernel, 50 TFLOP/s)
    const int wide2 = tuning().lg_gemm2_wide;
    if (variant == 2 && wide2 && a.KQ <= LG2_KQ_WIDE) {
        const long long nsub = (a.N + 31) / 32;
        long long spw = nsub * splits / 256;
        if (spw < 1) spw = 1;
        if (spw > 8) spw = 8;
        a.spw = (int)spw;
        if (a.mts <= 8) return lg_gemm2_launch<1, EPI, 8, LG2_KQ_WIDE>(a, st);
        return lg_gemm2_launch<2, EPI, 8, LG2_KQ_WIDE>(a, st);
    }
    const bool wide = variant != 3 && a.KQ * 8 * 1024 <= 160 * 1024 && a.N * splits >= 128 * 192;
    if (a.mts <= 4) return wide ? lg_gemm_launch<1, 8, EPI>(a, st) : lg_gemm_launch<1, 4, EPI>(a, st);
    if (a.mts <= 8) return wide ? lg_gemm_launch<2, 8, EPI>(a, st) : lg_gemm_launch<2, 4, EPI>(a, st);
    return wide ? lg_gemm_launch<4, 8, EPI>(a, st) : lg_gemm_launch<4, 4, EPI>(a, st);
}

hipError_t lg_gemm(const float* img, int M, int K, const float* in, int ldb, float* out, int ldc, long long N, int epi,
                   const float* e, int lde, float* dout, int ldd, int act, hipStream_t st, const float* e2, const float* a3, int ld3,
                   int first) {
    if (N <= 0) return hipSuccess;
    LgGemmArgs a{};
    a.img = img; a.in = in; a.out = out; a.e = e; a.dout = dout; a.e2 = e2; a.a3 = a3; a.N = N;
    a.M = M; a.K = K; a.MT = (M + 15) / 16; a.KQ = (K + 15) / 16; a.ldb = ldb; a.ldc = ldc; a.lde = lde; a.ldd = ldd; a.ld3 = ld3;
    a.act = act; a.first = first;
    switch (epi) {
        case LG_EPI_ACT: return lg_gemm_dispatch<LG_EPI_ACT>(a, st);
        case LG_EPI_MUL: return lg_gemm_dispatch<LG_EPI_MUL>(a, st);
        case LG_EPI_MUL2: return lg_gemm_dispatch<LG_EPI_MUL2>(a, st);
        case LG_EPI_BOTTOM: return lg_gemm_dispatch<LG_EPI_BOTTOM>(a, st);
        case LG_EPI_SBAR: return lg_gemm_dispatch<LG_EPI_SBAR>(a, st);
        default: return lg_gemm_dispatch<LG_EPI_PLAIN>(a, st);
    }
}

// ---- weight cotangents: C[M x Nc] (column-major, ld = M) += X[M x samples] Y[Nc x samples]^T over one chunk of samples ----
struct LgWgradArgs {
    float* slabs;            // slab c at slabs + c * slab_stride; C at its start
    long long slab_stride;
    const float* x; const float* y;
    long long B, chunk;      // samples in total, samples per chunk (multiple of 4)
    int M, Nc, ldx, ldy;
    int nchunks, rblocks, groups;   // grid decomposition (see the kernel)
};

// NTN: 16-column tiles of C a wave keeps (Nc <= 16 NTN).  The four waves of a workgroup own four 16-row strips of C and share
// every slice of Y (and of X): 16 samples at a time are fetched with coalesced loads (each sample's rows are contiguous) into
// registers while the previous slice is being multiplied, then parked in LDS as [row tile][lane group g][row n][k-step u] so
// that ONE lane-linear ds_read_b128 hands a lane its B operands of all four k-steps of a tile (sample 4 u + g, row 16 t + n).
// (Reading the operands straight from global memory made every wave fetch all of Y: 670 MB of L2 traffic per call at cfg4.)
// (three waves per SIMD for every NTN: at NTN = 9 that build spills 48 registers to scratch - outside the steady-state loop -
// and is still the faster one: 246 us per 256 x 257 call against 262 us with two waves and no spill)
template <int NTN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
lg_wgrad_kernel(LgWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RY = (NTN * 16 + 63) / 64;                     // loads per lane per sample for the Y slice
    // floats per staged slice.  The Y part has room for all 64 RY fetched rows, not just the 16 NTN used ones: parking is then
    // unconditional - a predicated ds_write is a control-flow join, at which the compiler's wait-count insertion waits for
    // EVERY outstanding load (vmcnt(0)) and the two-slice prefetch shrinks to one
    constexpr int YS = RY * 1024, XS = 4 * 256;
    // slice buffer b: Y part at smem + b (YS + XS), X part behind it
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroup -> (sample chunk, row block, column group).  The row blocks and column groups of ONE chunk read the same
    // samples of X and Y; workgroups b and b + 8 share an XCD (and its L2) under the observed round-robin placement, so the
    // sharers of a chunk are given ids that are equal mod 8: X and Y then come from HBM once and from that L2 afterwards
    // (with the sharers dealt over the XCDs every one of them pulled its own copy: ~200 MB per call at cfg4 instead of 66).
    // Placement is a speed matter only; the result does not depend on it.
    const int sharers = a.rblocks * a.groups;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chunk = (slot / sharers) * 8 + xcd, sub = slot % sharers;
    if (chunk >= a.nchunks) return;
    const int rblock = sub % a.rblocks;
    const int row0 = rblock * 64;                                 // this workgroup's 64 rows of C; wave w: rows row0 + 16 w ..
    // column group: C wider than NTN tiles is covered by several workgroups, each with its own columns of Y
    const int col_lo = (sub / a.rblocks) * (NTN * 16);
    const int Nc = a.Nc - col_lo < NTN * 16 ? a.Nc - col_lo : NTN * 16;
    const float* __restrict__ Y = a.y + col_lo;
    const long long c0 = (long long)chunk * a.chunk;
    const long long c1 = c0 + a.chunk < a.B ? c0 + a.chunk : a.B;
    // The slab's current values are requested before the chunk's last slice is multiplied and added after it: the read half
    // of the read-modify-write is in flight during those MFMAs instead of being waited for after them (measured at cfg4:
    // 29 of 82 us per call went to the flush).
    // C rows 16 mt + 4 g + r (natural row order of the accumulator), column 16 t + n: four consecutive floats per lane.
    const int mt = rblock * 4 + wave;
    const int r0 = 16 * mt + 4 * g;
    float* C = a.slabs + (long long)chunk * a.slab_stride + (long long)col_lo * a.M;
    const bool rows4 = r0 + 3 < a.M;
    f32x4 prev[NTN];
    const int voff = n * a.M + r0;
    auto load_prev = [&]() {
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
            const int col = 16 * t + n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (col < Nc && 16 * mt < a.M) {
                const float* cp = (C + (long long)(16 * t) * a.M) + voff;   // uniform tile base + one per-lane offset
                if (rows4) { const f32x4u q = *reinterpret_cast<const f32x4u*>(cp); v = f32x4{q[0], q[1], q[2], q[3]}; }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r0 + r < a.M) v[r] = cp[r];
                }
            }
            prev[t] = v;
        }
    };
    f32x4 acc[NTN];
#pragma unroll
    for (int t = 0; t < NTN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A slice in flight: component u of yv[i] / xv is row (lane + 64 i) of sample 4 u + wave - the four k-steps one reader lane
    // wants, in the four registers of ONE 16-byte LDS write (lane-linear, conflict-free).  (Fetching by k-step instead - sample
    // 4 w + q - scattered them as 4-byte writes with an 8-way bank conflict.)
    struct Slice { f32x4 yv[RY]; f32x4 xv; };
    Slice slA, slB;
    // f32 MFMAs and VALU instructions share the issue slot, so the loop carries no per-lane address arithmetic: a sample's
    // column starts at a wave-uniform address (scalar registers, 32-bit offsets from the chunk's first sample), the lane's
    // row offsets are loop invariants, and rows past the operand's end are clamped instead of masked (they only reach rows /
    // columns of C that are never stored).
    unsigned yoff[RY];
#pragma unroll
    for (int i = 0; i < RY; ++i) { const int r = lane + 64 * i; yoff[i] = 4u * (unsigned)(r < Nc ? r : Nc - 1); }   // bytes
    const unsigned xoff = 4u * (unsigned)(row0 + lane < a.M ? row0 + lane : a.M - 1);
    const float* __restrict__ Yc = Y + c0 * (long long)a.ldy;     // wave-uniform bases of the chunk
    const float* __restrict__ Xc = a.x + c0 * (long long)a.ldx;
    const int nrem = (int)(c1 - c0);
    // (buffer loads: lane byte offset in a VGPR that never changes, the sample's byte offset in an SGPR - no address VALU.
    // The resources end exactly behind the chunk's last sample, so a slice that reaches past the chunk reads zeros there and
    // the fetch needs NO branch: a load inside an `if` makes a control-flow join at which the compiler's wait-count
    // insertion waits for everything outstanding - which would shorten the two-slice prefetch to one.)
    const __amdgpu_buffer_rsrc_t rY =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Yc), 0, (int)((((long long)nrem - 1) * a.ldy + Nc) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xc), 0, (int)((((long long)nrem - 1) * a.ldx + a.M) * 4), 0x00020000);
    auto at = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
    };
    auto fetch = [&](int s, Slice& sl) {                          // s: first sample of the slice, relative to c0
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned ys = 4u * (unsigned)(s + 4 * u + wave) * (unsigned)a.ldy;   // wave-uniform byte offsets
            const unsigned xs = 4u * (unsigned)(s + 4 * u + wave) * (unsigned)a.ldx;
#pragma unroll
            for (int i = 0; i < RY; ++i) sl.yv[i][u] = at(rY, yoff[i], ys);
            sl.xv[u] = at(rX, xoff, xs);
        }
    };
    auto park = [&](int buf, const Slice& sl) {
        f32x4* yb = reinterpret_cast<f32x4*>(smem + buf * (YS + XS));
        f32x4* xb = reinterpret_cast<f32x4*>(smem + buf * (YS + XS) + YS);
#pragma unroll
        for (int i = 0; i < RY; ++i) {
            const int r = lane + 64 * i;                            // row 16 t + nn of Y -> [t][g' = wave][nn][u]
            yb[((r >> 4) * 4 + wave) * 16 + (r & 15)] = sl.yv[i];   // (rows >= 16 NTN: fetched clamped, parked, never read)
        }
        xb[((lane >> 4) * 4 + wave) * 16 + (lane & 15)] = sl.xv;
    };
    // The last column group of a strip may have one tile less than NTN (257 columns = 9 + 8 tiles): its waves skip that
    // tile's MFMAs (wave-uniform branch) instead of multiplying padding - 1 / 18 of the 256 x 257 product.
    const bool last_tile = (Nc + 15) / 16 >= NTN;
    // A wave whose 16-row strip lies wholly behind the last row of C (136 rows = 9 strips in three row blocks: three of the last
    // block's four waves) fetches and parks its share of every slice like the others, but multiplies nothing: its SIMD's MFMA
    // pipe is then free for the waves of the other workgroups resident on it (the MFMAs it used to issue on clamped rows took
    // a quarter of the pipe time of a 136 x 137 product for results that were never stored).
    const bool strip_live = 16 * mt < a.M;
    auto multiply = [&](int buf) {
        if (!strip_live) return;
        const f32x4* y4 = reinterpret_cast<const f32x4*>(smem + buf * (YS + XS)) + lane;
        const f32x4 av = reinterpret_cast<const f32x4*>(smem + buf * (YS + XS) + YS)[wave * 64 + lane];
        f32x4 bv[NTN];
#pragma unroll
        for (int t = 0; t < NTN; ++t) bv[t] = y4[t * 64];
        // k-step outermost: consecutive MFMAs go to different accumulators (no dependent back-to-back pairs)
        constexpr int NTM = NTN > 1 ? NTN - 1 : NTN;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NTM; ++t) acc[t] = mfma4(av[u], bv[t][u], acc[t]);
        if constexpr (NTN > 1) {
            if (last_tile) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[NTN - 1] = mfma4(av[u], bv[NTN - 1][u], acc[NTN - 1]);
            }
        }
    };
    // Slices of 16 samples.  A slice is fetched TWO slices ahead of its use (two register sets, alternating) and parked in
    // the LDS buffer the previous multiply has just left: ~2 slice times of memory latency are covered.
    fetch(0, slA);
    park(0, slA);
    fetch(16, slA);
    __builtin_amdgcn_sched_barrier(0);   // set A's requests in front of set B's, as the loop leaves them (its wait counts rely on one order)
    fetch(32, slB);
    __syncthreads();
    // (The steady-state loop has ONE exit, at its top: with the end-of-chunk tests inside the body, the exit path from the
    // middle shared the latch block with the back edge, the wait-count insertion merged "A older than B" with "B older than A"
    // there, and every park of set A waited for vmcnt(0) - the two-slice prefetch was one slice deep on every other slice.)
    int s = 0;
    // (-DLG_EXP_NOFETCH / _NOPARK / _NOSYNC: timing-only builds of profiles/ubench/wgrad_variants.hip - wrong results)
#ifdef LG_EXP_NOFETCH
#define LG_FETCH(a, b) ((void)0)
#else
#define LG_FETCH(a, b) fetch(a, b)
#endif
#ifdef LG_EXP_NOPARK
#define LG_PARK(a, b) ((void)0)
#else
#define LG_PARK(a, b) park(a, b)
#endif
#ifdef LG_EXP_NOSYNC
#define LG_SYNC() ((void)0)
#else
#define LG_SYNC() __syncthreads()
#endif
    for (; s + 32 < nrem; s += 32) {
        multiply(0);                                  // slice s; set A holds s + 16, set B s + 32
        LG_PARK(1, slA);
        LG_FETCH(s + 48, slA);
        LG_SYNC();
        multiply(1);                                  // slice s + 16; set B holds s + 32, set A s + 48
        LG_PARK(0, slB);
        LG_FETCH(s + 64, slB);
        LG_SYNC();
    }
    if (s + 16 >= nrem) { load_prev(); multiply(0); }   // one slice left
    else {                                              // two
        multiply(0);
        park(1, slA);
        __syncthreads();
        load_prev();
        multiply(1);
    }
    if (16 * mt >= a.M) return;
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
        const int col = 16 * t + n;
        if (col >= Nc) continue;
        float* cp = (C + (long long)(16 * t) * a.M) + voff;
        const f32x4 v = prev[t] + acc[t];
        if (rows4) *reinterpret_cast<f32x4u*>(cp) = f32x4u{v[0], v[1], v[2], v[3]};
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r0 + r < a.M) cp[r] = v[r];
        }
    }
}

bool lg_wgrad_supported(int M, int Nc) { return M >= 1 && Nc >= 1 && Nc <= 16 * 9 * 65535; }

// number of sample chunks (= slabs) a wgrad call uses for this (M, B): enough workgroups to fill the chip, chunks of >= 64 samples
int lg_wgrad_chunks(int M, long long B, int num_cus, long long* chunk_out, int per_cu_dflt, int Nc) {
    const int RB = ((M + 15) / 16 + 3) / 4;
    if (Nc <= 0) Nc = M + 1;                           // a square layer's cotangent with its bias column
    const int groups = ((Nc + 15) / 16 + 8) / 9;      // column groups (lg_wgrad)
    // workgroups per CU: 2 for the per-stage calls of the layer-wise path (B columns), 4 for the per-step calls of the
    // cooperative gradient (2 x stages x B columns: cfg4 loss + gradient 139.9 -> 133.3 ms; 1: 162, 3: 135, 8: 135)
    const int per_cu_env = tuning().lg_wgrad_per_cu;
    // (three workgroups are resident per CU.  The per-step calls' workgroups per CU grow with the number of workgroups that
    // share a chunk: their costs differ - dead strips, a short last column group - and more, smaller chunks even the CUs out.
    // Loss + gradient of the default architecture, B = 32 768, against 4 per CU for all: nvariables = 16: 50.1 -> 47.7 ms,
    // 20: 63.1 -> 59.9, 28: 100.6 -> 96.5, 32: 152.9 -> 146.0, 40: 218.2 -> 206.0; profiles/r4/r4u_wgrad_chunking.txt)
    const int t1 = tuning().lg_wgrad_t1, t2 = tuning().lg_wgrad_t2;
    const bool uneven = M % 64 != 0;   // (256 x 257 has no dead strip: 6 and 12 tie there, 108.3 against 108.5 ms at cfg4)
    // (the layer-wise path's per-stage calls, 2 per CU: 6 from 12 sharers on - nvariables = 48: 645 -> 627 ms, 56: 780 -> 735)
    const int by_rule = per_cu_dflt == 2 ? (RB * groups >= 12 ? 6 : 2)
                      : per_cu_dflt != 4 ? per_cu_dflt : (RB * groups >= t2 + (uneven ? 0 : 1)) ? 12 : RB * groups >= t1 ? 6 : 3;
    const int per_cu = per_cu_env > 0 ? per_cu_env : by_rule;
    long long want = ((long long)per_cu * num_cus + RB * groups - 1) / (RB * groups);   // workgroups per CU
    if (want < 1) want = 1;
    long long chunk = (B + want - 1) / want;
    chunk = (chunk + 63) / 64 * 64;
    if (chunk < 64) chunk = 64;
    const long long nch = (B + chunk - 1) / chunk;
    if (chunk_out) *chunk_out = chunk;
    return (int)(nch < 1 ? 1 : nch);
}

hipError_t lg_wgrad(float* slabs, long long slab_stride, long long chunk, int nchunks, int M, int Nc, const float* x, int ldx,
                    const float* y, int ldy, long long B, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    LgWgradArgs a{};
    a.slabs = slabs; a.slab_stride = slab_stride; a.x = x; a.y = y; a.B = B; a.chunk = chunk;
    a.M = M; a.Nc = Nc; a.ldx = ldx; a.ldy = ldy;
    const int MT = (M + 15) / 16, NTN = (Nc + 15) / 16;
    // strips wider than 9 tiles go to several workgroups (column groups of 9 tiles): 36 accumulator registers leave room for
    // the two-slice prefetch and three waves per SIMD; a 17-tile strip has neither (measured at cfg4: 67 us vs NN us per call)
    const int groups = (NTN + 8) / 9;
    a.nchunks = nchunks; a.rblocks = (MT + 3) / 4; a.groups = groups;
    const dim3 grid((unsigned)(((nchunks + 7) / 8) * 8 * a.rblocks * a.groups));
    // tiles per column group: the strip's tiles dealt evenly over its groups (11 tiles = 6 + 5, 13 = 7 + 6, 21 = 3 x 7), on
    // the instance of exactly that many tiles - with the 9-tile instance for every group a 168 x 169 cotangent multiplied
    // 18 tiles for its 11 (the short last group skips the one tile it lacks)
    const int ntn = NTN <= 3 ? (NTN <= 1 ? 1 : 3) : (NTN + groups - 1) / groups;
    const int lds = 2 * ((ntn * 16 + 63) / 64 * 1024 + 4 * 256) * (int)sizeof(float);   // <= 2 * 16 KB
    switch (ntn) {
        case 1: hipLaunchKernelGGL(lg_wgrad_kernel<1>, grid, dim3(256), lds, st, a); break;
        case 3: hipLaunchKernelGGL(lg_wgrad_kernel<3>, grid, dim3(256), lds, st, a); break;
        case 4: hipLaunchKernelGGL(lg_wgrad_kernel<4>, grid, dim3(256), lds, st, a); break;
        case 5: hipLaunchKernelGGL(lg_wgrad_kernel<5>, grid, dim3(256), lds, st, a); break;
        case 6: hipLaunchKernelGGL(lg_wgrad_kernel<6>, grid, dim3(256), lds, st, a); break;
        case 7: hipLaunchKernelGGL(lg_wgrad_kernel<7>, grid, dim3(256), lds, st, a); break;
        case 8: hipLaunchKernelGGL(lg_wgrad_kernel<8>, grid, dim3(256), lds, st, a); break;
        default: hipLaunchKernelGGL(lg_wgrad_kernel<9>, grid, dim3(256), lds, st, a); break;
    }
    return hipGetLastError();
}

}  // namespace cnf
