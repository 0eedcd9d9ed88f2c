// cnf_mfma_kernel.h — the per-wave fused solve kernel template (see cnf_mfma.hip for the design
// notes) and its instantiation record.  Included by the translation units that instantiate it.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "cnf_mfma_dev.h"

namespace cnf {

// Out[MT tiles] += A(image at smem+img)[MT x KS k-steps] * In   where In register s is k-step s.
// `in` is indexed [s]; KS live k-steps (KG = ceil(KS/4) groups in the image).
template <int MT, int KS, typename InT>
__device__ __forceinline__ void gemm_tiles(const float* __restrict__ img, int lane, const InT& in,
                                           f32x4 (&acc)[MT]) {
    constexpr int KG = (KS + 3) / 4;
    const f32x4* A = reinterpret_cast<const f32x4*>(img) + lane;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        f32x4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = A[(mt * KG + kg) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (kg * 4 + j < KS) {
                const float b = in(kg * 4 + j);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma4(a[mt][j], b, acc[mt]);
            }
        }
    }
}

template <int N>
struct RegIn {  // flat register array as k-step source
    const float (&v)[N];
    __device__ __forceinline__ float operator()(int s) const { return v[s]; }
};
template <int MT>
struct TileIn {  // accumulator tiles as k-step source: k-step s = tile s/4, register s%4
    const f32x4 (&v)[MT];
    __device__ __forceinline__ float operator()(int s) const { return v[s >> 2][s & 3]; }
};



// ---------------------------------------------------------------------------------------
// split-bf16 hidden product (ARITH = 1): Out[HT tiles] += W (H x H) * In, with both operands split
// exactly into three bf16 parts (x = hi + mid + lo, 8 + 8 + 8 significant bits) and six bf16
// MFMAs per (tile, 32-deep k-chunk): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi — every term
// down to 2^-24 relative is kept, accumulation stays f32, so the result is f32-equivalent
// (simulation + parity tests), at 16x the f32 MFMA rate per instruction.
// v_mfma_f32_16x16x32_bf16: lane l holds A[row l&15][k = 8 (l>>4) + j] and B[k = 8 (l>>4) + j][col l&15],
// j = 0..7; C/D as the f32 form.  The 8 k-values lane group g supplies for chunk c are the two
// accumulator tiles 2c, 2c+1 (4 registers each): k-slot j <-> feature 16 (2c + (j>>2)) + 4 (j&3) + g;
// the image packer (mfma_pack) stores W columns in that order, so chaining needs no data movement.
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // lo16 = bf16(a), hi16 = bf16(b), RNE
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// x[0..7] -> three packed bf16x8 fragments
__device__ __forceinline__ void split3_bf16(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        const unsigned h = cvt_pk_bf16(x0, x1);
        // (the residuals as packed subtractions - v_pk_add_f32 with neg - save 49 VALU issues per call but run 1.7 % slower
        // in a same-box A/B: packed f32 ops are an anti-lever beside bf16 MFMAs, as MI355X_MICROARCH.md says)
        const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
        const unsigned m = cvt_pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
        hi[i] = h; mid[i] = m; lo[i] = cvt_pk_bf16(s0, s1);
    }
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int HT>
__device__ __forceinline__ void gemm_hidden_bf16x6(const float* __restrict__ img, int lane,
                                                   const f32x4 (&in)[HT], f32x4 (&acc)[HT]) {
    static_assert(HT % 2 == 0, "split-bf16 hidden products need an even number of 16-row tiles");
    constexpr int NC = HT / 2;
    const u32x4* __restrict__ A = reinterpret_cast<const u32x4*>(img) + lane;   // [split][mt][chunk][lane]
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = in[2 * c][j]; x[4 + j] = in[2 * c + 1][j]; }
        u32x4 bh, bm, bl;
        split3_bf16(x, bh, bm, bl);
        u32x4 ah[HT], am[HT], al[HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
            ah[mt] = A[((0 * HT + mt) * NC + c) * 64];
            am[mt] = A[((1 * HT + mt) * NC + c) * 64];
            al[mt] = A[((2 * HT + mt) * NC + c) * 64];
        }
        // small terms first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) acc[mt] = mfma_bf16(al[mt], bh, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) acc[mt] = mfma_bf16(ah[mt], bl, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) acc[mt] = mfma_bf16(am[mt], bm, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) acc[mt] = mfma_bf16(am[mt], bh, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) acc[mt] = mfma_bf16(ah[mt], bm, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) acc[mt] = mfma_bf16(ah[mt], bh, acc[mt]);
    }
}

// hidden H x H product in the arithmetic of the instance
template <int HT, int ARITH>
__device__ __forceinline__ void gemm_hidden(const float* __restrict__ img, int lane, const f32x4 (&in)[HT],
                                            f32x4 (&acc)[HT]) {
    if constexpr (ARITH == 1) gemm_hidden_bf16x6<HT>(img, lane, in, acc);
    else gemm_tiles<HT, 4 * HT>(img, lane, TileIn<HT>{in}, acc);
}

// Activation of one accumulator tile.  For tanh the non-transcendental steps are written on 2-wide
// vectors so they select the packed VALU forms (v_pk_add_f32 / v_pk_fma_f32: two lanes' worth per issue
// slot, same issue cost as the scalar form — profiles/ubench/pk_f32_cost.result.txt); on gfx950 f32 VALU
// time adds to f32 MFMA time, so every issue slot saved is wall time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int ACT>
__device__ __forceinline__ void act_tile(const f32x4& a, f32x4& h, f32x4& d) {
    if constexpr (ACT == CNF_ACT_TANH_PRESCALED || ACT == CNF_ACT_TANH) {
        f32x2 x0 = {a[0], a[1]}, x1 = {a[2], a[3]};
        if constexpr (ACT == CNF_ACT_TANH) { x0 = x0 * kTanhPrescale; x1 = x1 * kTanhPrescale; }
        f32x2 e0 = {__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1])};
        f32x2 e1 = {__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1])};
        e0 = e0 + 1.f;
        e1 = e1 + 1.f;
        const f32x2 r0 = {fast_rcp(e0[0]), fast_rcp(e0[1])}, r1 = {fast_rcp(e1[0]), fast_rcp(e1[1])};
        const f32x2 two = {2.f, 2.f}, one = {1.f, 1.f};
        // (Do not name the packed instructions in inline asm here: the compiler inserts the MFMA -> VALU wait states
        // (s_nop) only for instructions it selected itself, and an asm statement reading an accumulator tile
        // right after the MFMAs returns garbage.  Measured anyway on values that were safe: no gain.)
        const f32x2 h0 = __builtin_elementwise_fma(r0, two, -one), h1 = __builtin_elementwise_fma(r1, two, -one);
        const f32x2 d0 = __builtin_elementwise_fma(-h0, h0, one), d1 = __builtin_elementwise_fma(-h1, h1, one);
        h = f32x4{h0[0], h0[1], h1[0], h1[1]};
        d = f32x4{d0[0], d0[1], d1[0], d1[1]};
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float dd;
            h[r] = act_fwd<ACT>(a[r], dd);
            d[r] = dd;
        }
    }
}

// One dynamics evaluation for a 16-sample tile.
//   forward pass (shared), then
//   ENG_VJP: pullback of KP probes with the transposed images  (g = eps^T J;  src/core/utils.jl:150-159)
//   ENG_TAN: pushforward of tangents with the forward images only:
//            Hutchinson JVP (g = J eps; src/core/utils.jl:161-170) or, with `exact`, the D unit
//            tangents whose i-th output row is J_ii (trace of src/core/utils.jl:79-88, icnf.jl:312)
template <int HT, int L, int ZR, int CR, int ACT, int ENGINE, int KP, int PRE, int ARITH>
__device__ __forceinline__ void dyn_eval(const float* __restrict__ smem, const float* __restrict__ gimg, int lane, float t,
                                         bool autonomous, bool reg_z, bool reg_j, bool exact, int D, int K,
                                         const float (&z)[ZR], const float (&y)[CR > 0 ? CR : 1],
                                         const float (&eps)[KP][ZR], const f32x4 (&pre_c)[HT],
                                         const f32x4 (&pre_q)[HT], float (&zd)[ZR], float& ld,
                                         float& ed, float& nd) {
    constexpr MfmaLayout LAY(HT, L, ZR, CR, ENGINE == ENG_VJP, ARITH);
    constexpr int DT = (ZR + 3) / 4;
    const int g = lane >> 4;
    f32x4 h[HT];
    f32x4 d[L][HT];  // act' of every hidden layer, kept for the pullback / pushforward

    // ---- layer 1: a = W1z z + w1t t + W1y y + b1 ----
    {
        f32x4 acc[HT];
        load_cvec<HT>(smem + LAY.v_b1, g, acc);
        if (!autonomous) {
            f32x4 wt[HT];
            load_cvec<HT>(smem + LAY.v_w1t, g, wt);
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) acc[mt] += wt[mt] * t;
        }
        gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{z}, acc);
        if constexpr (CR > 0) gemm_tiles<HT, CR>(smem + LAY.f1y, lane, RegIn<CR>{y}, acc);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) act_tile<ACT>(acc[mt], h[mt], d[0][mt]);
    }
    // ---- hidden layers 2..L ----
#pragma unroll
    for (int l = 1; l < L; ++l) {
        f32x4 acc[HT];
        load_cvec<HT>(smem + LAY.v_bh + (l - 1) * MfmaLayout::vecC(HT), g, acc);
        gemm_hidden<HT, ARITH>(smem + LAY.fh + (l - 1) * LAY.imgHid(), lane, h, acc);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) act_tile<ACT>(acc[mt], h[mt], d[l][mt]);
    }
    // ---- last layer (identity): zdot ----
    {
        f32x4 acc[DT];
        load_cvec<DT>(smem + LAY.v_bN, g, acc);
        gemm_tiles<DT, 4 * HT>((LAY.fN_global ? gimg : smem) + LAY.fN, lane, TileIn<HT>{h}, acc);
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = acc[s >> 2][s & 3];
    }
    ed = 0.f;
    if (reg_z) {
        float e2 = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) e2 = fmaf(zd[s], zd[s], e2);
        ed = sqrtf(group_sum(e2));   // Edot = |zdot|_2   (src/core/icnf.jl:184-199)
    }
    ld = 0.f;
    nd = 0.f;
    // Probes / tangents run one after the other (rolled loop): the operand images are re-read
    // from LDS each time, which costs LDS bandwidth the kernel has to spare and keeps the
    // register footprint flat.
    // K <= KP probes are live (KP is the instance's register capacity; K == KP == 1 for the hoisting instances)
    int nseed = (ENGINE == ENG_TAN && exact) ? D : K;
    if constexpr (ENGINE == ENG_TAN && L == 2) {
        if (LAY.qtr >= 0 && exact) {
            // Two hidden layers: tr J = sum_ab act'_2[a] W_2[a][b] act'_1[b] (W_1[:,0:D] W_3)[b][a] = act'_2^T Q act'_1 with the
            // constant Q = W_2 .* (W_1[:,0:D] W_3)^T packed beside the weights: ONE H x H product and a dot, instead of D
            // tangent passes (the batched-Jacobian trace of src/core/utils.jl:79-88, icnf.jl:312, for the default architecture).
            f32x4 qd[HT];
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) qd[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_tiles<HT, 4 * HT>(smem + LAY.qtr, lane, TileIn<HT>{d[0]}, qd);
            float tr = 0.f;
#pragma unroll
            for (int mt = 0; mt < HT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) tr = fmaf(qd[mt][r], d[1][mt][r], tr);
            ld = -group_sum(tr);
            nseed = 0;
        }
    }
    const float scale = (ENGINE == ENG_TAN && exact) ? 1.f : 1.f / (float)K;
#pragma clang loop unroll(disable)
    for (int p = 0; p < nseed; ++p) {
        float ep[ZR];
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            ep[s] = eps[0][s];
#pragma unroll
            for (int q = 1; q < KP; ++q) ep[s] = (p == q) ? eps[q][s] : ep[s];
            if (ENGINE == ENG_TAN && exact) ep[s] = (4 * s + g == p) ? 1.f : 0.f;   // unit vector e_p
        }
        int opq = 0;
        if (KP > 1 || ENGINE == ENG_TAN) asm volatile("" : "+v"(opq));
        const float* __restrict__ sm = smem + opq;
        f32x4 gacc[DT];
#pragma unroll
        for (int dt_ = 0; dt_ < DT; ++dt_) gacc[dt_] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (ENGINE == ENG_VJP) {
            f32x4 dl[HT];
            if constexpr (PRE) {   // W_N^T eps does not change during the solve: hoisted by the caller
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) dl[mt] = pre_c[mt] * d[L - 1][mt];
            } else {
                f32x4 acc[HT];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_tiles<HT, ZR>(sm + LAY.bN, lane, RegIn<ZR>{ep}, acc);   // W_N^T eps
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) dl[mt] = acc[mt] * d[L - 1][mt];
            }
#pragma unroll
            for (int l = L - 1; l >= 1; --l) {  // W_{l+1}^T delta, times act'(a_l)
                f32x4 acc[HT];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_hidden<HT, ARITH>(sm + LAY.bh + (l - 1) * LAY.imgHid(), lane, dl, acc);
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) dl[mt] = acc[mt] * d[l - 1][mt];
            }
            if constexpr (PRE == 2) {
                // without the |eps^T J| regulariser only <eps^T J, eps> = <delta_1, W_1[:,0:D] eps> is
                // needed: a dot product with the hoisted q = W_1[:,0:D] eps replaces the last product
                float qd = 0.f;
#pragma unroll
                for (int mt = 0; mt < HT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) qd = fmaf(dl[mt][r], pre_q[mt][r], qd);
                ld -= scale * group_sum(qd);
                continue;
            }
            gemm_tiles<DT, 4 * HT>(sm + LAY.b1, lane, TileIn<HT>{dl}, gacc);   // W_1[:,0:D]^T delta_1
        } else {
            f32x4 tau[HT];
            {
                f32x4 acc[HT];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (LAY.v_w1c >= 0 && exact) load_cvec<HT>(sm + LAY.v_w1c + p * MfmaLayout::vecC(HT), g, acc);   // W_1[:, p]: no product
                else gemm_tiles<HT, ZR>(sm + LAY.f1z, lane, RegIn<ZR>{ep}, acc);  // W_1[:,0:D] v
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) tau[mt] = acc[mt] * d[0][mt];
            }
#pragma unroll
            for (int l = 1; l < L; ++l) {  // act'(a_{l+1}) .* (W_{l+1} tau)
                f32x4 acc[HT];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_hidden<HT, ARITH>(sm + LAY.fh + (l - 1) * LAY.imgHid(), lane, tau, acc);
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) tau[mt] = acc[mt] * d[l][mt];
            }
            if (LAY.v_wNr >= 0 && exact) {
                // only J_pp = <W_N[p, :], tau> is needed: a dot with row p of W_N (kept in accumulator layout in the
                // image) instead of a full 16-row last-layer product: DT * 4 HT MFMAs -> 4 HT fmas + one reduction
                f32x4 wr[HT];
                load_cvec<HT>(sm + LAY.v_wNr + p * MfmaLayout::vecC(HT), g, wr);
                float jd = 0.f;
#pragma unroll
                for (int mt = 0; mt < HT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) jd = fmaf(tau[mt][r], wr[mt][r], jd);
                ld -= group_sum(jd);
                continue;
            }
            gemm_tiles<DT, 4 * HT>((LAY.fN_global ? gimg + opq : sm) + LAY.fN, lane, TileIn<HT>{tau}, gacc);  // W_N tau = J v
        }
        float dot = 0.f, n2 = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const float gv = gacc[s >> 2][s & 3];
            dot = fmaf(gv, ep[s], dot);     // ldot = -sum(eJ .* eps) / -sum(eps .* Jeps) / -J_pp
            n2 = fmaf(gv, gv, n2);
        }
        ld -= scale * group_sum(dot);
        if (reg_j) nd += scale * sqrtf(group_sum(n2));   // ndot = |eps^T J|_2 or |J eps|_2 (icnf.jl:229-245)
    }
}

// PRE: 0 none; 1 hoist c = W_N^T eps; 2 also hoist q = W_1[:,0:D] eps and skip the last pullback
// product (valid only without reg_j).  PRE > 0 needs ENGINE == ENG_VJP and KP == 1.
template <int HT, int L, int ZR, int CR, int ACT, int ENGINE, int KP, int PRE, int NTHREADS, int ARITH = 0>
__global__ void __launch_bounds__(NTHREADS)
mfma_solve_kernel(KArgs a) {
    constexpr MfmaLayout LAY(HT, L, ZR, CR, ENGINE == ENG_VJP, ARITH);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // stage the packed weight image: global (L2) -> LDS, 16 B per lane, coalesced
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.packed);
        f32x4* dst = reinterpret_cast<f32x4*>(smem);
        for (int i = threadIdx.x; i < LAY.lds_total / 4; i += NTHREADS) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, n = lane & 15;
    const int wave = threadIdx.x >> 6;
    constexpr int WPB = NTHREADS / 64;
    const long long ntiles = (a.B + 15) / 16;
    const int D = a.D, S = D + 3, C = a.C;
    const int K = KP == 1 ? 1 : a.K;   // live probes (<= KP)
    const int Kd = K * D;
    const bool reg_z = a.reg_z, reg_j = a.reg_j, autonomous = a.autonomous, exact = a.exact;

    // SIMD partners (waves w and w+4 of a workgroup) run the same program; left alone they fall
    // into lockstep, their activation phases coincide and the matrix pipe idles (measured: 72 %
    // MFMA-busy).  A static priority split makes one partner the pole wave and lets the other
    // fill its VALU phases; the dynamic tile queue then balances the uneven progress.
    if (a.prio_mode == 1) { if (__builtin_amdgcn_readfirstlane(wave) < 4) __builtin_amdgcn_s_setprio(1); }
    else if (a.prio_mode == 2) { if (__builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1); }
    else if (a.prio_mode == 3) {   // graded: wave quad 0 highest
        const int q = __builtin_amdgcn_readfirstlane(wave) >> 2;
        if (q == 0) __builtin_amdgcn_s_setprio(3);
        else if (q == 1) __builtin_amdgcn_s_setprio(2);
        else if (q == 2) __builtin_amdgcn_s_setprio(1);
    }
    const long long total_waves = (long long)gridDim.x * WPB;

    for (long long tile = (long long)blockIdx.x * WPB + wave; tile < ntiles;) {
        // next tile: static stride, or one returning atomic per tile on the launch's queue word
        long long next_tile = tile + total_waves;
        if (a.queue) {
            int tk = 0;
            if (lane == 0) tk = atomicAdd(a.queue, 1);
            next_tile = total_waves + (long long)__builtin_amdgcn_readfirstlane(tk);
        }
        const long long smp = tile * 16 + n;
        const bool valid = smp < a.B;
        const long long sc = valid ? smp : a.B - 1;   // clamp loads, mask stores
        float z[ZR], eps[KP][ZR], y[CR > 0 ? CR : 1];
        float lacc = 0.f, eacc = 0.f, nacc = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            if (a.x) z[s] = f < a.nvars ? a.x[sc * a.nvars + f] : 0.f;   // u0 = [x; 0]
            else z[s] = f < D ? a.u0[sc * S + f] : 0.f;
#pragma unroll
            for (int p = 0; p < KP; ++p) eps[p][s] = (f < D && p < K && a.eps) ? a.eps[sc * Kd + p * D + f] : 0.f;
        }
        if (!a.x) { lacc = a.u0[sc * S + D]; eacc = a.u0[sc * S + D + 1]; nacc = a.u0[sc * S + D + 2]; }
        y[0] = 0.f;
        if constexpr (CR > 0) {
#pragma unroll
            for (int s = 0; s < CR; ++s) { const int f = 4 * s + g; y[s] = f < C ? a.ys[sc * C + f] : 0.f; }
        }

        // ---- fixed-step explicit RK, stage loop rolled (one copy of the dynamics code) ----
        float kz[6][ZR], kl[6], ke[6], kn[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            kl[j] = ke[j] = kn[j] = 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) kz[j][s] = 0.f;
        }
        f32x4 pre_c[HT], pre_q[HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) pre_c[mt] = pre_q[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE >= 1) gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps[0]}, pre_c);
        if constexpr (PRE >= 2) {
            gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{eps[0]}, pre_q);
            if constexpr (ACT == CNF_ACT_TANH_PRESCALED) {   // the forward image carries the tanh pre-scale
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) pre_q[mt] *= (1.f / kTanhPrescale);
            }
        }
        const float dt = a.dt;
        const bool single = a.nsteps == 0;      // one dynamics call: du = f(u, p, t0)
        const int ns = single ? 1 : a.T.ns;
        const int nsteps = single ? 1 : a.nsteps;
#pragma clang loop unroll(disable)
        for (int step = 0; step < nsteps; ++step) {
            const float tn = a.t0 + (float)step * dt;
            if (a.ckpt) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)step * ntiles + tile) * 64 + lane) * ZR + s] = z[s];
            }
#pragma clang loop unroll(disable)
            for (int st = 0; st < ns; ++st) {
                float zs[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc = fmaf(a.T.a[st][j], kz[j][s], acc);
                    zs[s] = fmaf(dt, acc, z[s]);
                }
                float zd[ZR], ld, ed, nd;
                // The weight image is loop-invariant, and LLVM would hoist all ~300 operand reads out
                // of the RK loops (and spill them).  An opaque zero offset pins the reads per stage.
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                dyn_eval<HT, L, ZR, CR, ACT, ENGINE, KP, PRE, ARITH>(smem + opaque, a.packed + opaque, lane, tn + a.T.c[st] * dt, autonomous,
                                                              reg_z, reg_j, exact, D, K, zs, y, eps, pre_c, pre_q, zd,
                                                              ld, ed, nd);
                if (a.ckpt_k) {
#pragma unroll
                    for (int s = 0; s < ZR; ++s)
                        a.ckpt_k[((((long long)step * ns + st) * ntiles + tile) * 64 + lane) * ZR + s] = zd[s];
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const bool hit = (j == st);
                    kl[j] = hit ? ld : kl[j];
                    ke[j] = hit ? ed : ke[j];
                    kn[j] = hit ? nd : kn[j];
#pragma unroll
                    for (int s = 0; s < ZR; ++s) kz[j][s] = hit ? zd[s] : kz[j][s];
                }
            }
            if (single) break;
#ifndef CNF_NO_KFULL   // (A/B switch: the cost of this block on the metric kernel is measured with -DCNF_NO_KFULL)
            if (a.kfull && valid) {
                // all S rows of the step's stage derivatives in the ABI's layout, [stage][sample][row]: the embedded
                // error estimate of an adaptive attempt (cnf_step_embedded, nsteps = 1).  Once per step, outside
                // the stage loop, so the metric path pays one untaken branch per step.
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    if (j < ns) {
                        float* kf = a.kfull + ((long long)j * a.B + smp) * S;
#pragma unroll
                        for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) kf[f] = kz[j][s]; }
                        if (g == 0) { kf[D] = kl[j]; kf[D + 1] = ke[j]; kf[D + 2] = kn[j]; }
                    }
                }
            }
#endif
            float sl = 0.f, se = 0.f, sn = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float bj = a.T.b[j];
                sl = fmaf(bj, kl[j], sl); se = fmaf(bj, ke[j], se); sn = fmaf(bj, kn[j], sn);
            }
            lacc = fmaf(dt, sl, lacc); eacc = fmaf(dt, se, eacc); nacc = fmaf(dt, sn, nacc);
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 6; ++j) acc = fmaf(a.T.b[j], kz[j][s], acc);
                z[s] = fmaf(dt, acc, z[s]);
            }
        }

        if (single) {
            if (valid) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) a.u_out[smp * S + f] = kz[0][s]; }
                if (g == 0) { a.u_out[smp * S + D] = kl[0]; a.u_out[smp * S + D + 1] = ke[0]; a.u_out[smp * S + D + 2] = kn[0]; }
            }
            tile = next_tile;
            continue;
        }
        if (a.ckpt) {
#pragma unroll
            for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)a.nsteps * ntiles + tile) * 64 + lane) * ZR + s] = z[s];
        }
        // ---- epilogue: inference_sol (src/core/base_icnf.jl:158-172) ----
        float ss = 0.f, sa = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            const float v2 = z[s] * z[s];
            ss += v2;
            if (f >= a.nvars) sa += v2;
        }
        ss = group_sum(ss);
        sa = group_sum(sa);
        if (valid) {
            if (a.u_out) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) a.u_out[smp * S + f] = z[s]; }
                if (g == 0) { a.u_out[smp * S + D] = lacc; a.u_out[smp * S + D + 1] = eacc; a.u_out[smp * S + D + 2] = nacc; }
            }
            if (g == 0) {
                if (a.logp) a.logp[smp] = (-0.5f * (float)D * kLog2Pi - 0.5f * ss) - lacc;
                if (a.regs) {
                    a.regs[smp] = eacc;
                    a.regs[a.B + smp] = nacc;
                    a.regs[2 * a.B + smp] = a.reg_aug ? sqrtf(sa) : 0.f;
                }
            }
        }
        tile = next_tile;
    }
}

typedef hipError_t (*LaunchFn)(const KArgs&, int lds_bytes, int nblocks, hipStream_t);

template <int HT, int L, int ZR, int CR, int ACT, int ENGINE, int KP, int PRE, int NTHREADS, int ARITH = 0>
inline hipError_t launch_inst(const KArgs& a, int lds_bytes, int nblocks, hipStream_t st) {
    auto kern = mfma_solve_kernel<HT, L, ZR, CR, ACT, ENGINE, KP, PRE, NTHREADS, ARITH>;
    // > 64 KB of dynamic LDS has to be enabled once per device for this kernel
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (!once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NTHREADS), lds_bytes, st, a);
    return hipGetLastError();
}

struct Inst {
    int HT, L, ZR, CR, ACT, ENGINE, KP;
    int PRE;   // 2 requires !reg_j
    LaunchFn fn;
    int nthreads;
    int arith; // CNF_ARITH_*
};

#define MFMA_INST(HT, L, ZR, CR, ACT, ENG, KP, PRE, NT) \
    Inst { HT, L, ZR, CR, ACT, ENG, KP, PRE, &launch_inst<HT, L, ZR, CR, ACT, ENG, KP, PRE, NT>, NT, 0 }
#define MFMA_INST_BF16X6(HT, L, ZR, CR, ACT, ENG, KP, PRE, NT) \
    Inst { HT, L, ZR, CR, ACT, ENG, KP, PRE, &launch_inst<HT, L, ZR, CR, ACT, ENG, KP, PRE, NT, 1>, NT, 1 }

// generic zero-padded instances (cnf_mfma_generic.hip): D <= 16 and C <= 16 or C = 0
const Inst* mfma_generic_insts(int* count);
// the same shapes with a capacity of several Hutchinson probes (cnf_mfma_generic_probes.hip): VJP, K <= KP
const Inst* mfma_generic_probe_insts(int* count);
// state k-steps padded to 8 (D <= 32): the reference's default nets for nvariables 8..15 (cnf_mfma_generic_zr8.hip)
const Inst* mfma_generic_zr8_insts(int* count);


}  // namespace cnf
