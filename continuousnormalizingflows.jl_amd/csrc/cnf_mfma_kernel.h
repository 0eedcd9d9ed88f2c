// cnf_mfma_kernel.h — the per-wave fused solve kernel template (see cnf_mfma.hip for the design
// notes) and its instantiation record.  Included by the translation units that instantiate it.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <type_traits>

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

// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N-1.  Unlike `#pragma unroll`, every index is a constant
// when the IR is generated, so arrays indexed by it are split into registers by the first SROA run (a pragma-unrolled loop
// over pre_c[p] left the array in scratch: 13 scratch loads per dynamics call).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
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

// Packed f32 VALU (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32: two values per issue slot at the scalar forms' issue cost,
// profiles/ubench/pk_f32_cost.result.txt).  On gfx950 f32 VALU time ADDS to f32 MFMA time, so every issue slot is wall time.
// The compiler cannot be trusted with them here: LLVM's pre-emit peephole UNPACKS packed f32 instructions that sit behind an
// MFMA (43 of the 97 in this kernel: "co-issue in the MFMA shadow" - which v_mfma_f32_16x16x4_f32 does not offer), so the hot
// ones are written as inline asm.  The hazard recogniser does not look inside asm, so each statement carries its own wait
// states (MI300/MI350 ISA, "manually inserted wait states"):
//   * trans (v_exp / v_rcp) result read by a non-trans VALU: 1 wait state            -> leading s_nop 0
//   * MFMA (8-pass) result read by VALU, or its SrcC overwritten by VALU: 11         -> leading s_nop 7 + s_nop 3 (PK_AFTER_MFMA)
//   * VALU result read by an MFMA as SrcA/B: the asm result is consumed by compiler-selected instructions that do not know
//     the asm is a VALU                                                              -> trailing s_nop 1 (PK_BEFORE_MFMA)
// -DCNF_NO_PK_ASM falls back to the plain vector expressions (A/B switch).
#ifndef CNF_NO_PK_ASM
__device__ __forceinline__ void pk_add1(f32x2& a, f32x2& b) {          // a += 1, b += 1; inputs written by v_exp_f32
    asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %0, 1.0 op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %1, 1.0 op_sel_hi:[1,0]" : "+v"(a), "+v"(b));
}
// h = 2 r - 1, d = 1 - h^2 for two value pairs; r written by v_rcp_f32 (each d follows the OTHER pair's h: no back-to-back dependence)
__device__ __forceinline__ void pk_tanh_from_r(f32x2& ha, f32x2& hb, f32x2& da, f32x2& db, const f32x2& ra, const f32x2& rb) {
    asm volatile("s_nop 0\n\t"
                 "v_pk_fma_f32 %0, %4, 2.0, -1.0 op_sel_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 %1, %5, 2.0, -1.0 op_sel_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 %2, %0, %0, 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 %3, %1, %1, 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                 : "=&v"(ha), "=&v"(hb), "=&v"(da), "=&v"(db) : "v"(ra), "v"(rb));
}
// c[i] = a[i] .* b[i] for N accumulator tiles in ONE statement (one set of wait states per layer, not per tile);
// `a` may be MFMA results, c feeds MFMAs
#define CNF_PKMUL(o, x, y) "v_pk_mul_f32 %" #o ", %" #x ", %" #y "\n\t"
template <int N>
__device__ __forceinline__ void tiles_mul_block(const f32x4* a, const f32x4* b, f32x4* c) {
    static_assert(N >= 1 && N <= 4, "at most 4 tiles (24 asm operands) per statement");
    f32x2 x[2 * N], y[2 * N], o[2 * N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x[2 * i] = f32x2{a[i][0], a[i][1]}; x[2 * i + 1] = f32x2{a[i][2], a[i][3]};
        y[2 * i] = f32x2{b[i][0], b[i][1]}; y[2 * i + 1] = f32x2{b[i][2], b[i][3]};
    }
    if constexpr (N == 1) {
        asm volatile("s_nop 7\n\ts_nop 3\n\t" CNF_PKMUL(0, 2, 4) CNF_PKMUL(1, 3, 5) "s_nop 1"
            : "=&v"(o[0]), "=&v"(o[1]) : "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]));
    } else if constexpr (N == 2) {
        asm volatile("s_nop 7\n\ts_nop 3\n\t" CNF_PKMUL(0, 4, 8) CNF_PKMUL(1, 5, 9) CNF_PKMUL(2, 6, 10) CNF_PKMUL(3, 7, 11) "s_nop 1"
            : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
    } else if constexpr (N == 3) {
        asm volatile("s_nop 7\n\ts_nop 3\n\t" CNF_PKMUL(0, 6, 12) CNF_PKMUL(1, 7, 13) CNF_PKMUL(2, 8, 14) CNF_PKMUL(3, 9, 15) CNF_PKMUL(4, 10, 16)
            CNF_PKMUL(5, 11, 17) "s_nop 1"
            : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]));
    } else {
        asm volatile("s_nop 7\n\ts_nop 3\n\t" CNF_PKMUL(0, 8, 16) CNF_PKMUL(1, 9, 17) CNF_PKMUL(2, 10, 18) CNF_PKMUL(3, 11, 19) CNF_PKMUL(4, 12, 20)
            CNF_PKMUL(5, 13, 21) CNF_PKMUL(6, 14, 22) CNF_PKMUL(7, 15, 23) "s_nop 1"
            : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
              "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]));
    }
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = f32x4{o[2 * i][0], o[2 * i][1], o[2 * i + 1][0], o[2 * i + 1][1]};
}
// tiles_mul_block with the operands as named halves of whole tiles (no local operand arrays), for tiles that live in a
// [KP][HT] array (vjp_probe_hoisted): an array of f32x2 operands is vector-promoted to a
// <16 x float>, the tile loads that feed it become "widen to 16 lanes" shuffles, and the early VectorCombine turns those into
// 64-byte loads that straddle the tiles of a [KP][HT] array - which then cannot be split into registers and stays in scratch.
#define CNF_LO(t) __builtin_shufflevector(t, t, 0, 1)
#define CNF_HI(t) __builtin_shufflevector(t, t, 2, 3)
#define CNF_TILE(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3)
template <int N>
__device__ __forceinline__ void tiles_mul_block_named(const f32x4* a, const f32x4* b, f32x4* c) {
    static_assert(N >= 1 && N <= 4, "at most 4 tiles (24 asm operands) per statement");
    f32x2 o0, o1, o2, o3, o4, o5, o6, o7;
    if constexpr (N == 1) {
        const f32x4 a0 = a[0], b0 = b[0];
        asm volatile("s_nop 7\n\ts_nop 3\n\t" CNF_PKMUL(0, 2, 4) CNF_PKMUL(1, 3, 5) "s_nop 1"
            : "=&v"(o0), "=&v"(o1) : "v"(CNF_LO(a0)), "v"(CNF_HI(a0)), "v"(CNF_LO(b0)), "v"(CNF_HI(b0)));
        c[0] = CNF_TILE(o0, o1);
    } else if constexpr (N == 2) {
        const f32x4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
        asm volatile("s_nop 7\n\ts_nop 3\n\t" CNF_PKMUL(0, 4, 8) CNF_PKMUL(1, 5, 9) CNF_PKMUL(2, 6, 10) CNF_PKMUL(3, 7, 11) "s_nop 1"
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
            : "v"(CNF_LO(a0)), "v"(CNF_HI(a0)), "v"(CNF_LO(a1)), "v"(CNF_HI(a1)), "v"(CNF_LO(b0)), "v"(CNF_HI(b0)), "v"(CNF_LO(b1)), "v"(CNF_HI(b1)));
        c[0] = CNF_TILE(o0, o1); c[1] = CNF_TILE(o2, o3);
    } else if constexpr (N == 3) {
        const f32x4 a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2];
        asm volatile("s_nop 7\n\ts_nop 3\n\t" CNF_PKMUL(0, 6, 12) CNF_PKMUL(1, 7, 13) CNF_PKMUL(2, 8, 14) CNF_PKMUL(3, 9, 15) CNF_PKMUL(4, 10, 16)
            CNF_PKMUL(5, 11, 17) "s_nop 1"
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5)
            : "v"(CNF_LO(a0)), "v"(CNF_HI(a0)), "v"(CNF_LO(a1)), "v"(CNF_HI(a1)), "v"(CNF_LO(a2)), "v"(CNF_HI(a2)),
              "v"(CNF_LO(b0)), "v"(CNF_HI(b0)), "v"(CNF_LO(b1)), "v"(CNF_HI(b1)), "v"(CNF_LO(b2)), "v"(CNF_HI(b2)));
        c[0] = CNF_TILE(o0, o1); c[1] = CNF_TILE(o2, o3); c[2] = CNF_TILE(o4, o5);
    } else {
        const f32x4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
        asm volatile("s_nop 7\n\ts_nop 3\n\t" CNF_PKMUL(0, 8, 16) CNF_PKMUL(1, 9, 17) CNF_PKMUL(2, 10, 18) CNF_PKMUL(3, 11, 19) CNF_PKMUL(4, 12, 20)
            CNF_PKMUL(5, 13, 21) CNF_PKMUL(6, 14, 22) CNF_PKMUL(7, 15, 23) "s_nop 1"
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5), "=&v"(o6), "=&v"(o7)
            : "v"(CNF_LO(a0)), "v"(CNF_HI(a0)), "v"(CNF_LO(a1)), "v"(CNF_HI(a1)), "v"(CNF_LO(a2)), "v"(CNF_HI(a2)), "v"(CNF_LO(a3)), "v"(CNF_HI(a3)),
              "v"(CNF_LO(b0)), "v"(CNF_HI(b0)), "v"(CNF_LO(b1)), "v"(CNF_HI(b1)), "v"(CNF_LO(b2)), "v"(CNF_HI(b2)), "v"(CNF_LO(b3)), "v"(CNF_HI(b3)));
        c[0] = CNF_TILE(o0, o1); c[1] = CNF_TILE(o2, o3); c[2] = CNF_TILE(o4, o5); c[3] = CNF_TILE(o6, o7);
    }
}
#else
__device__ __forceinline__ void pk_add1(f32x2& a, f32x2& b) { a = a + 1.f; b = b + 1.f; }
__device__ __forceinline__ void pk_tanh_from_r(f32x2& ha, f32x2& hb, f32x2& da, f32x2& db, const f32x2& ra, const f32x2& rb) {
    const f32x2 two = {2.f, 2.f}, one = {1.f, 1.f};
    ha = __builtin_elementwise_fma(ra, two, -one); hb = __builtin_elementwise_fma(rb, two, -one);
    da = __builtin_elementwise_fma(-ha, ha, one); db = __builtin_elementwise_fma(-hb, hb, one);
}
template <int N>
__device__ __forceinline__ void tiles_mul_block(const f32x4* a, const f32x4* b, f32x4* c) {
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = a[i] * b[i];
}
template <int N>
__device__ __forceinline__ void tiles_mul_block_named(const f32x4* a, const f32x4* b, f32x4* c) { tiles_mul_block<N>(a, b, c); }
#endif
// c = a .* b over MT accumulator tiles, in statements of up to 4 tiles
template <int MT>
__device__ __forceinline__ void tiles_mul(const f32x4 (&a)[MT], const f32x4 (&b)[MT], f32x4 (&c)[MT]) {
    constexpr int FULL = MT / 4, REM = MT % 4;
#pragma unroll
    for (int q = 0; q < FULL; ++q) tiles_mul_block<4>(&a[4 * q], &b[4 * q], &c[4 * q]);
    if constexpr (REM > 0) tiles_mul_block<REM>(&a[4 * FULL], &b[4 * FULL], &c[4 * FULL]);
}
template <int MT>
__device__ __forceinline__ void tiles_mul_named(const f32x4 (&a)[MT], const f32x4 (&b)[MT], f32x4 (&c)[MT]) {
    constexpr int FULL = MT / 4, REM = MT % 4;
#pragma unroll
    for (int q = 0; q < FULL; ++q) tiles_mul_block_named<4>(&a[4 * q], &b[4 * q], &c[4 * q]);
    if constexpr (REM > 0) tiles_mul_block_named<REM>(&a[4 * FULL], &b[4 * FULL], &c[4 * FULL]);
}
__device__ __forceinline__ f32x4 tile_fma(const f32x4& a, float s, const f32x4& c) {   // a * s + c
    const f32x2 a0 = {a[0], a[1]}, a1 = {a[2], a[3]}, c0 = {c[0], c[1]}, c1 = {c[2], c[3]}, ss = {s, s};
    const f32x2 r0 = __builtin_elementwise_fma(a0, ss, c0), r1 = __builtin_elementwise_fma(a1, ss, c1);
    return f32x4{r0[0], r0[1], r1[0], r1[1]};
}
// sum over the tiles of <a, b>, accumulated two lanes' worth per issue
template <int MT>
__device__ __forceinline__ float tiles_dot(const f32x4 (&a)[MT], const f32x4 (&b)[MT]) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x2 a0 = {a[mt][0], a[mt][1]}, a1 = {a[mt][2], a[mt][3]}, b0 = {b[mt][0], b[mt][1]}, b1 = {b[mt][2], b[mt][3]};
        acc = __builtin_elementwise_fma(a0, b0, acc);
        acc = __builtin_elementwise_fma(a1, b1, acc);
    }
    return acc[0] + acc[1];
}

// softplus and its derivative on one accumulator tile: NNlib.softplus(a) = log1p(exp(-|a|)) + relu(a), d = sigmoid(a)
// (cnf_common.h: act_fwd) with the bare transcendentals - e = v_exp(-|a| log2 e) lies in (0, 1], so 1 + e in (1, 2] needs none of
// __expf's / __logf's range handling (those expand to ~16 VALU instructions per element; this is 8.5) - and the affine steps on
// register pairs (v_pk_add / v_pk_fma / v_pk_mul).  Same formulas, |error| <= 2e-7 as before.
__device__ __forceinline__ void softplus_tile(const f32x4& a, f32x4& h, f32x4& d) {
    constexpr float kNegLog2e = -1.4426950408889634f, kLn2 = 0.6931471805599453f;
    float e[4], r[4], lg[4], mx[4], sel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(__builtin_fabsf(a[i]) * kNegLog2e);
    const f32x2 one = {1.f, 1.f};
    const f32x2 s0 = f32x2{e[0], e[1]} + one, s1 = f32x2{e[2], e[3]} + one;
    const float sv[4] = {s0[0], s0[1], s1[0], s1[1]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r[i] = __builtin_amdgcn_rcpf(sv[i]);
        lg[i] = __builtin_amdgcn_logf(sv[i]);       // log2(1 + e)
        mx[i] = __builtin_fmaxf(a[i], 0.f);
        sel[i] = a[i] >= 0.f ? 1.f : e[i];
    }
    const f32x2 ln2 = {kLn2, kLn2};
    const f32x2 h0 = __builtin_elementwise_fma(f32x2{lg[0], lg[1]}, ln2, f32x2{mx[0], mx[1]});
    const f32x2 h1 = __builtin_elementwise_fma(f32x2{lg[2], lg[3]}, ln2, f32x2{mx[2], mx[3]});
    const f32x2 d0 = f32x2{r[0], r[1]} * f32x2{sel[0], sel[1]}, d1 = f32x2{r[2], r[3]} * f32x2{sel[2], sel[3]};
    h = f32x4{h0[0], h0[1], h1[0], h1[1]};
    d = f32x4{d0[0], d0[1], d1[0], d1[1]};
}
template <int ACT>
__device__ __forceinline__ void act_tile(const f32x4& a, f32x4& h, f32x4& d) {
    if constexpr (ACT == CNF_ACT_TANH_PRESCALED || ACT == CNF_ACT_TANH) {
        // tanh(a) = 2 / (1 + exp(-2a)) - 1, tanh' = 1 - tanh^2: v_exp and v_rcp per value, the rest two values per issue
        f32x2 x0 = {a[0], a[1]}, x1 = {a[2], a[3]};
        if constexpr (ACT == CNF_ACT_TANH) { x0 = x0 * kTanhPrescale; x1 = x1 * kTanhPrescale; }
        f32x2 e0 = {__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1])};
        f32x2 e1 = {__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1])};
        pk_add1(e0, e1);
        const f32x2 r0 = {fast_rcp(e0[0]), fast_rcp(e0[1])}, r1 = {fast_rcp(e1[0]), fast_rcp(e1[1])};
        f32x2 h0, h1, d0, d1;
        pk_tanh_from_r(h0, h1, d0, d1, r0, r1);
        h = f32x4{h0[0], h0[1], h1[0], h1[1]};
        d = f32x4{d0[0], d0[1], d1[0], d1[1]};
    } else if constexpr (ACT == CNF_ACT_SOFTPLUS) {
        softplus_tile(a, h, d);   // (round 4: every softplus kernel of the library - 8.5 instead of ~16 VALU instructions per element)
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float dd;
            h[r] = act_fwd<ACT>(a[r], dd);
            d[r] = dd;
        }
    }
}

// Phase fence for the instruction schedulers: MFMA, VALU and transcendental instructions stay on their side, LDS reads and
// scalar instructions may still cross (operand prefetch).  On gfx950 every MFMA -> VALU -> MFMA round trip costs ~9 issue
// cycles on top of the instructions themselves (profiles/ubench/mfma_valu_overlap.result.txt: 32.0 -> 45.3 cycles for one
// v_fma behind an MFMA, +4.4 per further one), and f32 MFMAs hide no VALU work, so interleaving the two - which LLVM does
// eagerly (39 round trips per dynamics call at cfg2) - only costs.  -DCNF_NO_PHASE_FENCE is the A/B switch.
__device__ __forceinline__ void phase_fence() {
#ifndef CNF_NO_PHASE_FENCE
    __builtin_amdgcn_sched_barrier(0x0104);
#endif
}

// Pullback of ONE Hutchinson probe whose solve-invariant c = W_N^T eps has been hoisted, for the instances that carry
// several probes (PRE = 1, KP > 1: cfg3).  Called KP times from a compile-time loop, so probe k's c_k and eps_k are named
// registers (a rolled probe loop would pick them with a v_cndmask per register per probe) and c_k is not recomputed on every
// dynamics call (8 of the 152 MFMAs of a probe at cfg3).  Every copy reads the operand images through its own opaque
// offset, so the copies' LDS reads are not merged (and spilled).  The K = 1 kernels keep their own (older) text in dyn_eval:
// their schedules are sensitive to any change of it (+-1.5 % from semantically neutral edits).
template <int HT, int L, int ZR, int CR, int ARITH>
__device__ __forceinline__ void vjp_probe_hoisted(const float* __restrict__ smem, int lane, bool reg_j, float scale,
                                                  const float (&ep)[ZR], const f32x4 (&pc)[HT], const f32x4 (&d)[L][HT],
                                                  float& ld, float& nd) {
    constexpr MfmaLayout LAY(HT, L, ZR, CR, true, ARITH);
    constexpr int DT = (ZR + 3) / 4;
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const float* __restrict__ sm = smem + opq;
    f32x4 dl[HT];
    tiles_mul_named<HT>(pc, d[L - 1], dl);
#pragma unroll
    for (int l = L - 1; l >= 1; --l) {  // W_{l+1}^T delta, times act'(a_l)
        f32x4 acc[HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        phase_fence();
        gemm_hidden<HT, ARITH>(sm + LAY.bh + (l - 1) * LAY.imgHid(), lane, dl, acc);
        phase_fence();
        tiles_mul_named<HT>(acc, d[l - 1], dl);
    }
    f32x4 gacc[DT];
#pragma unroll
    for (int dt_ = 0; dt_ < DT; ++dt_) gacc[dt_] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_tiles<DT, 4 * HT>(sm + LAY.b1, lane, TileIn<HT>{dl}, gacc);   // W_1[:,0:D]^T delta_1
    float dot = 0.f, n2 = 0.f;
#pragma unroll
    for (int s = 0; s < ZR; ++s) {
        const float gv = gacc[s >> 2][s & 3];
        dot = fmaf(gv, ep[s], dot);
        n2 = fmaf(gv, gv, n2);
    }
    ld -= scale * group_sum(dot);
    if (reg_j) nd += scale * fast_sqrt(group_sum(n2));   // ndot = |eps^T J|_2 (icnf.jl:229-245)
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
                                         float& ed, float& nd, const f32x4 (&pre_ck)[(PRE >= 1 && KP > 1) ? KP : 1][HT]) {
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
            for (int mt = 0; mt < HT; ++mt) acc[mt] = tile_fma(wt[mt], t, acc[mt]);
        }
        phase_fence();
        gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{z}, acc);
        if constexpr (CR > 0) gemm_tiles<HT, CR>(smem + LAY.f1y, lane, RegIn<CR>{y}, acc);
        phase_fence();
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) act_tile<ACT>(acc[mt], h[mt], d[0][mt]);
        phase_fence();
    }
    // ---- hidden layers 2..L ----
#pragma unroll
    for (int l = 1; l < L; ++l) {
        f32x4 acc[HT];
        load_cvec<HT>(smem + LAY.v_bh + (l - 1) * MfmaLayout::vecC(HT), g, acc);
        gemm_hidden<HT, ARITH>(smem + LAY.fh + (l - 1) * LAY.imgHid(), lane, h, acc);
        phase_fence();
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) act_tile<ACT>(acc[mt], h[mt], d[l][mt]);
        phase_fence();
    }
    // ---- last layer (identity): zdot ----
    {
        f32x4 acc[DT];
        load_cvec<DT>(smem + LAY.v_bN, g, acc);
        gemm_tiles<DT, 4 * HT>((LAY.fN_global ? gimg : smem) + LAY.fN, lane, TileIn<HT>{h}, acc);
        phase_fence();
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = acc[s >> 2][s & 3];
    }
    ed = 0.f;
    if (reg_z) {
        float e2 = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) e2 = fmaf(zd[s], zd[s], e2);
        // Edot = |zdot|_2   (src/core/icnf.jl:184-199); the several-probe instances take v_sqrt_f32 alone
        if constexpr (PRE >= 1 && KP > 1) ed = fast_sqrt(group_sum(e2));
        else ed = sqrtf(group_sum(e2));
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
    if constexpr (ENGINE == ENG_VJP && PRE >= 1 && KP > 1) {
        // K == KP exactly (the plan picks such an instance only for nprobes == KP)
        static_for<0, KP>([&](auto pi) {
            constexpr int p = decltype(pi)::value;
            vjp_probe_hoisted<HT, L, ZR, CR, ARITH>(smem, lane, reg_j, scale, eps[p], pre_ck[p], d, ld, nd);
        });
        return;
    }
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
                tiles_mul<HT>(pre_c, d[L - 1], dl);
            } else {
                f32x4 acc[HT];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_tiles<HT, ZR>(sm + LAY.bN, lane, RegIn<ZR>{ep}, acc);   // W_N^T eps
                tiles_mul<HT>(acc, d[L - 1], dl);
            }
#pragma unroll
            for (int l = L - 1; l >= 1; --l) {  // W_{l+1}^T delta, times act'(a_l)
                f32x4 acc[HT];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                phase_fence();
                gemm_hidden<HT, ARITH>(sm + LAY.bh + (l - 1) * LAY.imgHid(), lane, dl, acc);
                phase_fence();
                tiles_mul<HT>(acc, d[l - 1], dl);
            }
            if constexpr (PRE == 2) {
                // without the |eps^T J| regulariser only <eps^T J, eps> = <delta_1, W_1[:,0:D] eps> is
                // needed: a dot product with the hoisted q = W_1[:,0:D] eps replaces the last product
                const float qd = tiles_dot<HT>(dl, pre_q);
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
                tiles_mul<HT>(acc, d[0], tau);
            }
#pragma unroll
            for (int l = 1; l < L; ++l) {  // act'(a_{l+1}) .* (W_{l+1} tau)
                f32x4 acc[HT];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                phase_fence();
                gemm_hidden<HT, ARITH>(sm + LAY.fh + (l - 1) * LAY.imgHid(), lane, tau, acc);
                phase_fence();
                tiles_mul<HT>(acc, d[l], tau);
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

// Stage `n4` f32x4 of the packed image from global memory (L2) into LDS with `NTHREADS` threads, 16 B per lane, coalesced.
// Eight loads are in flight per lane before the first LDS write: the plain copy loop compiles to load - s_waitcnt vmcnt(0) -
// ds_write per iteration, i.e. one full L2 round trip per 8 KB (cfg2: 83 KB = 11 serial round trips, 5 - 10 us per launch with
// every workgroup of the chip reading the same lines) - nothing in a 3 ms solve, a third of a 29 us single dynamics call
// (boundary A: cnf_aug_f).
template <int NTHREADS>
__device__ __forceinline__ void stage_image(const float* __restrict__ packed, float* __restrict__ smem, int n4) {
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(packed);
    f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(smem);
    constexpr int DEPTH = 8;
    for (int base = threadIdx.x; base < n4; base += DEPTH * NTHREADS) {
        f32x4 t[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const int i = base + k * NTHREADS;
            t[k] = src[i < n4 ? i : n4 - 1];      // clamped, not predicated: no control flow between the loads
        }
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const int i = base + k * NTHREADS;
            if (i < n4) dst[i] = t[k];
        }
    }
}

// PRE: 0 none; 1 hoist c = W_N^T eps; 2 also hoist q = W_1[:,0:D] eps and skip the last pullback
// product (valid only without reg_j).  PRE > 0 needs ENGINE == ENG_VJP and KP == 1.
template <int HT, int L, int ZR, int CR, int ACT, int ENGINE, int KP, int PRE, int NTHREADS, int ARITH = 0>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, (NTHREADS + 255) / 256)))
mfma_solve_kernel(KArgs a) {
    constexpr MfmaLayout LAY(HT, L, ZR, CR, ENGINE == ENG_VJP, ARITH);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // stage the packed weight image: global (L2) -> LDS
    stage_image<NTHREADS>(a.packed, smem, LAY.lds_total / 4);
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

    // tiles go to workgroups first, then to the waves of a workgroup: a batch of fewer tiles than the chip has wave slots
    // spreads over the compute units (one latency-bound wave per SIMD) instead of filling a few of them
    for (long long tile = (long long)blockIdx.x + (long long)gridDim.x * wave; tile < ntiles;) {
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
        // No stage derivative is stored.  Each one is folded at once into (i) the running sums of the step update
        // (sum_j b_j k_j for z, dlogp, E, n) and (ii) the partial sums P[i] of the state increments of the stages still to
        // come: P[i] holds sum_{j <= st} a[st+1+i][j] kz_j, so after stage st   P[i] <- P[i+1] + a[st+2+i][st] * zdot   (the
        // shift is the fma's own operand choice - no moves, no predicated selects), and the next stage starts from
        // z + dt P[0].  Same fma chains in the same order as summing stored derivatives: bit-identical results.
        float P[5][ZR], zsum[ZR], lsum, esum, nsum;
        float zd[ZR], ld = 0.f, ed = 0.f, nd = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = 0.f;
        f32x4 pre_c[HT], pre_q[HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) pre_c[mt] = pre_q[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE >= 1 && KP == 1) gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps[0]}, pre_c);
        // several probes: c_k = W_N^T eps_k of every probe, once per tile (see vjp_probe_hoisted)
        static_assert(PRE == 0 || ENGINE == ENG_VJP, "hoisting is a VJP-engine feature");
        static_assert(PRE < 2 || KP == 1, "the dot-product shortcut (PRE = 2) is instantiated for one probe");
        f32x4 pre_ck[(PRE >= 1 && KP > 1) ? KP : 1][HT];
        if constexpr (PRE >= 1 && KP > 1)
            static_for<0, KP>([&](auto pi) {
                constexpr int p = decltype(pi)::value;
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) pre_ck[p][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps[p]}, pre_ck[p]);
            });
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
            lsum = esum = nsum = 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                zsum[s] = 0.f;
#pragma unroll
                for (int i = 0; i < 5; ++i) P[i][s] = 0.f;
            }
#pragma clang loop unroll(disable)
            for (int st = 0; st < ns; ++st) {
                float zs[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) zs[s] = fmaf(dt, P[0][s], z[s]);
                // The weight image is loop-invariant, and LLVM would hoist all ~300 operand reads out
                // of the RK loops (and spill them).  An opaque zero offset pins the reads per stage.
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                dyn_eval<HT, L, ZR, CR, ACT, ENGINE, KP, PRE, ARITH>(smem + opaque, a.packed + opaque, lane, tn + a.T.c[st] * dt, autonomous,
                                                              reg_z, reg_j, exact, D, K, zs, y, eps, pre_c, pre_q, zd,
                                                              ld, ed, nd, pre_ck);
                if (a.ckpt_k) {
#pragma unroll
                    for (int s = 0; s < ZR; ++s)
                        a.ckpt_k[((((long long)step * ns + st) * ntiles + tile) * 64 + lane) * ZR + s] = zd[s];
                }
#ifndef CNF_NO_KFULL   // (A/B switch: the cost of this block on the metric kernel is measured with -DCNF_NO_KFULL)
                if (a.kfull && valid) {
                    // all S rows of this stage's derivative in the ABI's layout, [stage][sample][row]: the embedded error
                    // estimate of an adaptive attempt (cnf_step_embedded, nsteps = 1); the metric path pays one untaken branch
                    float* kf = a.kfull + ((long long)st * a.B + smp) * S;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) kf[f] = zd[s]; }
                    if (g == 0) { kf[D] = ld; kf[D + 1] = ed; kf[D + 2] = nd; }
                }
#endif
                const float bst = a.T.b[st];
                lsum = fmaf(bst, ld, lsum); esum = fmaf(bst, ed, esum); nsum = fmaf(bst, nd, nsum);
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    zsum[s] = fmaf(bst, zd[s], zsum[s]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) P[i][s] = fmaf(a.acol[st][i], zd[s], P[i + 1][s]);
                    P[4][s] = a.acol[st][4] * zd[s];
                }
            }
            if (single) break;
            lacc = fmaf(dt, lsum, lacc); eacc = fmaf(dt, esum, eacc); nacc = fmaf(dt, nsum, nacc);
#pragma unroll
            for (int s = 0; s < ZR; ++s) z[s] = fmaf(dt, zsum[s], z[s]);
        }

        if (single) {
            if (valid) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) a.u_out[smp * S + f] = zd[s]; }
                if (g == 0) { a.u_out[smp * S + D] = ld; a.u_out[smp * S + D + 1] = ed; a.u_out[smp * S + D + 2] = nd; }
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

#ifdef CNF_WITH_DEVICE_CONTROLLER
// ---- adaptive Tsit5 with the step controller on the device: one launch per solve ----
// OrdinaryDiffEq's adaptive loop as cnf_api_adaptive.hip::api_solve_tsit5 restates it on the host (Hairer's initial step,
// embedded estimate dt sum btilde_i k_i scaled by abstol + reltol max(|u|, |u_new|), RMS over the whole S x B state, PI
// controller, first-same-as-last), for batches of at most one tile per resident wave.  The host loop pays 3-4 launches and
// a device-to-host round trip per attempt (~70 us); here an attempt costs its six dynamics calls and one grid-wide sum.
// Sum of a double over the 64 lanes, the same bits in every lane (each step adds a lane's value to its partner's, and the
// partner does the mirror-image addition).  Within a row of 16 lanes the partners come from DPP modifiers (quad permutes,
// half-row and row mirrors), across rows from the gfx950 row / half swaps - all VALU-speed; the shuffle form
// (`__shfl_xor` on a double = two ds_bpermute_b32 + an LDS round trip per step, six steps) cost ~2 us per call, and a
// grid-wide sum makes two calls for each of its three values.
__device__ __forceinline__ double wave_sum_f64(double v) {
    auto halves = [](double x, unsigned& lo, unsigned& hi) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
        lo = (unsigned)b; hi = (unsigned)(b >> 32);
    };
    auto whole = [](unsigned lo, unsigned hi) { return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); };
    unsigned lo, hi;
#define CNF_DPP_STEP(CTRL)                                                                       \
    halves(v, lo, hi);                                                                           \
    v += whole((unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, false),        \
               (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, false));
    CNF_DPP_STEP(0xB1)    // quad_perm [1,0,3,2]: lane ^ 1
    CNF_DPP_STEP(0x4E)    // quad_perm [2,3,0,1]: lane ^ 2
    CNF_DPP_STEP(0x141)   // row_half_mirror: i <-> 7 - i (the other quad of the 8)
    CNF_DPP_STEP(0x140)   // row_mirror: i <-> 15 - i (the other 8 of the row)
#undef CNF_DPP_STEP
    halves(v, lo, hi);
    {   // rows 16 apart
        const u32x2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = whole(a[0], b[0]) + whole(a[1], b[1]);
    }
    halves(v, lo, hi);
    {   // halves 32 apart
        const u32x2_t a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = whole(a[0], b[0]) + whole(a[1], b[1]);
    }
    return v;
}

// Grid-wide sums of three per-lane values over all tiles, in two levels: the waves of a workgroup meet in LDS; wave 0 of every
// workgroup publishes the workgroup's partials, waits for the partials of all workgroups and adds them up in workgroup order.  Every
// wave ends with the same three doubles (same partials, same order).  `round` counts the calls (the same in every wave): its parity
// picks the slot set, round + 1 is the tag.
// A partial travels as 8-byte words {tag, half of the double} written with agent-scope atomic stores (write-through) and read
// with agent-scope atomic loads (past the XCD's L2): a word is there or not, and a reader that sees the tag in all six words of a
// slot has the three doubles - ONE trip to the coherence point for the writer (not waited for) and one for the reader.  The first
// form of this function (partials, then a release increment of an arrival counter, a poll of it, an acquire fence, the partials)
// was four dependent trips and two whole-L2 write-backs / invalidations: ~10 us per sum, 2/3 of a small-batch VCABM solve.
// The tag is {launch epoch, round + 1 (mod 2^16)}: what an earlier launch left in a slot carries another epoch, what this launch left
// there two rounds ago another round, so nothing is cleared between launches (AArgs::epoch).  A slot set is reused two rounds
// later, which its owner reaches only after every workgroup has published the round in between, i.e. has finished reading this one.
// (Spare waves of a workgroup have left the kernel; the hardware barrier counts the waves still running.)
// Returns false when the other workgroups did not arrive within ~2^22 polls (seconds): a safety net, not a code path - a
// grid whose workgroups are not all resident would otherwise spin for ever (and take the GPU with it).  The first workgroup
// to give up raises q.stats[5] (to the launch's epoch); every poller watches that word too, so the whole grid leaves within one poll.
__device__ __forceinline__ bool grid_sum3(const AArgs& q, unsigned& round, int wave, int nact, int lane, double& v0, double& v1, double& v2) {
    __shared__ double wg_part[3][16];
    __shared__ double wg_tot[3];
    __shared__ int wg_ok;
    v0 = wave_sum_f64(v0);
    v1 = wave_sum_f64(v1);
    v2 = wave_sum_f64(v2);
    if (lane == 0) { wg_part[0][wave] = v0; wg_part[1][wave] = v1; wg_part[2][wave] = v2; }
    __syncthreads();
    if (wave == 0) {
        const unsigned nb = gridDim.x;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        if (lane == 0)
            for (int w = 0; w < nact; ++w) { a0 += wg_part[0][w]; a1 += wg_part[1][w]; a2 += wg_part[2][w]; }
        int ok = 1;
        if (nb > 1) {   // (a single workgroup has its sums already)
            typedef unsigned long long u64;
            const u64 tag = (u64)((q.epoch << 16) | ((round + 1u) & 0xffffu)) << 32;
            u64* sl = (u64*)q.slots + (size_t)(round & 1u) * 6u * (size_t)nb;
            if (lane == 0) {
                u64* mine = sl + 6u * (size_t)blockIdx.x;
                const u64 b0 = (u64)__double_as_longlong(a0), b1 = (u64)__double_as_longlong(a1), b2 = (u64)__double_as_longlong(a2);
                __hip_atomic_store(mine + 0, tag | (b0 & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + 1, tag | (b0 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + 2, tag | (b1 & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + 3, tag | (b1 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + 4, tag | (b2 & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + 5, tag | (b2 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            a0 = a1 = a2 = 0.0;
            unsigned polls = 0;
            for (unsigned i = lane; i < nb && ok; i += 64) {
                const u64* s = sl + 6u * (size_t)i;
                u64 w[6];
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        w[j] = __hip_atomic_load(s + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ready = ready && (w[j] & 0xffffffff00000000ull) == tag;
                    }
                    if (ready) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++polls & 63u) == 0u &&
                        (polls > (1u << 22) || __hip_atomic_load(q.stats + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)q.epoch)) {
                        __hip_atomic_store(q.stats + 5, (int)q.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
                if (ok) {
                    a0 += __longlong_as_double((long long)((w[0] & 0xffffffffull) | (w[1] << 32)));
                    a1 += __longlong_as_double((long long)((w[2] & 0xffffffffull) | (w[3] << 32)));
                    a2 += __longlong_as_double((long long)((w[4] & 0xffffffffull) | (w[5] << 32)));
                }
            }
            ok = __builtin_amdgcn_ballot_w64(ok == 0) == 0ull ? 1 : 0;
            a0 = wave_sum_f64(a0);
            a1 = wave_sum_f64(a1);
            a2 = wave_sum_f64(a2);
        }
        if (lane == 0) { wg_tot[0] = a0; wg_tot[1] = a1; wg_tot[2] = a2; wg_ok = ok; }
    }
    __syncthreads();
    v0 = wg_tot[0];
    v1 = wg_tot[1];
    v2 = wg_tot[2];
    ++round;
    return wg_ok != 0;
}
__device__ __forceinline__ bool grid_sum2(const AArgs& q, unsigned& round, int wave, int nact, int lane, double& v0, double& v1) {
    double v2 = 0.0;
    return grid_sum3(q, round, wave, nact, lane, v0, v1, v2);
}

template <int HT, int L, int ZR, int CR, int ACT, int ENGINE, int KP, int PRE, int NTHREADS, int ARITH = 0>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, (NTHREADS + 255) / 256)))
mfma_adaptive_kernel(KArgs a, AArgs q) {
    constexpr MfmaLayout LAY(HT, L, ZR, CR, ENGINE == ENG_VJP, ARITH);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stage_image<NTHREADS>(a.packed, smem, LAY.lds_total / 4);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, n = lane & 15;
    const int wave = threadIdx.x >> 6;
    const long long ntiles = (a.B + 15) / 16;
    const long long tile = (long long)blockIdx.x + (long long)gridDim.x * wave;
    if (tile >= ntiles) return;   // the host sizes the grid so that every tile has a wave; spare waves leave
    // waves of this workgroup that own a tile (wave w owns tile blockIdx + gridDim w): they meet in grid_sum2
    const int nact = (int)((ntiles - 1 - (long long)blockIdx.x) / (long long)gridDim.x) + 1 < NTHREADS / 64
                         ? (int)((ntiles - 1 - (long long)blockIdx.x) / (long long)gridDim.x) + 1 : NTHREADS / 64;
    const int D = a.D, S = D + 3, C = a.C;
    const int K = KP == 1 ? 1 : a.K;
    const int Kd = K * D;
    const bool reg_z = a.reg_z, reg_j = a.reg_j, autonomous = a.autonomous, exact = a.exact;
    const long long smp = tile * 16 + n;
    const bool valid = smp < a.B;
    const long long sc = valid ? smp : a.B - 1;
    float z[ZR], eps[KP][ZR], y[CR > 0 ? CR : 1];
#pragma unroll
    for (int s = 0; s < ZR; ++s) {
        const int f = 4 * s + g;
        z[s] = f < D ? a.u0[sc * S + f] : 0.f;
#pragma unroll
        for (int p = 0; p < KP; ++p) eps[p][s] = (f < D && p < K && a.eps) ? a.eps[sc * Kd + p * D + f] : 0.f;
    }
    float la = a.u0[sc * S + D], ea = a.u0[sc * S + D + 1], na = a.u0[sc * S + D + 2];
    y[0] = 0.f;
    if constexpr (CR > 0) {
#pragma unroll
        for (int s = 0; s < CR; ++s) { const int f = 4 * s + g; y[s] = f < C ? a.ys[sc * C + f] : 0.f; }
    }
    f32x4 pre_c[HT], pre_q[HT];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) pre_c[mt] = pre_q[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PRE >= 1 && KP == 1) gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps[0]}, pre_c);
    // several probes: c_k = W_N^T eps_k of every probe, once per tile (see vjp_probe_hoisted)
    static_assert(PRE == 0 || ENGINE == ENG_VJP, "hoisting is a VJP-engine feature");
    static_assert(PRE < 2 || KP == 1, "the dot-product shortcut (PRE = 2) is instantiated for one probe");
    f32x4 pre_ck[(PRE >= 1 && KP > 1) ? KP : 1][HT];
    if constexpr (PRE >= 1 && KP > 1)
        static_for<0, KP>([&](auto pi) {
            constexpr int p = decltype(pi)::value;
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) pre_ck[p][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps[p]}, pre_ck[p]);
        });
    if constexpr (PRE >= 2) {
        gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{eps[0]}, pre_q);
        if constexpr (ACT == CNF_ACT_TANH_PRESCALED) {
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) pre_q[mt] *= (1.f / kTanhPrescale);
        }
    }

    const double t0 = (double)a.t0, t1 = (double)q.t1;
    const double span = fabs(t1 - t0), tdir = t1 >= t0 ? 1.0 : -1.0, ntot = (double)S * (double)a.B;
    const float abstol = q.abstol, reltol = q.reltol;
    // a lane's share of the state: its ZR rows of z, and - in lane group 0 - the three scalar rows
    auto row_live = [&](int s) { return valid && 4 * s + g < D; };
    const bool scal_live = valid && g == 0;

    float k1z[ZR], k1l = 0.f, k1e = 0.f, k1n = 0.f;      // derivative at (z, t): first stage of the next attempt
    float P[6][ZR], erz[ZR], lsum = 0.f, esum = 0.f, nsum = 0.f, erl = 0.f, ere = 0.f, ern = 0.f;
    float zs[ZR], zd[ZR], ld = 0.f, ed = 0.f, nd = 0.f;
#pragma unroll
    for (int s = 0; s < ZR; ++s) {
        zs[s] = z[s]; zd[s] = 0.f; k1z[s] = 0.f; erz[s] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) P[i][s] = 0.f;
    }
    double t = t0, dt = 0.0, qold = 1e-4, step = 0.0, h0 = 0.0, d1 = 0.0;
    float tf = 0.f, dtf = 0.f, tcur = a.t0;
    bool last = false;
    int phase = 0, st = 0, it = 0, naccept = 0, nreject = 0, nf = 0, status = 0;
    unsigned round = 0;

    for (;;) {
        // one dynamics call per trip: at (u0, t0) first, at the Euler point of Hairer's rule second, then the stages
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        dyn_eval<HT, L, ZR, CR, ACT, ENGINE, KP, PRE, ARITH>(smem + opaque, a.packed + opaque, lane, tcur, autonomous, reg_z, reg_j, exact,
                                                      D, K, zs, y, eps, pre_c, pre_q, zd, ld, ed, nd, pre_ck);
        ++nf;
        bool begin = false;
        if (phase == 0) {
#pragma unroll
            for (int s = 0; s < ZR; ++s) k1z[s] = zd[s];
            k1l = ld; k1e = ed; k1n = nd;
            if (q.dt_init != 0.f) {
                dt = fmin((double)fabsf(q.dt_init), span);
                begin = true;
            } else {   // ode_determine_initdt (Hairer, Noersett, Wanner I, II.4), order 5
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    if (row_live(s)) {
                        const float sk = fmaf(fabsf(z[s]), reltol, abstol), r0 = z[s] / sk, r1 = k1z[s] / sk;
                        s0 += (double)r0 * (double)r0; s1 += (double)r1 * (double)r1;
                    }
                }
                if (scal_live) {
                    const float u3[3] = {la, ea, na}, f3[3] = {k1l, k1e, k1n};
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float sk = fmaf(fabsf(u3[i]), reltol, abstol), r0 = u3[i] / sk, r1 = f3[i] / sk;
                        s0 += (double)r0 * (double)r0; s1 += (double)r1 * (double)r1;
                    }
                }
                if (!grid_sum2(q, round, wave, nact, lane, s0, s1)) { status = 4; break; }
                const double d0 = sqrt(s0 / ntot);
                d1 = sqrt(s1 / ntot);
                h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
                h0 = fmin(h0, span);
                if (!(isfinite(h0) && h0 > 0.0)) { status = 3; break; }
                const float hf = (float)(tdir * h0);
#pragma unroll
                for (int s = 0; s < ZR; ++s) zs[s] = fmaf(hf, k1z[s], z[s]);
                tcur = (float)(t0 + tdir * h0);
                phase = 1;
                continue;
            }
        } else if (phase == 1) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                if (row_live(s)) {
                    const float r = (zd[s] - k1z[s]) / fmaf(fabsf(z[s]), reltol, abstol);
                    s0 += (double)r * (double)r;
                }
            }
            if (scal_live) {
                const float u3[3] = {la, ea, na}, f3[3] = {k1l, k1e, k1n}, g3[3] = {ld, ed, nd};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float r = (g3[i] - f3[i]) / fmaf(fabsf(u3[i]), reltol, abstol);
                    s0 += (double)r * (double)r;
                }
            }
            if (!grid_sum2(q, round, wave, nact, lane, s0, s1)) { status = 4; break; }
            const double d2 = sqrt(s0 / ntot) / h0, dmax = fmax(d1, d2);
            const double h1 = dmax <= 1e-15 ? fmax(1e-6, h0 * 1e-3) : pow(10.0, -(2.0 + log10(dmax)) / 5.0);
            dt = fmin(fmin(100.0 * h0, h1), span);
            if (!(isfinite(dt) && dt > 0.0)) { status = 3; break; }
            begin = true;
        } else {
            // stage st (1..6) of the attempt has been evaluated
            if (a.ckpt_k && st < 6 && naccept < q.ckpt_cap) {   // k_{st+1} of step `naccept` (a retry of the step overwrites it)
#pragma unroll
                for (int s = 0; s < ZR; ++s)
                    a.ckpt_k[((((long long)naccept * 6 + st) * ntiles + tile) * 64 + lane) * ZR + s] = zd[s];
            }
            erl = fmaf(q.bt[st], ld, erl); ere = fmaf(q.bt[st], ed, ere); ern = fmaf(q.bt[st], nd, ern);
#pragma unroll
            for (int s = 0; s < ZR; ++s) erz[s] = fmaf(q.bt[st], zd[s], erz[s]);
            if (st < 6) {
                const float bst = q.b[st];
                lsum = fmaf(bst, ld, lsum); esum = fmaf(bst, ed, esum); nsum = fmaf(bst, nd, nsum);
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) P[i][s] = fmaf(q.acol[st][i], zd[s], P[i + 1][s]);
                    P[5][s] = q.acol[st][5] * zd[s];
                    zs[s] = fmaf(dtf, P[0][s], z[s]);     // the next stage's state; for st + 1 = 6 the new state itself
                }
                ++st;
                tcur = tf + q.c[st] * dtf;
                continue;
            }
            // the attempt is complete: zs = z_new, (zd, ld, ed, nd) = the derivative there
            const float ln = fmaf(dtf, lsum, la), en = fmaf(dtf, esum, ea), nn = fmaf(dtf, nsum, na);
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                if (row_live(s)) {
                    const float r = dtf * erz[s] / fmaf(fmaxf(fabsf(z[s]), fabsf(zs[s])), reltol, abstol);
                    s0 += (double)r * (double)r;
                }
            }
            if (scal_live) {
                const float u3[3] = {la, ea, na}, v3[3] = {ln, en, nn}, e3[3] = {erl, ere, ern};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float r = dtf * e3[i] / fmaf(fmaxf(fabsf(u3[i]), fabsf(v3[i])), reltol, abstol);
                    s0 += (double)r * (double)r;
                }
            }
            if (!grid_sum2(q, round, wave, nact, lane, s0, s1)) { status = 4; break; }
            const double eest = sqrt(s0 / ntot);
            if (!isfinite(eest)) { status = 1; break; }
            const double beta1 = 7.0 / 50.0, beta2 = 2.0 / 25.0, gamma = 0.9, qmin = 0.2, qmax = 10.0;
            const double q11 = eest > 0.0 ? pow(eest, beta1) : 0.0;
            const double qq = eest == 0.0 ? 1.0 / qmax : fmax(1.0 / qmax, fmin(1.0 / qmin, (q11 / pow(qold, beta2)) / gamma));
            if (eest <= 1.0) {   // accept
                t = last ? t1 : t + tdir * step;
#pragma unroll
                for (int s = 0; s < ZR; ++s) { z[s] = zs[s]; k1z[s] = zd[s]; }
                la = ln; ea = en; na = nn;
                k1l = ld; k1e = ed; k1n = nd;
                if (tile == 0 && lane == 0) {
                    if (naccept < q.dts_cap) q.dts[naccept] = (float)(tdir * step);
                    if (q.host_rec && naccept < kHostRec) { q.host_rec[8 + 2 * naccept] = __float_as_int((float)(tdir * step)); q.host_rec[9 + 2 * naccept] = 5; }
                }
                ++naccept;
                qold = fmax(eest, 1e-4);
                dt = step / qq;
            } else {             // reject: same (t, z, k1), smaller step
                ++nreject;
                dt = step / fmin(1.0 / qmin, q11 / gamma);
            }
            begin = true;
        }
        if (begin) {
            if (fabs(t1 - t) <= 1e-7 * fmax(1.0, span)) break;
            if (it >= q.maxiters) { status = 2; break; }
            ++it;
            last = dt >= fabs(t1 - t) * (1.0 - 1e-6);
            step = last ? fabs(t1 - t) : dt;     // tstop: never step over t1
            tf = (float)t;
            dtf = (float)(tdir * step);
            lsum = q.b[0] * k1l; esum = q.b[0] * k1e; nsum = q.b[0] * k1n;
            erl = q.bt[0] * k1l; ere = q.bt[0] * k1e; ern = q.bt[0] * k1n;
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                erz[s] = q.bt[0] * k1z[s];
#pragma unroll
                for (int i = 0; i < 6; ++i) P[i][s] = q.acol[0][i] * k1z[s];
                zs[s] = fmaf(dtf, P[0][s], z[s]);
            }
            if (a.ckpt && naccept < q.ckpt_cap) {   // the step's start state and first stage derivative (first-same-as-last)
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    a.ckpt[(((long long)naccept * ntiles + tile) * 64 + lane) * ZR + s] = z[s];
                    if (a.ckpt_k) a.ckpt_k[((((long long)naccept * 6) * ntiles + tile) * 64 + lane) * ZR + s] = k1z[s];
                }
            }
            st = 1;
            phase = 2;
            tcur = tf + q.c[1] * dtf;
        }
    }
    if (a.ckpt && naccept <= q.ckpt_cap && status == 0) {   // the final state closes the checkpoint list
#pragma unroll
        for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)naccept * ntiles + tile) * 64 + lane) * ZR + s] = z[s];
    }

    if (valid && a.u_out) {
#pragma unroll
        for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) a.u_out[smp * S + f] = z[s]; }
        if (g == 0) { a.u_out[smp * S + D] = la; a.u_out[smp * S + D + 1] = ea; a.u_out[smp * S + D + 2] = na; }
    }
    if (tile == 0 && lane == 0) {
        q.stats[0] = naccept; q.stats[1] = nreject; q.stats[2] = nf; q.stats[3] = status; q.stats[4] = 5;
        if (q.host_rec) {
            q.host_rec[0] = naccept; q.host_rec[1] = nreject; q.host_rec[2] = nf; q.host_rec[3] = status; q.host_rec[4] = 5;
            q.host_rec[5] = status == 4; q.host_rec[6] = (a.ckpt && naccept <= q.ckpt_cap && status == 0) ? 1 : 0; q.host_rec[7] = 0;
        }
    }
}

// occ_out != null: only report how many workgroups of this kernel one compute unit holds (registers and LDS considered)
typedef hipError_t (*LaunchAdaptFn)(const KArgs&, const AArgs&, int lds_bytes, int nblocks, hipStream_t, int* occ_out);

template <int HT, int L, int ZR, int CR, int ACT, int ENGINE, int KP, int PRE, int NTHREADS, int ARITH = 0>
inline hipError_t launch_adapt_inst(const KArgs& a, const AArgs& q, int lds_bytes, int nblocks, hipStream_t st, int* occ_out) {
    auto kern = mfma_adaptive_kernel<HT, L, ZR, CR, ACT, ENGINE, KP, PRE, NTHREADS, ARITH>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (!once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    if (occ_out) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ_out, (const void*)kern, NTHREADS, (size_t)lds_bytes);
    // cooperative launch: the runtime refuses a grid whose workgroups cannot all be resident, which the grid-wide sums need
    KArgs ka = a;
    AArgs qa = q;
    void* params[2] = {&ka, &qa};
    return hipLaunchCooperativeKernel((const void*)kern, dim3(nblocks), dim3(NTHREADS), params, (unsigned)lds_bytes, st);
}

// ---- the reference's default solver VCABM (variable-step variable-order Adams PECE) in one launch ----
// The device passes of cnf_vcabm.hip and the host policy of cnf_api_adaptive.hip::cnf_solve_vcabm, per tile in registers:
// a lane keeps its rows of u_n, f_n and the thirteen modified divided differences Phi*_j(n-1) (HNW I, III.5).  Nothing is
// double-buffered: Phi*_j(n) = beta_j Phi_j(n) is a short chain from f_n and Phi*(n-1), recomputed where the corrector and
// the order-raising estimate need it and committed in place when a step is accepted - a rejected attempt has stored
// nothing.  The step coefficients (beta, g, the error constants) are computed in double by every wave from the step-size
// history, through a per-wave LDS scratch (run-time indexed loops).  One dynamics call site serves all four kinds of
// evaluation (f at t0, Hairer's Euler point, the predictor, the corrected state).  256-thread workgroups: one wave per SIMD,
// the whole register file (the difference table alone is 13 x (ZR + 3) registers).
template <int HT, int L, int ZR, int CR, int ACT, int ENGINE, int KP, int PRE, int NTHREADS, int ARITH = 0>
// (the exact-shape instances of nets of at most two hidden tiles - D <= 4, no conditions: four state rows per lane - keep the whole
// difference table within 256 registers: they may share a SIMD, which doubles the batch the one-launch solve takes - 32 768 samples
// at two workgroups per CU, CNF_DC_PER_CU.  The zero-padded instances (16 state rows) spill 110 - 130 registers under that limit.)
__global__ void __launch_bounds__(NTHREADS)
    __attribute__((amdgpu_waves_per_eu((HT <= 2 && ZR == 1 && CR == 0) ? 2 : 1, (HT <= 2 && ZR == 1 && CR == 0) ? 2 : (NTHREADS + 255) / 256)))
mfma_vcabm_kernel(KArgs a, AArgs q) {
    constexpr MfmaLayout LAY(HT, L, ZR, CR, ENGINE == ENG_VJP, ARITH);
    constexpr int NR = ZR + 3;        // rows of the state a lane holds: ZR of z, then dlogp, E, n
    constexpr int KS = 13;            // Phi*_0 .. Phi*_12
    constexpr int WPB = NTHREADS / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double sc_hist[WPB][KS + 1];   // signed sizes of the accepted steps, newest first
    __shared__ double sc_dts[WPB][KS + 3], sc_cq[WPB][KS + 3], sc_gd[WPB][KS + 1], sc_gs[WPB][KS + 2];
    __shared__ float sc_beta[WPB][KS], sc_g[WPB][KS + 1];
    stage_image<NTHREADS>(a.packed, smem, LAY.lds_total / 4);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long ntiles = (a.B + 15) / 16;
    const long long tile = (long long)blockIdx.x + (long long)gridDim.x * wave;
    if (tile >= ntiles) return;
    const int nact = (int)((ntiles - 1 - (long long)blockIdx.x) / (long long)gridDim.x) + 1 < WPB
                         ? (int)((ntiles - 1 - (long long)blockIdx.x) / (long long)gridDim.x) + 1 : WPB;
    const int D = a.D, S = D + 3, C = a.C;
    const int K = KP == 1 ? 1 : a.K;
    const int Kd = K * D;
    const bool reg_z = a.reg_z, reg_j = a.reg_j, autonomous = a.autonomous, exact = a.exact;
    const long long smp = tile * 16 + n;
    const bool valid = smp < a.B;
    const long long sc = valid ? smp : a.B - 1;
    float U[NR], F[NR], PS[KS][NR], Pp[NR], UN[NR], X[NR];
    float eps[KP][ZR], y[CR > 0 ? CR : 1];
    bool live[NR];                    // the entries of the S x B state this lane answers for in a norm
#pragma unroll
    for (int s = 0; s < ZR; ++s) {
        const int f = 4 * s + g;
        U[s] = f < D ? a.u0[sc * S + f] : 0.f;
        live[s] = valid && f < D;
#pragma unroll
        for (int p = 0; p < KP; ++p) eps[p][s] = (f < D && p < K && a.eps) ? a.eps[sc * Kd + p * D + f] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { U[ZR + r] = a.u0[sc * S + D + r]; live[ZR + r] = valid && g == 0; }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        F[r] = Pp[r] = UN[r] = X[r] = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j) PS[j][r] = 0.f;
    }
    y[0] = 0.f;
    if constexpr (CR > 0) {
#pragma unroll
        for (int s = 0; s < CR; ++s) { const int f = 4 * s + g; y[s] = f < C ? a.ys[sc * C + f] : 0.f; }
    }
    f32x4 pre_c[HT], pre_q[HT];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) pre_c[mt] = pre_q[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PRE >= 1 && KP == 1) gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps[0]}, pre_c);
    // several probes: c_k = W_N^T eps_k of every probe, once per tile (see vjp_probe_hoisted)
    static_assert(PRE == 0 || ENGINE == ENG_VJP, "hoisting is a VJP-engine feature");
    static_assert(PRE < 2 || KP == 1, "the dot-product shortcut (PRE = 2) is instantiated for one probe");
    f32x4 pre_ck[(PRE >= 1 && KP > 1) ? KP : 1][HT];
    if constexpr (PRE >= 1 && KP > 1)
        static_for<0, KP>([&](auto pi) {
            constexpr int p = decltype(pi)::value;
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) pre_ck[p][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps[p]}, pre_ck[p]);
        });
    if constexpr (PRE >= 2) {
        gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{eps[0]}, pre_q);
        if constexpr (ACT == CNF_ACT_TANH_PRESCALED) {
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) pre_q[mt] *= (1.f / kTanhPrescale);
        }
    }
    double* hist = sc_hist[wave];
    double *dtsv = sc_dts[wave], *cq = sc_cq[wave], *gd = sc_gd[wave], *gs = sc_gs[wave];
    float *beta = sc_beta[wave], *gg = sc_g[wave];
    for (int i = 0; i <= KS; ++i) hist[i] = 0.0;

    const double t0 = (double)a.t0, t1 = (double)q.t1;
    const double span = fabs(t1 - t0), tdir = t1 >= t0 ? 1.0 : -1.0, ntot = (double)S * (double)a.B;
    const float abstol = q.abstol, reltol = q.reltol;
    const double gamma = 0.9, qmin = 0.2, qmax = 10.0;
    double tpol = t0, tvc = t0, dt = 0.0, hstep = 0.0, h0 = 0.0, d1 = 0.0, eest = 0.0;
    float dtf = 0.f, tcur = a.t0, e0 = 0.f, e1 = 0.f, e2 = 0.f;
    int k = 1, m = 0, step = 1, nhist = 0, it = 0, naccept = 0, nreject = 0, nf = 0, status = 0, max_order = 0;
    bool last = false, select = false, lower = false, want_up = false;
    int phase = 0;                    // 0: f(u0, t0); 1: Hairer's Euler point; 2: predictor; 3: corrected state
    unsigned round = 0;
    float zs[ZR], zd[ZR], ld = 0.f, ed = 0.f, nd = 0.f;
#pragma unroll
    for (int s = 0; s < ZR; ++s) { zs[s] = U[s]; zd[s] = 0.f; }

    // scaled squared norm pieces: r = v / (abstol + |ref| reltol)
    auto sk_of = [&](float ref) { return fmaf(fabsf(ref), reltol, abstol); };
    // x^y for x > 0 as exp(y log x): a third of the instructions of pow() (no special cases, no extended-precision logarithm), a few
    // ulp - the step-size factor it feeds is clamped to [1/qmax, 1/qmin] and the step it scales rounded to float
    auto pow_pos = [](double x, double y) { return exp(y * log(x)); };

#ifdef VC_TRACE
    unsigned long long vc_t[6] = {0, 0, 0, 0, 0, 0}, vc_s = 0, vc_k0 = __builtin_amdgcn_s_memtime();   // dyn, sums, begin, phase 2, phase 3, rest
#define VC_ON() do { asm volatile("" ::: "memory"); vc_s = __builtin_amdgcn_s_memtime(); } while (0)
#define VC_OFF(k) do { asm volatile("" ::: "memory"); vc_t[k] += __builtin_amdgcn_s_memtime() - vc_s; } while (0)
#else
#define VC_ON() do {} while (0)
#define VC_OFF(k) do {} while (0)
#endif
    for (;;) {
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        VC_ON();
        dyn_eval<HT, L, ZR, CR, ACT, ENGINE, KP, PRE, ARITH>(smem + opaque, a.packed + opaque, lane, tcur, autonomous, reg_z, reg_j, exact,
                                                      D, K, zs, y, eps, pre_c, pre_q, zd, ld, ed, nd, pre_ck);
        VC_OFF(0);
        ++nf;
#pragma unroll
        for (int s = 0; s < ZR; ++s) X[s] = zd[s];
        X[ZR] = ld; X[ZR + 1] = ed; X[ZR + 2] = nd;
        bool begin = false;
        if (phase == 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) F[r] = X[r];
            if (q.dt_init != 0.f) {
                dt = fmin((double)fabsf(q.dt_init), span);
                begin = true;
            } else {   // ode_determine_initdt, RMS norm over all S B entries; exponent 1 / (current order) = 1 at the start
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (live[r]) {
                        const float sk = sk_of(U[r]), r0 = U[r] / sk, r1 = F[r] / sk;
                        s0 += (double)r0 * (double)r0; s1 += (double)r1 * (double)r1;
                    }
                }
                VC_ON();
                const bool gs_ok = grid_sum3(q, round, wave, nact, lane, s0, s1, s2);
                VC_OFF(1);
                if (!gs_ok) { status = 4; break; }
                const double d0 = sqrt(s0 / ntot);
                d1 = sqrt(s1 / ntot);
                h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
                h0 = fmin(h0, span);
                const float hf = (float)(tdir * h0);
#pragma unroll
                for (int s = 0; s < ZR; ++s) zs[s] = fmaf(hf, F[s], U[s]);
                tcur = (float)(t0 + tdir * h0);
                phase = 1;
                continue;
            }
        } else if (phase == 1) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (live[r]) { const float rr = (X[r] - F[r]) / sk_of(U[r]); s0 += (double)rr * (double)rr; }
            }
            VC_ON();
            const bool gs_ok = grid_sum3(q, round, wave, nact, lane, s0, s1, s2);
            VC_OFF(1);
            if (!gs_ok) { status = 4; break; }
            const double d2 = sqrt(s0 / ntot) / h0, dmax = fmax(d1, d2);
            const double h1 = dmax <= 1e-15 ? fmax(1e-6, h0 * 1e-3) : pow(10.0, -(2.0 + log10(dmax)) / 1.0);
            dt = fmin(fmin(100.0 * h0, h1), span);
            if (!(isfinite(dt) && dt > 0.0)) { status = 3; break; }
            begin = true;
        } else if (phase == 2) {
            // C: Phi_j(n+1) from d = f(p, t + dt); u_new = p + dt g_k Phi_k(n+1); error sums of orders k, k-1, k-2
            VC_ON();
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            const float dg = dtf * gg[k];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float phiF = F[r], psn = phiF, phi = X[r], phim1 = 0.f, phim2 = 0.f;
#pragma unroll
                for (int j = 1; j < KS; ++j) {
                    if (j <= k) {
                        phim2 = phim1; phim1 = phi; phi -= psn;                   // psn = Phi*_{j-1}(n)
                        if (j < m) { phiF -= PS[j - 1][r]; psn = beta[j] * phiF; }
                    }
                }
                UN[r] = fmaf(dg, phi, Pp[r]);
                if (live[r]) {
                    const float inv = 1.f / fmaf(fmaxf(fabsf(U[r]), fabsf(UN[r])), reltol, abstol);
                    const float r0 = e0 * phi * inv, r1 = e1 * phim1 * inv, r2 = e2 * phim2 * inv;
                    s0 += (double)r0 * (double)r0; s1 += (double)r1 * (double)r1; s2 += (double)r2 * (double)r2;
                }
            }
            VC_OFF(3);
            VC_ON();
            const bool gs_ok = grid_sum3(q, round, wave, nact, lane, s0, s1, s2);
            VC_OFF(1);
            if (!gs_ok) { status = 4; break; }
            eest = sqrt(s0 / ntot);
            if (!isfinite(eest)) { status = 1; break; }
            if (eest > 1.0) {   // reject: same state, smaller step, same order (nothing was stored)
                ++nreject;
                dt = hstep / fmax(1.0 / qmax, fmin(1.0 / qmin, pow_pos(eest, 1.0 / (double)(k + 1)) / gamma));
                begin = true;
            } else {
                select = step > 4 && k >= 3;
                lower = select && fmax(sqrt(s2 / ntot), sqrt(s1 / ntot)) <= eest;
                want_up = select && !lower && k < 12;
#pragma unroll
                for (int s = 0; s < ZR; ++s) zs[s] = UN[s];
                tcur = (float)(tvc + (double)dtf);
                phase = 3;
                continue;
            }
        } else {
            // accepted: X = f(u_new, t + dt).  Order k + 1 estimate, then commit Phi*(n), u, f, the history
            VC_ON();
            int knew = k;
            if (!select) knew = k + 1 < 3 ? k + 1 : 3;
            else if (lower) knew = k - 1;
            else if (want_up) {
                double gs_up;
                if (k <= 4) {   // register form of the loop below (see the step coefficients further down)
                    double gr[6];
                    gr[0] = 1.0;
#pragma unroll
                    for (int j = 1; j <= 5; ++j) {
                        gr[j] = 0.0;
                        if (j <= k + 1) {
                            double acc = 0.0;
#pragma unroll
                            for (int i = 0; i < j; ++i) acc += gr[i] / (double)(j - i + 1);
                            gr[j] = -acc;
                        }
                    }
                    gs_up = k == 4 ? gr[5] : (k == 3 ? gr[4] : (k == 2 ? gr[3] : (k == 1 ? gr[2] : gr[1])));
                } else {
                    gs[0] = 1.0;
                    for (int j = 1; j <= k + 1; ++j) {
                        double acc = 0.0;
                        for (int i = 0; i < j; ++i) acc += gs[i] / (double)(j - i + 1);
                        gs[j] = -acc;
                    }
                    gs_up = gs[k + 1];
                }
                const float eu = (float)((double)dtf * gs_up);
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    float phiF = F[r], psn = phiF, phi = X[r];
#pragma unroll
                    for (int j = 0; j < KS; ++j) {
                        if (j <= k) {
                            phi -= psn;                                          // psn = Phi*_j(n)
                            if (j + 1 < m) { phiF -= PS[j][r]; psn = beta[j + 1] * phiF; }
                        }
                    }
                    if (live[r]) {
                        const float rr = eu * phi / fmaf(fmaxf(fabsf(U[r]), fabsf(UN[r])), reltol, abstol);
                        s0 += (double)rr * (double)rr;
                    }
                }
                VC_OFF(4);
                VC_ON();
                const bool gs_ok = grid_sum3(q, round, wave, nact, lane, s0, s1, s2);
                VC_OFF(1);
                if (!gs_ok) { status = 4; break; }
                if (sqrt(s0 / ntot) < eest) { knew = k + 1; eest = 1.0; }
                VC_ON();
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float phi = F[r], keep = PS[0][r];
                PS[0][r] = phi;
#pragma unroll
                for (int j = 1; j < KS; ++j) {
                    if (j < m) { phi -= keep; keep = PS[j][r]; PS[j][r] = beta[j] * phi; }
                }
                U[r] = UN[r];
                F[r] = X[r];
            }
            for (int i = KS; i > 0; --i) hist[i] = hist[i - 1];
            hist[0] = (double)dtf;
            tvc += (double)dtf;
            nhist += 1;
            const double qq = eest == 0.0 ? 1.0 / qmax : fmax(1.0 / qmax, fmin(1.0 / qmin, pow_pos(eest, 1.0 / (double)(knew + 1)) / gamma));
            tpol = last ? t1 : tpol + tdir * hstep;
            if (tile == 0 && lane == 0) {
                if (naccept < q.dts_cap) { q.dts[naccept] = dtf; q.orders[naccept] = k; }
                if (q.host_rec && naccept < kHostRec) { q.host_rec[8 + 2 * naccept] = __float_as_int(dtf); q.host_rec[9 + 2 * naccept] = k; }
            }
            ++naccept;
            if (k > max_order) max_order = k;
            k = knew; ++step;
            dt = hstep / qq;
            begin = true;
            VC_OFF(4);
        }
        if (begin) {
            VC_ON();
            if (fabs(t1 - tpol) <= 1e-7 * fmax(1.0, span)) break;
            if (it >= q.maxiters) { status = 2; break; }
            ++it;
            last = dt >= fabs(t1 - tpol) * (1.0 - 1e-6);
            hstep = last ? fabs(t1 - tpol) : dt;      // tstop: never step over t1
            dtf = (float)(tdir * hstep);
            m = k + 1 < nhist + 1 ? k + 1 : nhist + 1;
            const int ng = k + 1;
            if (k <= 4) {
                // Orders 1 .. 4 (what a default-tolerance solve runs at): the recurrences of the general form below, operation for
                // operation, on register arrays - static indices, wave-uniform branches.  The run-time indexed LDS loops cost a
                // one-wave kernel ~6 k cycles per step (every access a round trip nothing covers), a third of the step.
                const double hh[5] = {(double)dtf, hist[0], hist[1], hist[2], hist[3]};   // step sizes newest first, the candidate in front
                double bb = 1.0, num = 0.0, den = 0.0;
                beta[0] = 1.f;
#pragma unroll
                for (int j = 1; j < 5; ++j) {
                    if (j < m) { num += hh[j - 1]; den += hh[j]; bb *= num / den; beta[j] = (float)bb; }
                }
                double c[5], gr[5];
#pragma unroll
                for (int qi = 1; qi <= 5; ++qi) c[qi - 1] = 1.0 / ((double)qi * (double)(qi + 1));
                gr[0] = 1.0;
                double xi = hh[0];
#pragma unroll
                for (int j = 1; j < 5; ++j) {
                    gr[j] = 0.0;
                    if (j < ng) {
                        if (j > 1) {
                            xi += hh[j - 1];
#pragma unroll
                            for (int qi = 0; qi < 6 - j; ++qi) {
                                if (qi < ng - j + 1) c[qi] = c[qi] - c[qi + 1] * (double)dtf / xi;
                            }
                        }
                        gr[j] = c[0];
                    }
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    if (j < ng) gg[j] = (float)gr[j];
                }
                const double gk = k == 4 ? gr[4] : (k == 3 ? gr[3] : (k == 2 ? gr[2] : gr[1]));
                const double gk1 = k == 4 ? gr[3] : (k == 3 ? gr[2] : (k == 2 ? gr[1] : gr[0]));
                const double gk2 = k == 4 ? gr[2] : (k == 3 ? gr[1] : gr[0]);
                const double gk3 = k == 4 ? gr[1] : gr[0];
                e0 = (float)((double)dtf * (gk - gk1));
                e1 = k >= 2 ? (float)((double)dtf * (gk1 - gk2)) : 0.f;
                e2 = k >= 3 ? (float)((double)dtf * (gk2 - gk3)) : 0.f;
            } else {
                // step sizes newest first, the candidate in front
                dtsv[0] = (double)dtf;
                for (int i = 0; i <= KS; ++i) dtsv[i + 1] = hist[i];
                double bb = 1.0, num = 0.0, den = 0.0;
                beta[0] = 1.f;
                for (int j = 1; j < m; ++j) { num += dtsv[j - 1]; den += dtsv[j]; bb *= num / den; beta[j] = (float)bb; }
                gd[0] = 1.0;
                for (int qi = 1; qi <= ng; ++qi) cq[qi - 1] = 1.0 / ((double)qi * (double)(qi + 1));
                double xi = dtsv[0];
                for (int j = 1; j < ng; ++j) {
                    if (j > 1) {
                        xi += dtsv[j - 1];
                        for (int qi = 0; qi < ng - j + 1; ++qi) cq[qi] = cq[qi] - cq[qi + 1] * (double)dtf / xi;
                    }
                    gd[j] = cq[0];
                }
                for (int j = 0; j < ng; ++j) gg[j] = (float)gd[j];
                e0 = (float)((double)dtf * (gd[k] - gd[k - 1]));
                e1 = k >= 2 ? (float)((double)dtf * (gd[k - 1] - gd[k - 2])) : 0.f;
                e2 = k >= 3 ? (float)((double)dtf * (gd[k - 2] - gd[k - 3])) : 0.f;
            }
            // P: p = u + dt sum_{j<k} g_j Phi*_j(n)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float phi = F[r], acc = gg[0] * phi;
#pragma unroll
                for (int j = 1; j < KS; ++j) {
                    if (j < m) {
                        phi -= PS[j - 1][r];
                        const float sj = beta[j] * phi;
                        if (j < k) acc = fmaf(gg[j], sj, acc);
                    }
                }
                Pp[r] = fmaf(dtf, acc, U[r]);
            }
#pragma unroll
            for (int s = 0; s < ZR; ++s) zs[s] = Pp[s];
            tcur = (float)(tvc + (double)dtf);
            phase = 2;
            VC_OFF(2);
        }
    }
#ifdef VC_TRACE
    if (tile == 0 && lane == 0) {
        const unsigned long long tot = __builtin_amdgcn_s_memtime() - vc_k0;
        printf("VC_TRACE steps %d nf %d total %llu | dyn %llu sums %llu begin %llu ph2 %llu ph3 %llu rest %llu\n", naccept, nf, tot, vc_t[0], vc_t[1], vc_t[2], vc_t[3], vc_t[4],
               tot - vc_t[0] - vc_t[1] - vc_t[2] - vc_t[3] - vc_t[4]);
    }
#endif

    if (valid && a.u_out) {
#pragma unroll
        for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) a.u_out[smp * S + f] = U[s]; }
        if (g == 0) { a.u_out[smp * S + D] = U[ZR]; a.u_out[smp * S + D + 1] = U[ZR + 1]; a.u_out[smp * S + D + 2] = U[ZR + 2]; }
    }
    if (tile == 0 && lane == 0) {
        q.stats[0] = naccept; q.stats[1] = nreject; q.stats[2] = nf; q.stats[3] = status; q.stats[4] = max_order;
        if (q.host_rec) {
            q.host_rec[0] = naccept; q.host_rec[1] = nreject; q.host_rec[2] = nf; q.host_rec[3] = status; q.host_rec[4] = max_order;
            q.host_rec[5] = status == 4; q.host_rec[6] = 0; q.host_rec[7] = 0;
        }
    }
}

template <int HT, int L, int ZR, int CR, int ACT, int ENGINE, int KP, int PRE, int NTHREADS, int ARITH = 0>
inline hipError_t launch_vcabm_inst(const KArgs& a, const AArgs& q, int lds_bytes, int nblocks, hipStream_t st, int* occ_out) {
    auto kern = mfma_vcabm_kernel<HT, L, ZR, CR, ACT, ENGINE, KP, PRE, NTHREADS, ARITH>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (!once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    if (occ_out) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ_out, (const void*)kern, NTHREADS, (size_t)lds_bytes);
    KArgs ka = a;
    AArgs qa = q;
    void* params[2] = {&ka, &qa};
    return hipLaunchCooperativeKernel((const void*)kern, dim3(nblocks), dim3(NTHREADS), params, (unsigned)lds_bytes, st);
}
#else
typedef hipError_t (*LaunchAdaptFn)(const KArgs&, const AArgs&, int lds_bytes, int nblocks, hipStream_t, int* occ_out);
#endif

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
    LaunchAdaptFn fn_adapt;   // adaptive Tsit5 with the device-side step controller, or null (host loop)
    LaunchAdaptFn fn_vcabm;   // the default solver VCABM with policy and passes on the device (256-thread workgroups), or null
};

#define MFMA_INST(HT, L, ZR, CR, ACT, ENG, KP, PRE, NT) \
    Inst { HT, L, ZR, CR, ACT, ENG, KP, PRE, &launch_inst<HT, L, ZR, CR, ACT, ENG, KP, PRE, NT>, NT, 0, nullptr, nullptr }
#define MFMA_INST_BF16X6(HT, L, ZR, CR, ACT, ENG, KP, PRE, NT) \
    Inst { HT, L, ZR, CR, ACT, ENG, KP, PRE, &launch_inst<HT, L, ZR, CR, ACT, ENG, KP, PRE, NT, 1>, NT, 1, nullptr, nullptr }
#ifdef CNF_WITH_DEVICE_CONTROLLER
// the same instance with its device-controlled adaptive twin
#define MFMA_INST_AD(HT, L, ZR, CR, ACT, ENG, KP, PRE, NT) \
    Inst { HT, L, ZR, CR, ACT, ENG, KP, PRE, &launch_inst<HT, L, ZR, CR, ACT, ENG, KP, PRE, NT>, NT, 0, \
           &launch_adapt_inst<HT, L, ZR, CR, ACT, ENG, KP, PRE, NT>, &launch_vcabm_inst<HT, L, ZR, CR, ACT, ENG, KP, PRE, 256> }
#endif

// generic zero-padded instances (cnf_mfma_generic.hip): D <= 16 and C <= 16 or C = 0
const Inst* mfma_generic_insts(int* count);            // tanh nets
const Inst* mfma_generic_softplus_insts(int* count);   // softplus nets (cnf_mfma_generic_softplus.hip)
// the same shapes with a capacity of several Hutchinson probes (cnf_mfma_generic_probes.hip): VJP, K <= KP
const Inst* mfma_generic_probe_insts(int* count);
// state k-steps padded to 8 (D <= 32): the reference's default nets for nvariables 8..15 (cnf_mfma_generic_zr8.hip)
const Inst* mfma_generic_zr8_insts(int* count);


}  // namespace cnf
