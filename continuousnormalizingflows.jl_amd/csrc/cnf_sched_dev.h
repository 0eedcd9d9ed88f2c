// cnf_sched_dev.h - device helpers of the hand-scheduled per-wave kernels (cnf_grad2.hip: the register-accumulator reverse sweep;
// cnf_mfma2.hip: the fixed-step solve of one-probe VJP flows): a scheduling fence, fragment loads and the product loop that keeps
// them one k-group ahead of the MFMAs and runs a hook behind every k-step.
#pragma once
#include "cnf_mfma_kernel.h"

namespace cnf {

// Scheduling fence: nothing crosses.  The stage below is laid out by hand as MFMA runs / VALU phases / LDS bursts; left to itself the
// scheduler sank every fragment load to its first use (one exposed LDS round trip per 16 MFMAs: ~5 k of the 36 k cycles of a stage in
// the first build of this file, profiles/r5/r5b_cfg2_grad2_phase_trace.txt) and interleaved the activations with the products (f32
// MFMAs hide no VALU work on gfx950 and every MFMA -> VALU -> MFMA round trip costs ~9 issue cycles, cnf_mfma_kernel.h::phase_fence).
#define G2_FENCE() __builtin_amdgcn_sched_barrier(0)

// fragments of k-group kg of an A image with KG k-groups: one ds_read_b128 per M-tile - one ds_read_b64 where only NJ <= 2 of the
// group's k-steps are multiplied (D <= 8: with the full read the compiler overlaps the destination registers of consecutive tiles
// in their unused halves and serialises the reads behind lgkmcnt(0))
template <int MT, int NJ = 4>
__device__ __forceinline__ void afrag(const float* img, int lane, int KG, int kg, f32x4 (&a)[MT]) {
    const f32x4* A = reinterpret_cast<const f32x4*>(img) + lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if constexpr (NJ <= 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(&A[(mt * KG + kg) * 64]);
            a[mt] = f32x4{v[0], v[1], 0.f, 0.f};
        } else {
            a[mt] = A[(mt * KG + kg) * 64];
        }
    }
}

// gemm_tiles with the fragment loads one k-group AHEAD of the MFMAs that use them: `pre` holds k-group 0 on entry (requested by the
// previous product); behind the first k-step of the last k-group, k-group 0 of the NEXT product (image nimg, NKG k-groups, NMT tiles)
// is requested into `npre`.  One wave per SIMD: there is no other wave to cover an LDS round trip.
// `hook(q)` runs behind the MT MFMAs of k-step q: the place for the few LDS instructions a phase needs besides the fragments - the
// cotangent operands' tile stores and transposed fragment reads.  A wave issues in order, and the matrix pipe stays busy for 32 cycles
// behind an MFMA: an LDS or scalar instruction issued in that shadow is free (two ds_read_b128 per gap, MI355X_MICROARCH.md), the same
// instructions in one burst between two products stop the pipe (40 of them per product in the second build: LDS issue 2.1 k of a
// stage's 37 k cycles, profiles/r5/r5c_cfg2_grad2_pmc.txt).
struct NoHook {
    template <int Q>
    __device__ __forceinline__ void operator()(std::integral_constant<int, Q>) const {}
};
template <int MT, int KS, int NMT, int NKS = 4, typename InT, typename Hook = NoHook>
__device__ __forceinline__ void gemm_pf(const float* img, int lane, const InT& in, const f32x4 (&pre)[MT], f32x4 (&acc)[MT],
                                        const float* nimg, int NKG, f32x4 (&npre)[NMT], Hook&& hook = NoHook{}) {
    constexpr int KG = (KS + 3) / 4;
    constexpr int LASTJ = KS - 4 * (KG - 1);   // k-steps of the last k-group
    f32x4 cur[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cur[mt] = pre[mt];
    static_for<0, KG>([&](auto kgc) {
        constexpr int kg = decltype(kgc)::value;
        f32x4 nxt[MT];
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value, q = kg * 4 + j;
            if constexpr (q < KS) {
                const float b = in(q);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma4(cur[mt][j], b, acc[mt]);
                G2_FENCE();
                if constexpr (j == 0) {
                    if constexpr (kg + 1 < KG) afrag<MT, (kg + 2 == KG ? LASTJ : 4)>(img, lane, KG, kg + 1, nxt);
                    else if (nimg) afrag<NMT, (NKS < 4 ? NKS : 4)>(nimg, lane, NKG, 0, npre);
                }
                hook(std::integral_constant<int, q>{});
                G2_FENCE();
            }
        });
        if constexpr (kg + 1 < KG) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) cur[mt] = nxt[mt];
        }
    });
}


// C vectors (bias, time column) of a stage from ONE lane base: smg = (image base) + 4 g floats, computed once per stage; the vector's
// and the tile's offsets ride in the instruction's immediate (load_cvec's address is re-derived per tile: four VALU adds per vector)
template <int MT>
__device__ __forceinline__ void load_cvec_g(const float* smg, int vec_off, f32x4 (&out)[MT]) {
    const f32x4* v = reinterpret_cast<const f32x4*>(smg);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) out[mt] = v[vec_off / 4 + mt * 4];
}

// tanh (and optionally tanh' = 1 - h^2) of FOUR accumulator tiles, stage by stage over all 16 values - 16 v_exp, the +1 on register
// pairs, 16 v_rcp, h = 2 r - 1 and d = 1 - h^2 on register pairs - with ONE wait-state statement per packed step instead of one per
// tile (act_tile: 2 x 4 per layer): the same operations on the same values, so the same bits.  PRESCALED: the pre-activation
// arrives multiplied by -2 log2(e) (folded into the forward images).
template <bool PRESCALED, bool WITH_D>
__device__ __forceinline__ void tanh_tiles4(const f32x4 (&a)[4], f32x4 (&h)[4], f32x4 (&d)[4]) {
    f32x2 e[8];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        f32x2 x0 = {a[mt][0], a[mt][1]}, x1 = {a[mt][2], a[mt][3]};
        if constexpr (!PRESCALED) { x0 = x0 * kTanhPrescale; x1 = x1 * kTanhPrescale; }
        e[2 * mt] = f32x2{__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1])};
        e[2 * mt + 1] = f32x2{__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1])};
    }
#define CNF_PKADD1(i) "v_pk_add_f32 %" #i ", %" #i ", 1.0 op_sel_hi:[1,0]\n\t"
    asm volatile("s_nop 0\n\t" CNF_PKADD1(0) CNF_PKADD1(1) CNF_PKADD1(2) CNF_PKADD1(3) CNF_PKADD1(4) CNF_PKADD1(5) CNF_PKADD1(6) CNF_PKADD1(7)
                 : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]));
#undef CNF_PKADD1
    f32x2 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = f32x2{fast_rcp(e[i][0]), fast_rcp(e[i][1])};
    f32x2 hh[8];
#define CNF_PKH(o, i) "v_pk_fma_f32 %" #o ", %" #i ", 2.0, -1.0 op_sel_hi:[1,0,0]\n\t"
    asm volatile("s_nop 0\n\t" CNF_PKH(0, 8) CNF_PKH(1, 9) CNF_PKH(2, 10) CNF_PKH(3, 11) CNF_PKH(4, 12) CNF_PKH(5, 13) CNF_PKH(6, 14) CNF_PKH(7, 15)
                 : "=&v"(hh[0]), "=&v"(hh[1]), "=&v"(hh[2]), "=&v"(hh[3]), "=&v"(hh[4]), "=&v"(hh[5]), "=&v"(hh[6]), "=&v"(hh[7])
                 : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
#undef CNF_PKH
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) h[mt] = f32x4{hh[2 * mt][0], hh[2 * mt][1], hh[2 * mt + 1][0], hh[2 * mt + 1][1]};
    if constexpr (WITH_D) {
        f32x2 dd[8];
#define CNF_PKD(o, i) "v_pk_fma_f32 %" #o ", %" #i ", %" #i ", 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        asm volatile(CNF_PKD(0, 8) CNF_PKD(1, 9) CNF_PKD(2, 10) CNF_PKD(3, 11) CNF_PKD(4, 12) CNF_PKD(5, 13) CNF_PKD(6, 14) CNF_PKD(7, 15)
                     : "=&v"(dd[0]), "=&v"(dd[1]), "=&v"(dd[2]), "=&v"(dd[3]), "=&v"(dd[4]), "=&v"(dd[5]), "=&v"(dd[6]), "=&v"(dd[7])
                     : "v"(hh[0]), "v"(hh[1]), "v"(hh[2]), "v"(hh[3]), "v"(hh[4]), "v"(hh[5]), "v"(hh[6]), "v"(hh[7]));
#undef CNF_PKD
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) d[mt] = f32x4{dd[2 * mt][0], dd[2 * mt][1], dd[2 * mt + 1][0], dd[2 * mt + 1][1]};
    }
}

}  // namespace cnf
