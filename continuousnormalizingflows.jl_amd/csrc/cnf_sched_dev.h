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


}  // namespace cnf
