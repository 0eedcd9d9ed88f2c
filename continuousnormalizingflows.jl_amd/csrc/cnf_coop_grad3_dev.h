// cnf_coop_grad3_dev.h - device helpers shared by the second-order reverse sweeps of the cooperative gradient's second form
// (cnf_coop_grad3.hip: one workgroup per CU with the whole register file; cnf_coop_grad3w.hip: two workgroups per CU at 256
// registers per wave): the tile units of a wave, fragment loads, the product over k-groups with its placed-access hook, G.
#pragma once
#define CNF_NO_PK_ASM 1
#define CNF_NO_PHASE_FENCE 1
#include "cnf_coop_d_dev.h"
#include "cnf_tiles.h"

#define G3_SYNC()                                                       \
    do {                                                                \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
        __builtin_amdgcn_s_barrier();                                   \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
    } while (0)

// -DG3_TRACE: s_memtime at every phase boundary of one stage of one workgroup, printed by every wave (a measuring build)
#ifdef G3_TRACE
#define G3_T(k) do { asm volatile("" ::: "memory"); tr[k] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#else
#define G3_T(k)
#endif

namespace cnf {

struct G3Args {
    CG3Args c;
    DImg g;
};

namespace {

template <int A, int NC>
struct G3U {                 // this wave's units of one chain: shared tiles x NC sample tiles, the left-over tile x NC sample tiles
    f32x4 S[A][NC];
    f32x4 R[NC];
};
template <int A>
struct G3Off { unsigned S[A]; unsigned Rr; };

template <int A>
__device__ __forceinline__ G3Off<A> g3_offsets(const DRs& R, int KP, int mtS0, int tR) {
    G3Off<A> t;
#pragma unroll
    for (int m = 0; m < A; ++m) { t.S[m] = R.lane16 + (unsigned)((mtS0 + m) * KP) * 1024u; asm volatile("" : "+v"(t.S[m])); }
    t.Rr = R.lane16 + (unsigned)(tR * KP) * 1024u; asm volatile("" : "+v"(t.Rr));
    return t;
}
template <int A, bool LO>
__device__ __forceinline__ void g3_load_a(const DRs& R, const G3Off<A>& T, unsigned img, int kg, f32x4 (&aS)[A], f32x4& aR) {
    const unsigned so = img + (unsigned)kg * 1024u;
#pragma unroll
    for (int m = 0; m < A; ++m) aS[m] = dloadv(R, T.S[m], so);
    if constexpr (LO) aR = dloadv(R, T.Rr, so);
}
template <int NC>
__device__ __forceinline__ void g3_load_b(const f32x4* __restrict__ bimg, int kg, int lane, f32x4 (&bq)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) bq[c] = bimg[(kg * NC + c) * 64 + lane];
}
template <int A, bool LO, int JN, int NC>
__device__ __forceinline__ void g3_mfma(const f32x4 (&aS)[A], const f32x4& aR, const f32x4 (&bq)[NC], bool v0, G3U<A, NC>& u) {
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) u.S[m][c] = mfma4(aS[m][j], bq[c][j], u.S[m][c]);
    if constexpr (LO) {
        if (v0) {
#pragma unroll
            for (int j = 0; j < JN; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) u.R[c] = mfma4(aR[j], bq[c][j], u.R[c]);
        }
    }
}
template <int A, bool LO, int NC>
__device__ __forceinline__ void g3_mfma_rem(const f32x4 (&aS)[A], const f32x4& aR, const f32x4 (&bq)[NC], bool v0, int rem, G3U<A, NC>& u) {
    if (rem == 4) g3_mfma<A, LO, 4, NC>(aS, aR, bq, v0, u);
    else if (rem == 3) g3_mfma<A, LO, 3, NC>(aS, aR, bq, v0, u);
    else if (rem == 2) g3_mfma<A, LO, 2, NC>(aS, aR, bq, v0, u);
    else g3_mfma<A, LO, 1, NC>(aS, aR, bq, v0, u);
}
// u += A(image) * B(LDS image: [k-group][NC column tiles][64 lanes]) over KG k-groups, the last one with `rem` k-steps; aS0 / aR0
// arrive holding the fragments of k-group 0.  Two fragment sets ping-pong, one k-group of lead.
// `hook()` runs right behind the product's LAST fragment request: what is issued there has nothing younger of this product waiting
// behind it in the in-order memory counter, and travels under the product's last k-groups and the phase that follows.
template <int A, bool LO, int NC, class Hook>
__device__ __forceinline__ void g3_gemm(const DRs& R, const G3Off<A>& T, unsigned img, int KG, int rem, bool v0,
                                        const f32x4* __restrict__ bimg, int lane, f32x4 (&aS0)[A], f32x4& aR0, G3U<A, NC>& u, Hook&& hook) {
    f32x4 aS1[A], aR1 = {0.f, 0.f, 0.f, 0.f}, bq0[NC], bq1[NC];
    g3_load_b<NC>(bimg, 0, lane, bq0);
    const int KGf = KG - 1;
    int kg = 0;
#pragma clang loop unroll(disable)
    for (; kg + 2 <= KGf; kg += 2) {
        g3_load_a<A, LO>(R, T, img, kg + 1, aS1, aR1);
        g3_load_b<NC>(bimg, kg + 1, lane, bq1);
        g3_mfma<A, LO, 4, NC>(aS0, aR0, bq0, v0, u);
        g3_load_a<A, LO>(R, T, img, kg + 2, aS0, aR0);
        g3_load_b<NC>(bimg, kg + 2, lane, bq0);
        g3_mfma<A, LO, 4, NC>(aS1, aR1, bq1, v0, u);
    }
    if (kg < KGf) {
        g3_load_a<A, LO>(R, T, img, KG - 1, aS1, aR1);
        g3_load_b<NC>(bimg, KG - 1, lane, bq1);
        hook();
        g3_mfma<A, LO, 4, NC>(aS0, aR0, bq0, v0, u);
        g3_mfma_rem<A, LO, NC>(aS1, aR1, bq1, v0, rem, u);
    } else {
        hook();
        g3_mfma_rem<A, LO, NC>(aS0, aR0, bq0, v0, rem, u);
    }
}

// sbar = hbar .* act' + dbar .* G as ONE spelled-out sequence - the product dbar .* G rounded, then a fused multiply-add - so that every
// instantiation of both sweeps (cnf_coop_grad3.hip, cnf_coop_grad3w.hip) rounds alike: left to -ffp-contract=fast the compiler fused one
// product or the other depending on the code around it, and the two sweeps - which a call is routed to by its size - differed in the
// last bit (3e-7 of the gradient) the day one of them changed shape
__device__ __forceinline__ f32x4 g3_sbar(const f32x4& hbar, const f32x4& d, const f32x4& dbar, const f32x4& G) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = __fmul_rn(dbar[e], G[e]);
    return __builtin_elementwise_fma(hbar, d, t);
}

// G = delta act'' / act'  (so that  a2 .* act'' = dbar .* u .* act'' = dbar .* G  with delta = u .* act'):  tanh: act'' = -2 h act';
// softplus: act'' = act' (1 - act')
template <int ACT>
__device__ __forceinline__ f32x4 g3_G(const f32x4& h, const f32x4& dl, const f32x4& d) {
    if constexpr (ACT == CNF_ACT_SOFTPLUS) return dl * (1.f - d);
    else return h * dl * -2.f;
}

}  // namespace

}  // namespace cnf
