// cnf_coop_d.hip — the cooperative solve kernel with its tiles DEALT over four owner waves (round 4).
//
// Serves the one-probe Hutchinson-VJP configuration of the plans of cnf_coop_x.hip - TrainMode inference / loss and the
// checkpointing forward half of the cooperative gradient - i.e. what the reference's constructor-default architecture
// (src/core/icnf.jl:53-103: D = 2 nvariables + 1, two softplus layers of 4 (D + 1), lambdas 0.01) runs for nvariables >= 16.
// Same math and reference map as cnf_coop.hip / cnf_coop_x.hip (src/core/icnf.jl:517-559, src/core/utils.jl:150-159); same
// packed operand image (MfmaLayout(HT_inst, L, ZR_inst, 0, true), read through RUN-TIME offsets and pitches, so the plan, its
// device-side repacking and the reverse sweep that shares the image are untouched).
//
// What was wrong with the extended kernel on these shapes (profiles/r3/r3U_nv20_coopx_pmc.txt: 57 % MFMA-busy,
// 0.28-0.54 of peak): it executes 1.7 x the MFMAs the shape needs.  (i) Its instances pad the hidden width to 8 / 12 / 16 / 20 / 24
// tiles and the state to 8 / 16 / 24 k-steps, and only the K side of a product skips the padding: at nvariables = 16 (H = 136:
// 9 tiles, D = 33: 9 k-steps) the M side runs 12 tiles - one wave of four multiplies nothing but zeros - and the state products
// run 16 k-steps.  (ii) With 32-sample super-tiles only two of the four waves own a sample tile, and the two D-row products
// (zdot = W_N h_L, g = W_1[:,0:D]^T delta_1: a third of a stage's MFMAs at D ~ H/4) run on those two.
//
// Here a workgroup owns a 64-sample super-tile (one wave per SIMD, as cnf_coop.hip) and EVERY wave owns one sample tile:
//   * the HT = 4 A + b hidden M-tiles of a product are dealt exactly: wave w computes tiles [w A, (w+1) A) for all four sample
//     tiles (each weight fragment feeds 16 MFMAs) and the b < 4 left-over tiles for its OWN sample tile only.  A is a template
//     parameter, b a run-time count (wave-uniform branches around MFMA-only blocks), so nine hidden tiles cost nine tiles' worth
//     of MFMAs on every wave;
//   * k-loops run the real k-groups and the real k-steps of the last one (H = 136: 34 k-steps, not 36 or 48);
//   * the D-row products are split along K by OWNERSHIP (the organisation of the tile-split form, cnf_coop.hip): wave w
//     multiplies the k-groups of the features it has just produced - straight from its registers, before the barrier - for
//     all four sample tiles and publishes partial tiles; the owner of a sample tile adds the four partials in wave order.  The
//     activations of the last hidden layer and the cotangent of the first are never published, a weight fragment of W_N / W_1^T
//     feeds 16 MFMAs instead of 4, and L = 2 needs ONE exchange buffer.
#define CNF_NO_PK_ASM 1
#define CNF_NO_PHASE_FENCE 1
#include "cnf_coop_d_dev.h"

namespace cnf {

// instances whose Runge-Kutta sums live in the plan's global ring (KArgs::rk) instead of accumulation registers: the ones that
// spilled with them parked (12 and more state registers: 28 - 167 registers; three hidden layers at A = 3: 56)
constexpr bool cd_rk_in_ring(int A, int L, int ZR) { return ZR >= 12 || (A == 3 && L == 3); }

template <int A>
struct UAcc {
    f32x4 S[A][4];   // tiles [w A, (w+1) A) x the four sample tiles
    f32x4 R[3];      // left-over tiles 4 A + r (r < b) x this wave's own sample tile
};
template <int A>
__device__ __forceinline__ void uacc_fill(UAcc<A>& u, const f32x4 (&vS)[A], const f32x4 (&vR)[3]) {
#pragma unroll
    for (int m = 0; m < A; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) u.S[m][q] = vS[m];
#pragma unroll
    for (int r = 0; r < 3; ++r) u.R[r] = vR[r];
}
template <int A>
__device__ __forceinline__ void uacc_mfma_fence(UAcc<A>& u) {   // before accumulator tiles of a product go straight into park()
#pragma unroll
    for (int m = 0; m < A; ++m) mfma_results_fence(u.S[m][0], u.S[m][1], u.S[m][2], u.S[m][3]);
    f32x4 dummy = u.R[0];
    mfma_results_fence(u.R[0], u.R[1], u.R[2], dummy);
}
template <int A>
__device__ __forceinline__ void uacc_zero(UAcc<A>& u) {
#pragma unroll
    for (int m = 0; m < A; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) u.S[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) u.R[r] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// A fragments of k-group kg for this wave's units of an H-row product: image `img` (byte offset), k-group pitch KP
template <int A>
__device__ __forceinline__ void dealt_load_a(const DRs& R, const TileOff<A>& T, unsigned img, int kg, f32x4 (&aS)[A], f32x4 (&aR)[3]) {
    const unsigned so = img + (unsigned)kg * 1024u;
#pragma unroll
    for (int m = 0; m < A; ++m) aS[m] = dloadv(R, T.S[m], so);
#pragma unroll
    for (int r = 0; r < 3; ++r) aR[r] = dloadv(R, T.Rr[r], so);
}
__device__ __forceinline__ void dealt_load_b(const f32x4* __restrict__ bimg, int kg, int wave, int lane, f32x4 (&bq)[4], f32x4& bo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = bimg[(kg * 4 + q) * 64 + lane];
    // the own tile's address is rebuilt from the (opaque) scalar wave index on every call - one v_add - instead of living in a
    // loop-invariant vector register of its own: that register is what the allocator spilled to scratch in two earlier builds
    int wv = wave;
    asm volatile("" : "+s"(wv));
    bo = bimg[(kg * 4 + wv) * 64 + lane];
}
template <int A, int JN>
__device__ __forceinline__ void dealt_mfma(const f32x4 (&aS)[A], const f32x4 (&aR)[3], const f32x4 (&bq)[4], const f32x4& bo, int b,
                                           UAcc<A>& u) {
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) u.S[m][q] = mfma4(aS[m][j], bq[q][j], u.S[m][q]);
    if (b > 0) {
#pragma unroll
        for (int j = 0; j < JN; ++j) u.R[0] = mfma4(aR[0][j], bo[j], u.R[0]);
    }
    if (b > 1) {
#pragma unroll
        for (int j = 0; j < JN; ++j) u.R[1] = mfma4(aR[1][j], bo[j], u.R[1]);
    }
    if (b > 2) {
#pragma unroll
        for (int j = 0; j < JN; ++j) u.R[2] = mfma4(aR[2][j], bo[j], u.R[2]);
    }
}
// the last k-group of a product: `rem` (1 .. 4, wave-uniform) k-steps are real
template <int A>
__device__ __forceinline__ void dealt_mfma_rem(const f32x4 (&aS)[A], const f32x4 (&aR)[3], const f32x4 (&bq)[4], const f32x4& bo, int b,
                                               int rem, UAcc<A>& u) {
    if (rem == 4) dealt_mfma<A, 4>(aS, aR, bq, bo, b, u);
    else if (rem == 3) dealt_mfma<A, 3>(aS, aR, bq, bo, b, u);
    else if (rem == 2) dealt_mfma<A, 2>(aS, aR, bq, bo, b, u);
    else dealt_mfma<A, 1>(aS, aR, bq, bo, b, u);
}

// u += A(image) * B(LDS image) over KG k-groups, the last one with `rem` k-steps.  aS0 / aR0 arrive holding the fragments of
// k-group 0 (requested by the caller one phase earlier); two fragment sets ping-pong, the loads of k-group kg + 1 are issued
// before the MFMAs of k-group kg.
template <int A>
__device__ __forceinline__ void dealt_gemm(const DRs& R, const TileOff<A>& T, unsigned img, int KG, int rem, int b,
                                           const f32x4* __restrict__ bimg, int wave, int lane, f32x4 (&aS0)[A], f32x4 (&aR0)[3],
                                           UAcc<A>& u) {
    f32x4 aS1[A], aR1[3], bq0[4], bq1[4], bo0, bo1;
    dealt_load_b(bimg, 0, wave, lane, bq0, bo0);
    const int KGf = KG - 1;   // full k-groups
    int kg = 0;
#pragma clang loop unroll(disable)
    for (; kg + 2 <= KGf; kg += 2) {
        dealt_load_a<A>(R, T, img, kg + 1, aS1, aR1);
        dealt_load_b(bimg, kg + 1, wave, lane, bq1, bo1);
        dealt_mfma<A, 4>(aS0, aR0, bq0, bo0, b, u);
        dealt_load_a<A>(R, T, img, kg + 2, aS0, aR0);
        dealt_load_b(bimg, kg + 2, wave, lane, bq0, bo0);
        dealt_mfma<A, 4>(aS1, aR1, bq1, bo1, b, u);
    }
    if (kg < KGf) {   // one full k-group and the last one
        dealt_load_a<A>(R, T, img, KG - 1, aS1, aR1);
        dealt_load_b(bimg, KG - 1, wave, lane, bq1, bo1);
        dealt_mfma<A, 4>(aS0, aR0, bq0, bo0, b, u);
        dealt_mfma_rem<A>(aS1, aR1, bq1, bo1, b, rem, u);
    } else {
        dealt_mfma_rem<A>(aS0, aR0, bq0, bo0, b, rem, u);
    }
}

// K-split D-row product by ownership: the partial tiles sum over this wave's shared k-groups of W(dm, k) x[k, q] go to
// pbuf[wave][dm][q] (all four sample tiles, B operands straight from the registers that hold x), own[dm] = the same over its
// left-over k-groups for its own sample tile.  `img`: the D-row image (fN or b1).  The k-group that is the configuration's last
// one runs `rem` k-steps.  With QH = 2 the sample tiles are taken two at a time (the fragments are fetched twice): 32 instead of
// 64 accumulator registers at DT = 4, which is what keeps the 16-state-register instances out of scratch.
template <int A, int DT, int QH>
__device__ __forceinline__ void dealt_drow(const DRs& R, const unsigned (&vd)[DT], unsigned img, int kgS0, int kgR0, int KG, int rem, int b,
                                           const UAcc<A>& x, f32x4 (&f0)[DT], f32x4* __restrict__ pw, int lane, f32x4 (&own)[DT]) {
    // f0 arrives holding the fragments of k-group kgS0 (dealt_drow_first: requested before the activation phase)
    constexpr int QN = 4 / QH;
    f32x4 f1[DT];
#pragma unroll
    for (int dm = 0; dm < DT; ++dm) own[dm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qh = 0; qh < QH; ++qh) {
        f32x4 part[DT][QN];
#pragma unroll
        for (int dm = 0; dm < DT; ++dm)
#pragma unroll
            for (int q = 0; q < QN; ++q) part[dm][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qh > 0) {
#pragma unroll
            for (int dm = 0; dm < DT; ++dm) f0[dm] = dloadv(R, vd[dm], img + (unsigned)kgS0 * 1024u);
        }
        // shared block: k-groups kgS0 .. kgS0 + A - 1, then (first pass only) the left-over k-groups (clamped loads), pipelined by one
        constexpr int MEND_FIRST = A + 3;
#pragma unroll
        for (int m = 0; m < MEND_FIRST; ++m) {
            if (qh > 0 && m >= A) break;
            f32x4(&cur)[DT] = (m & 1) ? f1 : f0;
            f32x4(&nxt)[DT] = (m & 1) ? f0 : f1;
            const int mend = qh > 0 ? A : A + 3;
            if (m + 1 < mend) {
                const int raw = m + 1 < A ? kgS0 + m + 1 : kgR0 + (m + 1 - A);
                const int kgn = raw < KG ? raw : KG - 1;
#pragma unroll
                for (int dm = 0; dm < DT; ++dm) nxt[dm] = dloadv(R, vd[dm], img + (unsigned)kgn * 1024u);
            }
            if (m < A) {
                const bool last = kgS0 + m == KG - 1;   // only when b == 0 and this is the last wave's last tile
                if (!last || rem == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                            for (int q = 0; q < QN; ++q) part[dm][q] = mfma4(cur[dm][j], x.S[m][qh * QN + q][j], part[dm][q]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < rem) {
#pragma unroll
                            for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                                for (int q = 0; q < QN; ++q) part[dm][q] = mfma4(cur[dm][j], x.S[m][qh * QN + q][j], part[dm][q]);
                        }
                }
            } else {
                const int r = m - A;
                if (r < b) {
                    // (k-steps beyond the last real one multiply zero weights: run them rather than branch per k-step)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int dm = 0; dm < DT; ++dm) own[dm] = mfma4(cur[dm][j], x.R[r][j], own[dm]);
                }
            }
        }
#pragma unroll
        for (int dm = 0; dm < DT; ++dm)
#pragma unroll
            for (int q = 0; q < QN; ++q) pw[(dm * 4 + qh * QN + q) * 64 + lane] = part[dm][q];
    }
}

template <int DT>
__device__ __forceinline__ void dealt_drow_first(const DRs& R, const unsigned (&vd)[DT], unsigned img, int kgS0, f32x4 (&f0)[DT]) {
#pragma unroll
    for (int dm = 0; dm < DT; ++dm) f0[dm] = dloadv(R, vd[dm], img + (unsigned)kgS0 * 1024u);
}

// One dynamics evaluation for a 64-sample super-tile; every wave owns sample tile `wave`.
template <int A, int L, int ZR, int ACT>
__device__ __forceinline__ void coopd_eval(const DRs& R, const float* __restrict__ CV, const DImg& Gin, f32x4* __restrict__ xbuf, int XB,
                                           f32x4* __restrict__ zbuf, const f32x4* __restrict__ ebuf, f32x4* __restrict__ pbuf,
                                           const f32x4* __restrict__ ybuf, int lane, int wave, float t, bool autonomous, bool reg_z, bool reg_j,
                                           const float (&zs)[ZR], float (&zd)[ZR], float& ld, float& ed, float& nd,
                                           float* __restrict__ gout, const UAcc<A>& cP, f32x4 (&aS)[A], f32x4 (&aR)[3],
                                           float* __restrict__ fsb = nullptr, long long fsl = 0, int gqs = 4) {
    // fsb (checkpointing solves, round 6): the stage store of the second-order reverse sweep (cnf_tiles.h) - every h_l and delta_l
    // tile of this evaluation leaves for HBM from the registers of the wave that computed it, tile-native (one 16-byte store per
    // lane; HTs = 4 A + b tiles per sample tile).  fsb: float pointer at this super-tile's first tile of (kind h, layer 0) for this
    // stage; `fsl` floats separate consecutive (kind, layer) arrays.
    // cP: c = W_N^T eps of this wave's units, PARKED - eps is fixed for the whole solve (src/core/base_icnf.jl:258-259), so the
    //     product is taken once per super-tile (coopd_hoist_c), and delta_L = c .* act'_L falls out of the last activation pass;
    // aS / aR arrive holding the layer-1 fragments of k-group 0 and leave holding them again for the next evaluation (requested
    //     before the last barrier: the load is in flight across the Runge-Kutta update and the state publish)
    constexpr int DT = ZR / 4;
    static_assert(ZR % 4 == 0, "state registers in whole M-tiles");
    // every image offset of this evaluation hangs off an opaque zero: the several hundred wave-uniform fragment addresses are
    // then recomputed per evaluation (a few scalar instructions each) instead of being hoisted out of the stage and step loops,
    // where they do not fit the scalar register file (241 scalar spills in the first build, v_readlane reloads inside the k-loops)
    int opq = 0;
    asm volatile("" : "+s"(opq));
    DImg G = Gin;
    G.f1z += opq; G.fh += opq; G.fN += opq; G.bN += opq; G.bh += opq; G.b1 += opq; G.f1y += opq;
    const float* __restrict__ P = CV - G.v_b1;   // C vectors: the LDS copy, addressed by their image offsets
    const int g = lane >> 4;
    const int b = G.b;
    const int mtS0 = wave * A, mtR0 = 4 * A, mtRmax = 4 * A + b - 1;   // (b = 0: the clamped left-over loads re-read tile 4 A - 1)
    const unsigned F1Z = (unsigned)G.f1z * 4u, FH = (unsigned)G.fh * 4u, FN = (unsigned)G.fN * 4u, BN = (unsigned)G.bN * 4u,
                   BH = (unsigned)G.bh * 4u, B1 = (unsigned)G.b1 * 4u, IMGH = (unsigned)G.imgH * 4u;
    UAcc<A> acc;
    UAcc<A> d[L];      // act' of the hidden layers below the last (this wave's units), kept for the pullback: PARKED (see park)
    const TileOff<A> TZ = tile_offsets<A>(R, G.KPZ, mtS0, mtR0, mtRmax);   // state-column images (k-group pitch KPZ)
    const TileOff<A> TH = tile_offsets<A>(R, G.HTP, mtS0, mtR0, mtRmax);   // H-column images
    unsigned vd[DT];
#pragma unroll
    for (int dm = 0; dm < DT; ++dm) { vd[dm] = R.lane16 + (unsigned)(dm * G.HTP) * 1024u; asm volatile("" : "+v"(vd[dm])); }
    // C vector (bias, time column) of this wave's units
    auto cvec_units = [&](const float* __restrict__ vec, f32x4 (&vS)[A], f32x4 (&vR)[3]) {
#pragma unroll
        for (int m = 0; m < A; ++m) vS[m] = *reinterpret_cast<const f32x4*>(vec + ((mtS0 + m) * 4 + g) * 4);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int mt = mtR0 + r < mtRmax ? mtR0 + r : mtRmax;
            vR[r] = *reinterpret_cast<const f32x4*>(vec + (mt * 4 + g) * 4);
        }
    };
    auto fs_store = [&](int kl, const UAcc<A>& v) {
        if (!fsb) return;
        // (the array's address is made scalar by hand: a buffer resource in vector registers would put every store in a waterfall loop)
        const unsigned long long pa = (unsigned long long)(fsb + (long long)kl * fsl);
        const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa), phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)phi << 32) | plo), 0, 0x7fffffff, 0x00020000);
        const int HTs = 4 * A + b;
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v.S[m][q]), r, (int)R.lane16, (q * HTs + mtS0 + m) * 1024, 0 /* plain stores: non-temporal ones measured the same */);
                CNF_STORE_DATA_HAZARD(v.S[m][q]);
            }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
            if (rr < b) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v.R[rr]), r, (int)R.lane16, (wave * HTs + mtR0 + rr) * 1024, 0 /* plain stores: non-temporal ones measured the same */);
                CNF_STORE_DATA_HAZARD(v.R[rr]);
            }
    };
    // ---- layer 1: a = W1z z + w1t t + b1 ----
    {
        f32x4 bS[A], bR[3], wS[A], wR[3];
        cvec_units(P + G.v_b1, bS, bR);
        cvec_units(P + G.v_w1t, wS, wR);
#pragma unroll
        for (int kg = 0; kg < DT; ++kg) {   // this wave's stage state as the B image of sample tile `wave`
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = zs[4 * kg + j];
            zbuf[(kg * 4 + wave) * 64 + lane] = v;
        }
        if (!autonomous) {
#pragma unroll
            for (int m = 0; m < A; ++m) bS[m] = tile_fma(wS[m], t, bS[m]);
#pragma unroll
            for (int r = 0; r < 3; ++r) bR[r] = tile_fma(wR[r], t, bR[r]);
        }
        uacc_fill<A>(acc, bS, bR);
        __syncthreads();
        dealt_gemm<A>(R, TZ, F1Z, G.KGZ, G.remZ, b, zbuf, wave, lane, aS, aR, acc);
        if (G.remC > 0) {   // + W_1[:, condition columns] y  (CondLayer rows [z; t; ys], src/layers/cond_layer.jl:7-31): one k-group
            const TileOff<A> TC = tile_offsets<A>(R, G.KPC, mtS0, mtR0, mtRmax);
            dealt_load_a<A>(R, TC, (unsigned)G.f1y * 4u, 0, aS, aR);
            dealt_gemm<A>(R, TC, (unsigned)G.f1y * 4u, 1, G.remC, b, ybuf, wave, lane, aS, aR, acc);
        }
    }
    f32x4 own[DT], fd[DT];
    UAcc<A> h;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int cur = l & 1;   // exchange buffer of layer l + 1's activations (L = 2 only ever uses buffer 0)
        if (l + 1 < L) dealt_load_a<A>(R, TH, FH + (unsigned)l * IMGH, 0, aS, aR);
        else dealt_drow_first<DT>(R, vd, FN, mtS0, fd);
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (l + 1 < L) {
                    f32x4 dd;
                    act_pair<ACT>(acc.S[m][q], h.S[m][q], dd);
                    d[l].S[m][q] = park4(dd);
                } else {
                    h.S[m][q] = act_only<ACT>(acc.S[m][q]);
                }
            }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (l + 1 < L) {
                f32x4 dd;
                act_pair<ACT>(acc.R[r], h.R[r], dd);
                d[l].R[r] = park4(dd);
            } else {
                h.R[r] = act_only<ACT>(acc.R[r]);
            }
        }
        // (the stage-store writes are PLACED behind the product that follows the phase that made the tiles: the memory counter
        // retires in order, and a store burst in front of a product makes its second k-group wait for the acknowledgements.  The
        // tiles then stay live across that product: the A = 3 instances have no registers for it - 39 / 64 spilled - and store in front)
        constexpr bool FSP = A <= 2;
        if constexpr (!FSP) fs_store(l, h);
        if (l + 1 < L) {
#pragma unroll
            for (int m = 0; m < A; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) xbuf[cur * XB + ((mtS0 + m) * 4 + q) * 64 + lane] = h.S[m][q];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (r < b) xbuf[cur * XB + ((mtR0 + r) * 4 + wave) * 64 + lane] = h.R[r];
            f32x4 bS[A], bR[3];
            cvec_units(P + G.v_bh + l * G.vecH, bS, bR);
            uacc_fill<A>(acc, bS, bR);
            __syncthreads();
            dealt_gemm<A>(R, TH, FH + (unsigned)l * IMGH, G.KGH, G.remH, b, xbuf + cur * XB, wave, lane, aS, aR, acc);
            if constexpr (FSP) fs_store(l, h);   // h_{l+1} (1-based) of this stage
        }
    }
    // ---- zdot = W_N h_L + b_N: partials over this wave's own k-groups, from registers ----
    constexpr int QH = DT >= 4 ? 2 : 1;
    f32x4* __restrict__ pw = pbuf + (wave * DT) * 4 * 64;   // this wave's partial tiles: [dm][q][lane]
    if (G.xalias) __syncthreads();   // the partial tiles share the exchange buffer: its readers (the last hidden product) are done
    dealt_drow<A, DT, QH>(R, vd, FN, mtS0, mtR0, G.KGH, G.remH, b, h, fd, pw, lane, own);
    // the first fragments of the pullback's first product are requested before the barrier
    if (L > 1) dealt_load_a<A>(R, TH, BH + (unsigned)(L - 2) * IMGH, 0, aS, aR);
    if constexpr (A <= 2) fs_store(L - 1, h);    // h_L
    // delta_L = c .* act'_L, act'_L rebuilt from h_L (which the product above has consumed)
#pragma unroll
    for (int m = 0; m < A; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) h.S[m][q] = unpark4(cP.S[m][q]) * dact_from_h<ACT>(h.S[m][q]);
#pragma unroll
    for (int r = 0; r < 3; ++r) h.R[r] = unpark4(cP.R[r]) * dact_from_h<ACT>(h.R[r]);
    __syncthreads();
    {
        f32x4 zacc[DT];
#pragma unroll
        for (int dm = 0; dm < DT; ++dm) {
            zacc[dm] = *reinterpret_cast<const f32x4*>(P + G.v_bN + (dm * 4 + g) * 4);   // (LDS)
#pragma unroll
            for (int w = 0; w < 4; ++w) zacc[dm] += pbuf[((w * DT + dm) * 4 + wave) * 64 + lane];
            zacc[dm] += own[dm];
        }
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = zacc[s >> 2][s & 3];
    }
    ed = 0.f;
    if (reg_z) {
        float e2 = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) e2 = fmaf(zd[s], zd[s], e2);
        ed = sqrtf(group_sum(e2));   // Edot = |zdot|_2   (src/core/icnf.jl:184-199)
    }
    // ---- pullback: delta_L = c .* act'_L (above), delta_l = (W_{l+1}^T delta_{l+1}) .* act'_l ----
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
        // delta_{l+1} (1-based)
        if (l == 0) dealt_drow_first<DT>(R, vd, B1, mtS0, fd);
        if (l == L - 1) {
            // (h already holds delta_L)
        } else {
#pragma unroll
            for (int m = 0; m < A; ++m) {
                f32x4 dd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) dd[q] = unpark4(d[l].S[m][q]);
                tiles_mul<4>(acc.S[m], dd, h.S[m]);
            }
            f32x4 dd[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) dd[r] = unpark4(d[l].R[r]);
            tiles_mul<3>(acc.R, dd, h.R);
        }
        if constexpr (A > 2) fs_store(L + l, h); // delta_{l+1} (1-based): in front (see above)
        if (l > 0) {
            // exchange buffer: h_l sat in buffer (l - 1) & 1; every reader passed a barrier since.  L = 2: buffer 0 again.
            const int wbuf = (L == 2) ? 0 : ((l - 1) & 1) ^ 1;
            if (l < L - 1) dealt_load_a<A>(R, TH, BH + (unsigned)(l - 1) * IMGH, 0, aS, aR);
            if (G.xalias && l == L - 1) __syncthreads();   // the owners have read the zdot partials out of this buffer
#pragma unroll
            for (int m = 0; m < A; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) xbuf[wbuf * XB + ((mtS0 + m) * 4 + q) * 64 + lane] = h.S[m][q];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (r < b) xbuf[wbuf * XB + ((mtR0 + r) * 4 + wave) * 64 + lane] = h.R[r];
            uacc_zero<A>(acc);
            __syncthreads();
            dealt_gemm<A>(R, TH, BH + (unsigned)(l - 1) * IMGH, G.KGH, G.remH, b, xbuf + wbuf * XB, wave, lane, aS, aR, acc);
            if constexpr (A <= 2) fs_store(L + l, h);   // delta_{l+1} (1-based), behind the product that read it
        }
    }
    // ---- g = W_1[:,0:D]^T delta_1 = eps^T J: partials from registers ----
    if (G.xalias) __syncthreads();
    dealt_drow<A, DT, QH>(R, vd, B1, mtS0, mtR0, G.KGH, G.remH, b, h, fd, pw, lane, own);
    dealt_load_a<A>(R, TZ, F1Z, 0, aS, aR);   // the next evaluation's layer-1 fragments
    if constexpr (A <= 2) fs_store(L, h);        // delta_1
    __syncthreads();
    {
        float dot = 0.f, n2 = 0.f;
#pragma unroll
        for (int dm = 0; dm < DT; ++dm) {
            f32x4 ga = pbuf[((0 * DT + dm) * 4 + wave) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) ga += pbuf[((w * DT + dm) * 4 + wave) * 64 + lane];
            ga += own[dm];
            const f32x4 ev = ebuf[(dm * 4 + wave) * 64 + lane];   // this lane's probe values sit in the B image of eps: no registers held
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dot = fmaf(ga[j], ev[j], dot);   // <eps^T J, eps>
                n2 = fmaf(ga[j], ga[j], n2);
                if (gout) gout[gqs * dm + j] = ga[j];     // checkpointing solve: g of this stage for the reverse sweep (gqs: floats between 16-row groups)
            }
        }
        ld = -group_sum(dot);
        nd = reg_j ? sqrtf(group_sum(n2)) : 0.f;   // ndot = |eps^T J|_2 (src/core/icnf.jl:229-245)
    }
    // (the next evaluation's first LDS write is the state image, whose last readers - layer 1 - passed several barriers ago;
    //  pbuf is rewritten only behind the next evaluation's own barriers)
}

// One dynamics evaluation in TestMode (exact trace) for a two-hidden-layer flow - the reference's default architecture
// (src/core/icnf.jl:297-339, src/core/utils.jl:79-88): ldot = -tr J with tr J = act'_2^T Q act'_1 and the constant
// Q = W_2 .* (W_1[:,0:D] W_3)^T packed behind the operand images (cnf_mfma.hip: mfma_pack), i.e. the forward chain, ONE more H x H
// product whose B operand is act'_1, and a dot - no pullback, no probe.  Same dealing as coopd_eval: every wave computes its
// units of Q act'_1, multiplies them by its own act'_2 and sums over its features; the four waves' partial traces of a sample
// tile meet in LDS.  The exact-trace dynamics carry no regularisers (Edot = ndot = 0).
template <int A, int ZR, int ACT>
__device__ __forceinline__ void coopd_eval_exact(const DRs& R, const float* __restrict__ CV, const DImg& Gin, f32x4* __restrict__ xbuf,
                                                 f32x4* __restrict__ zbuf, f32x4* __restrict__ pbuf, float* __restrict__ red,
                                                 const f32x4* __restrict__ ybuf, int lane, int wave, float t, bool autonomous, const float (&zs)[ZR], float (&zd)[ZR],
                                                 float& ld, f32x4 (&aS)[A], f32x4 (&aR)[3]) {
    constexpr int DT = ZR / 4;
    int opq = 0;
    asm volatile("" : "+s"(opq));
    DImg G = Gin;
    G.f1z += opq; G.fh += opq; G.fN += opq; G.q_off += opq; G.f1y += opq;
    const float* __restrict__ P = CV - G.v_b1;
    const int g = lane >> 4;
    const int b = G.b;
    const int mtS0 = wave * A, mtR0 = 4 * A, mtRmax = 4 * A + b - 1;
    const unsigned F1Z = (unsigned)G.f1z * 4u, FH = (unsigned)G.fh * 4u, FN = (unsigned)G.fN * 4u, QI = (unsigned)G.q_off * 4u;
    UAcc<A> acc, h, d1p, a2p;   // d1p: act'_1, a2p: the second layer's pre-activations of this wave's units - parked
    const TileOff<A> TZ = tile_offsets<A>(R, G.KPZ, mtS0, mtR0, mtRmax);
    const TileOff<A> TH = tile_offsets<A>(R, G.HTP, mtS0, mtR0, mtRmax);
    unsigned vd[DT];
#pragma unroll
    for (int dm = 0; dm < DT; ++dm) { vd[dm] = R.lane16 + (unsigned)(dm * G.HTP) * 1024u; asm volatile("" : "+v"(vd[dm])); }
    auto cvec_units = [&](const float* __restrict__ vec, f32x4 (&vS)[A], f32x4 (&vR)[3]) {
#pragma unroll
        for (int m = 0; m < A; ++m) vS[m] = *reinterpret_cast<const f32x4*>(vec + ((mtS0 + m) * 4 + g) * 4);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int mt = mtR0 + r < mtRmax ? mtR0 + r : mtRmax;
            vR[r] = *reinterpret_cast<const f32x4*>(vec + (mt * 4 + g) * 4);
        }
    };
    auto publish = [&](const UAcc<A>& v) {
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) xbuf[((mtS0 + m) * 4 + q) * 64 + lane] = v.S[m][q];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (r < b) xbuf[((mtR0 + r) * 4 + wave) * 64 + lane] = v.R[r];
    };
    // ---- layer 1 ----
    {
        f32x4 bS[A], bR[3], wS[A], wR[3];
        cvec_units(P + G.v_b1, bS, bR);
        cvec_units(P + G.v_w1t, wS, wR);
#pragma unroll
        for (int kg = 0; kg < DT; ++kg) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = zs[4 * kg + j];
            zbuf[(kg * 4 + wave) * 64 + lane] = v;
        }
        if (!autonomous) {
#pragma unroll
            for (int m = 0; m < A; ++m) bS[m] = tile_fma(wS[m], t, bS[m]);
#pragma unroll
            for (int r = 0; r < 3; ++r) bR[r] = tile_fma(wR[r], t, bR[r]);
        }
        uacc_fill<A>(acc, bS, bR);
        __syncthreads();
        dealt_gemm<A>(R, TZ, F1Z, G.KGZ, G.remZ, b, zbuf, wave, lane, aS, aR, acc);
        if (G.remC > 0) {   // + W_1[:, condition columns] y  (CondLayer rows [z; t; ys], src/layers/cond_layer.jl:7-31): one k-group
            const TileOff<A> TC = tile_offsets<A>(R, G.KPC, mtS0, mtR0, mtRmax);
            dealt_load_a<A>(R, TC, (unsigned)G.f1y * 4u, 0, aS, aR);
            dealt_gemm<A>(R, TC, (unsigned)G.f1y * 4u, 1, G.remC, b, ybuf, wave, lane, aS, aR, acc);
        }
    }
    // ---- hidden layer 1: publish h_1; act'_1 waits (parked) for the exchange buffer ----
    dealt_load_a<A>(R, TH, FH, 0, aS, aR);
#pragma unroll
    for (int m = 0; m < A; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 dd;
            act_pair<ACT>(acc.S[m][q], h.S[m][q], dd);
            d1p.S[m][q] = park4(dd);
        }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        f32x4 dd;
        act_pair<ACT>(acc.R[r], h.R[r], dd);
        d1p.R[r] = park4(dd);
    }
    publish(h);
    {
        f32x4 bS[A], bR[3];
        cvec_units(P + G.v_bh, bS, bR);
        uacc_fill<A>(acc, bS, bR);
    }
    __syncthreads();
    dealt_gemm<A>(R, TH, FH, G.KGH, G.remH, b, xbuf, wave, lane, aS, aR, acc);   // a_2 = W_2 h_1 + b_2
    // ---- Q act'_1 FIRST (act'_1 as the B image), the pre-activations a_2 parked meanwhile: the trace then meets act'_2 the
    //      moment it is computed, and neither act' has to outlive a product ----
    dealt_load_a<A>(R, TH, QI, 0, aS, aR);
    uacc_mfma_fence<A>(acc);
#pragma unroll
    for (int m = 0; m < A; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) { a2p.S[m][q] = park4(acc.S[m][q]); h.S[m][q] = unpark4(d1p.S[m][q]); }
#pragma unroll
    for (int r = 0; r < 3; ++r) { a2p.R[r] = park4(acc.R[r]); h.R[r] = unpark4(d1p.R[r]); }
    __syncthreads();   // every wave is done reading h_1
    publish(h);
    uacc_zero<A>(acc);
    __syncthreads();
    dealt_gemm<A>(R, TH, QI, G.KGH, G.remH, b, xbuf, wave, lane, aS, aR, acc);   // Q act'_1
    // ---- hidden layer 2: h_2 feeds zdot from registers; act'_2 meets Q act'_1 ----
    f32x4 own[DT], fd[DT];
    dealt_drow_first<DT>(R, vd, FN, mtS0, fd);
    float tr[4], trown = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) tr[q] = 0.f;
#pragma unroll
    for (int m = 0; m < A; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 dd;
            act_pair<ACT>(unpark4(a2p.S[m][q]), h.S[m][q], dd);
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[q] = fmaf(acc.S[m][q][r], dd[r], tr[q]);
        }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        f32x4 dd;
        act_pair<ACT>(unpark4(a2p.R[r]), h.R[r], dd);
        if (r < b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) trown = fmaf(acc.R[r][j], dd[j], trown);
        }
    }
    if (G.xalias) __syncthreads();            // the partial tiles share the exchange buffer: the Q product's readers are done
    dealt_drow<A, DT, (DT >= 4 ? 2 : 1)>(R, vd, FN, mtS0, mtR0, G.KGH, G.remH, b, h, fd, pbuf + (wave * DT) * 4 * 64, lane, own);
    dealt_load_a<A>(R, TZ, F1Z, 0, aS, aR);   // the next evaluation's layer-1 fragments
#pragma unroll
    for (int q = 0; q < 4; ++q) red[(wave * 4 + q) * 64 + lane] = group_sum(tr[q]);
    trown = group_sum(trown);
    __syncthreads();
    {
        f32x4 zacc[DT];
#pragma unroll
        for (int dm = 0; dm < DT; ++dm) {
            zacc[dm] = *reinterpret_cast<const f32x4*>(P + G.v_bN + (dm * 4 + g) * 4);
#pragma unroll
            for (int w = 0; w < 4; ++w) zacc[dm] += pbuf[((w * DT + dm) * 4 + wave) * 64 + lane];
            zacc[dm] += own[dm];
        }
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = zacc[s >> 2][s & 3];
        ld = -(((red[(0 * 4 + wave) * 64 + lane] + red[(1 * 4 + wave) * 64 + lane]) + red[(2 * 4 + wave) * 64 + lane]) +
               red[(3 * 4 + wave) * 64 + lane] + trown);
    }
    // (xalias: the next evaluation's first write into the exchange buffer - h_1 - comes behind two barriers)
}

// once per super-tile: c = W_N^T eps of this wave's units (B operand: the probe image ebuf, published and fenced by the caller),
// parked; and the first evaluation's layer-1 fragments
template <int A, int ZR>
__device__ __forceinline__ void coopd_hoist_c(const DRs& R, const DImg& G, const f32x4* __restrict__ ebuf, int lane, int wave,
                                              UAcc<A>& cP, f32x4 (&aS)[A], f32x4 (&aR)[3]) {
    const int b = G.b, mtS0 = wave * A, mtR0 = 4 * A, mtRmax = 4 * A + b - 1;
    const TileOff<A> TZ = tile_offsets<A>(R, G.KPZ, mtS0, mtR0, mtRmax);
    UAcc<A> acc;
    uacc_zero<A>(acc);
    dealt_load_a<A>(R, TZ, (unsigned)G.bN * 4u, 0, aS, aR);
    dealt_gemm<A>(R, TZ, (unsigned)G.bN * 4u, G.KGZ, G.remZ, b, ebuf, wave, lane, aS, aR, acc);
    uacc_mfma_fence<A>(acc);
#pragma unroll
    for (int m = 0; m < A; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) cP.S[m][q] = park4(acc.S[m][q]);
#pragma unroll
    for (int r = 0; r < 3; ++r) cP.R[r] = park4(acc.R[r]);
    dealt_load_a<A>(R, TZ, (unsigned)G.f1z * 4u, 0, aS, aR);
}

constexpr int coopd_lds_bytes(int HT, int L, int DT, bool alias, int cvn, bool cond = false) {
    return ((L == 2 ? 1 : 2) * HT * 4 * 64 + 2 * DT * 4 * 64 + (alias ? 0 : 4 * DT * 4 * 64) + (cond ? 4 * 64 : 0)) * 16 + (cvn + 3) / 4 * 16;
}

// The Runge-Kutta running sums of the later stages' increments (P_i <- P_{i+1} + a_{s+1+i,s} zdot, as the other kernels keep them)
// and the step sum - six rows of ZR registers, touched once per stage - are parked in accumulation registers (see park): the
// same fma chains in the same order, so the numbers are the same; 72 - 96 of the 256 architectural registers stay free for the
// 4 A + b accumulator tiles and two fragment sets of the k-loops.
// MODE 0: one Hutchinson probe, VJP (TrainMode); 1: exact trace through the Q product (TestMode, two hidden layers)
// NL: rows of the Runge-Kutta state (P_1 .. P_4, the step sum, z) kept in a per-wave LDS slice, the other 6 - NL parked in
// accumulation registers (0: all parked, or all in the global ring where cd_rk_in_ring says so).  The (2, 12) instance - the
// reference's default architecture at nvariables = 16 .. 21 - is 28 registers short with six rows parked and has 36 KB of LDS to
// spare: three rows there, three parked, and the 6.9 GB per solve the global ring moved through L2 and HBM are gone.
template <int A, int L, int ZR, int ACT, int MODE, int NL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
coopd_solve_kernel(DArgs da) {
    static_assert(MODE == 0 || L == 2, "the exact-trace form is the two-hidden-layer Q product");
    const KArgs& a = da.k;
    const DImg& G = da.g;
    constexpr int DT = ZR / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HT = 4 * A + G.b;
    const int XB = HT * 4 * 64;
    f32x4* xbuf = reinterpret_cast<f32x4*>(smem);            // [1 or 2][HT][4 sample tiles][64 lanes]
    f32x4* zbuf = xbuf + (L == 2 ? 1 : 2) * XB;               // [DT][4][64]
    f32x4* ebuf = zbuf + DT * 4 * 64;                         // [DT][4][64]
    f32x4* pbuf = G.xalias ? xbuf : ebuf + DT * 4 * 64;       // [4 waves][DT][4][64] partial tiles of the D-row products
    f32x4* ybuf = ebuf + DT * 4 * 64 + (G.xalias ? 0 : 4 * DT * 4 * 64);   // [1][4][64]: conditions (constant over the solve)
    float* cbuf = reinterpret_cast<float*>(ybuf + (G.remC > 0 ? 4 * 64 : 0));   // C vectors (biases, time column)
    f32x4* lrk0 = reinterpret_cast<f32x4*>(cbuf + (G.cvn + 3) / 4 * 4);         // (NL > 0) [4 waves][NL][DT][64]: rows of the Runge-Kutta state
    for (int i = threadIdx.x; i < G.cvn; i += 256) cbuf[i] = a.packed[G.v_b1 + i];
    // (the first __syncthreads of the super-tile loop orders these writes before any read)
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = a.D, S = D + 3;
    const bool reg_z = a.reg_z, reg_j = a.reg_j, autonomous = a.autonomous;
    const long long nst = (a.B + 63) / 64;
    DRs R{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, 0x7fffffff, 0x00020000), (unsigned)lane * 16u};

    for (long long st = blockIdx.x; st < nst; st += gridDim.x) {
        const long long smp = st * 64 + wave * 16 + n;
        const bool valid = smp < a.B;
        const long long sc = valid ? smp : a.B - 1;
        // RKG: z and the five running sums in a per-wave slice of the plan's global ring (L2-resident; read and written once per
        // stage as 16-byte accesses, behind the evaluation's last product and the next evaluation's first fragment requests) -
        // see cnf_coop_d2.hip, where the same move took the 20 .. 24-tile instances from 77 to 110 TFLOP/s
        constexpr bool RKL = NL > 0;
        constexpr bool RKG = !RKL && cd_rk_in_ring(A, L, ZR);
        f32x4* __restrict__ rk = RKG ? reinterpret_cast<f32x4*>(a.rk) + ((long long)(blockIdx.x * 4 + wave) * 6 * DT) * 64 + lane : nullptr;
        f32x4* lrk = lrk0 + (wave * (RKL ? NL : 1) * DT) * 64 + lane;
        float zs[ZR], zp[RKG || RKL ? 1 : ZR], pk[RKG || RKL ? 1 : 5][RKG || RKL ? 1 : ZR];   // stage state; z and the running sums, parked
        f32x4 pkv[RKL ? 6 - NL : 1][RKL ? DT : 1];                                            // (RKL) rows NL .. 5, parked
        float lacc = 0.f, eacc = 0.f, nacc = 0.f;
        __syncthreads();   // the previous super-tile's readers of the LDS images are done
#pragma unroll
        for (int kg = 0; kg < DT; ++kg) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = 4 * kg + j, f = 4 * s + g;
                if (a.x) zs[s] = f < a.nvars ? a.x[sc * a.nvars + f] : 0.f;
                else zs[s] = f < D ? a.u0[sc * S + f] : 0.f;
                if constexpr (!RKG && !RKL) zp[s] = park(zs[s]);
                v[j] = (MODE == 0 && f < D) ? a.eps[sc * D + f] : 0.f;
            }
            if constexpr (MODE == 0) ebuf[(kg * 4 + wave) * 64 + lane] = v;   // the probe of this wave's sample tile as a B image, for the whole solve
        }
        if (G.remC > 0) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int f = 4 * j + g; v[j] = f < a.C ? a.ys[sc * a.C + f] : 0.f; }
            ybuf[wave * 64 + lane] = v;
        }
        if constexpr (RKG) {
#pragma unroll
            for (int q = 0; q < DT; ++q) rk[(5 * DT + q) * 64] = f32x4{zs[4 * q], zs[4 * q + 1], zs[4 * q + 2], zs[4 * q + 3]};
        }
        if constexpr (RKL) {
            static_assert(NL <= 5, "z stays parked");
#pragma unroll
            for (int q = 0; q < DT; ++q) pkv[5 - NL][q] = park4(f32x4{zs[4 * q], zs[4 * q + 1], zs[4 * q + 2], zs[4 * q + 3]});
        }
        if (!a.x) { lacc = a.u0[sc * S + D]; eacc = a.u0[sc * S + D + 1]; nacc = a.u0[sc * S + D + 2]; }
        UAcc<A> cP;
        f32x4 aS[A], aR[3];
        if constexpr (MODE == 0) {
            __syncthreads();
            coopd_hoist_c<A, ZR>(R, G, ebuf, lane, wave, cP, aS, aR);
        } else {
            const int b_ = G.b;
            const TileOff<A> TZ = tile_offsets<A>(R, G.KPZ, wave * A, 4 * A, 4 * A + b_ - 1);
            dealt_load_a<A>(R, TZ, (unsigned)G.f1z * 4u, 0, aS, aR);
        }

        float zd[ZR], ld = 0.f, ed = 0.f, nd = 0.f;
        const float dt0 = a.dt;
        const bool single = a.nsteps == 0;
        const int ns = single ? 1 : (a.T.ns < 6 ? a.T.ns : 6);
        const int nsteps = single ? 1 : a.nsteps;
        // checkpoints for the cooperative gradient: [..][16-sample tile][lane][ckzr], or (KArgs::ck_tiles) [..][tile][16-row group][lane][4]:
        // state register s of this lane sits at lane * ckls + (s >> 2) * ckqs + (s & 3) inside the tile's 64 ckzr floats
        const long long cktile = st * 4 + wave, ckntp = nst * 4;
        const int ckzr = G.ckzr;
        const int ckls = a.ck_tiles ? 4 : ckzr, ckqs = a.ck_tiles ? 256 : 4;
        auto ck_put = [&](float* c, const float (&v)[ZR]) {
#pragma unroll
            for (int q = 0; q < ZR / 4; ++q) *reinterpret_cast<f32x4*>(c + q * ckqs) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            for (int q = ZR / 4; q < ckzr / 4; ++q) *reinterpret_cast<f32x4*>(c + q * ckqs) = f32x4{0.f, 0.f, 0.f, 0.f};
        };
#pragma clang loop unroll(disable)
        for (int step = 0; step < nsteps; ++step) {
            const float tn = a.tgrid ? a.tgrid[step] : a.t0 + (float)step * dt0;
            const float dt = a.tgrid ? a.tgrid[step + 1] - tn : dt0;
            if (a.ckpt && !single) {
                ck_put(a.ckpt + ((long long)step * ckntp + cktile) * 64 * ckzr + lane * ckls, zs);   // (at a step's start the stage state IS z)
            }
            float lsum = 0.f, esum = 0.f, nsum = 0.f;
#pragma clang loop unroll(disable)
            for (int sg = 0; sg < ns; ++sg) {
                const long long ckrow = (((long long)step * ns + sg) * ckntp + cktile) * 64 * ckzr + lane * ckls;
                float* gout = (a.ckpt_g && !single) ? a.ckpt_g + ckrow : nullptr;
                if constexpr (MODE == 0) {
                    float* fsb = nullptr;
                    long long fsl = 0;
                    if (a.kfull && !single) {   // (the checkpointing solve's use of KArgs::kfull: base of the stage store, cnf_tiles.h)
                        fsl = (long long)nsteps * ns * ckntp * HT * 256;
                        fsb = a.kfull + ((((long long)step * ns + sg) * ckntp + st * 4) * HT) * 256;
                    }
                    coopd_eval<A, L, ZR, ACT>(R, cbuf, G, xbuf, XB, zbuf, ebuf, pbuf, ybuf, lane, wave, tn + a.T.c[sg] * dt, autonomous, reg_z, reg_j,
                                              zs, zd, ld, ed, nd, gout, cP, aS, aR, fsb, fsl, ckqs);
                }
                else
                    coopd_eval_exact<A, ZR, ACT>(R, cbuf, G, xbuf, zbuf, pbuf, reinterpret_cast<float*>(ebuf), ybuf, lane, wave, tn + a.T.c[sg] * dt,
                                                 autonomous, zs, zd, ld, aS, aR);   // (the probe image's LDS holds the partial traces)
                if (gout)
                    for (int q = ZR / 4; q < ckzr / 4; ++q) *reinterpret_cast<f32x4*>(gout + q * ckqs) = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.ckpt_k && !single) ck_put(a.ckpt_k + ckrow, zd);
                const float bst = a.T.b[sg];
                lsum = fmaf(bst, ld, lsum); esum = fmaf(bst, ed, esum); nsum = fmaf(bst, nd, nsum);
                if (single) break;
                // running sums (parked rows pk[0..3] = P_1 .. P_4, pk[4] = the step sum): P_i <- fma(acol[sg][i], zdot, P_{i+1}),
                // P_4 <- acol[sg][4] zdot, step sum <- fma(b, zdot, step sum); the next stage state is z + dt P_0
                const float c0 = a.acol[sg][0], c1 = a.acol[sg][1], c2 = a.acol[sg][2], c3 = a.acol[sg][3], c4 = a.acol[sg][4];
                const bool first = sg == 0, lastst = sg == ns - 1;
                if constexpr (RKG) {
                    f32x4 o[5][DT], zz[DT];
#pragma unroll
                    for (int q = 0; q < DT; ++q) {
#pragma unroll
                        for (int r = 0; r < 5; ++r) o[r][q] = first ? f32x4{0.f, 0.f, 0.f, 0.f} : rk[(r * DT + q) * 64];
                        zz[q] = rk[(5 * DT + q) * 64];
                    }
#pragma unroll
                    for (int q = 0; q < DT; ++q) {
                        const f32x4 k = {zd[4 * q], zd[4 * q + 1], zd[4 * q + 2], zd[4 * q + 3]};
                        const f32x4 p0 = k * c0 + o[0][q], nsu = k * bst + o[4][q];
                        rk[(0 * DT + q) * 64] = k * c1 + o[1][q];
                        rk[(1 * DT + q) * 64] = k * c2 + o[2][q];
                        rk[(2 * DT + q) * 64] = k * c3 + o[3][q];
                        rk[(3 * DT + q) * 64] = k * c4;
                        rk[(4 * DT + q) * 64] = nsu;
                        f32x4 zn4 = p0 * dt + zz[q];
                        if (lastst) { zn4 = nsu * dt + zz[q]; rk[(5 * DT + q) * 64] = zn4; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) zs[4 * q + j] = zn4[j];
                    }
                } else if constexpr (RKL) {
                    // the same expressions as the ring's path, row r from LDS (r < NL) or from its parked registers
                    auto get = [&](int r, int q) -> f32x4 { return r < NL ? lrk[(r * DT + q) * 64] : unpark4(pkv[r < NL ? 0 : r - NL][q]); };
                    auto put = [&](int r, int q, const f32x4& v) {
                        if (r < NL) lrk[(r * DT + q) * 64] = v;
                        else pkv[r < NL ? 0 : r - NL][q] = park4(v);
                    };
                    f32x4 o[5][DT], zz[DT];
#pragma unroll
                    for (int q = 0; q < DT; ++q) {
#pragma unroll
                        for (int r = 0; r < 5; ++r) o[r][q] = first ? f32x4{0.f, 0.f, 0.f, 0.f} : get(r, q);
                        zz[q] = get(5, q);
                    }
#pragma unroll
                    for (int q = 0; q < DT; ++q) {
                        const f32x4 k = {zd[4 * q], zd[4 * q + 1], zd[4 * q + 2], zd[4 * q + 3]};
                        const f32x4 p0 = k * c0 + o[0][q], nsu = k * bst + o[4][q];
                        put(0, q, k * c1 + o[1][q]);
                        put(1, q, k * c2 + o[2][q]);
                        put(2, q, k * c3 + o[3][q]);
                        put(3, q, k * c4);
                        put(4, q, nsu);
                        f32x4 zn4 = p0 * dt + zz[q];
                        if (lastst) { zn4 = nsu * dt + zz[q]; put(5, q, zn4); }
#pragma unroll
                        for (int j = 0; j < 4; ++j) zs[4 * q + j] = zn4[j];
                    }
                } else {
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    const float k = zd[s];
                    const float o1 = first ? 0.f : unpark(pk[0][s]), o2 = first ? 0.f : unpark(pk[1][s]),
                                o3 = first ? 0.f : unpark(pk[2][s]), o4 = first ? 0.f : unpark(pk[3][s]),
                                os = first ? 0.f : unpark(pk[4][s]);
                    const float zz = unpark(zp[s]);
                    const float p0 = fmaf(c0, k, o1);
                    pk[0][s] = park(fmaf(c1, k, o2));
                    pk[1][s] = park(fmaf(c2, k, o3));
                    pk[2][s] = park(fmaf(c3, k, o4));
                    pk[3][s] = park(c4 * k);
                    const float nsu = fmaf(bst, k, os);
                    pk[4][s] = park(nsu);
                    zs[s] = fmaf(dt, p0, zz);
                    if (lastst) { const float zn = fmaf(dt, nsu, zz); zp[s] = park(zn); zs[s] = zn; }
                }
                }
            }
            if (single) break;
            lacc = fmaf(dt, lsum, lacc); eacc = fmaf(dt, esum, eacc); nacc = fmaf(dt, nsum, nacc);
        }
        if (a.ckpt && !single) {
            ck_put(a.ckpt + ((long long)nsteps * ckntp + cktile) * 64 * ckzr + lane * ckls, zs);
        }
        if (single) {
            if (valid) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) a.u_out[smp * S + f] = zd[s]; }
                if (g == 0) { a.u_out[smp * S + D] = ld; a.u_out[smp * S + D + 1] = ed; a.u_out[smp * S + D + 2] = nd; }
            }
            continue;
        }
        // ---- epilogue: inference_sol (src/core/base_icnf.jl:158-172) ----
        float ss = 0.f, sa = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            const float v2 = zs[s] * zs[s];
            ss += v2;
            if (f >= a.nvars) sa += v2;
        }
        ss = group_sum(ss);
        sa = group_sum(sa);
        if (valid) {
            if (a.u_out) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) a.u_out[smp * S + f] = zs[s]; }
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
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
template <int A, int L, int ZR, int ACT, int MODE, int NL = 0>
static hipError_t launch_coopd(const DArgs& a, int lds, int nblocks, hipStream_t st) {
    auto kern = coopd_solve_kernel<A, L, ZR, ACT, MODE, NL>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (!once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

struct CoopDInst {
    int A, L, ZR, ACT, MODE;
    hipError_t (*fn)(const DArgs&, int, int, hipStream_t);
    int NL;                                                      // rows of the Runge-Kutta state in LDS of the form below (0: none)
    hipError_t (*fn_lds)(const DArgs&, int, int, hipStream_t);   // the same instance with NL rows of the Runge-Kutta state in LDS, or null
};
constexpr int kCdLdsRows = 3;
#define CD_INST(A, L, ZR, ACT) CoopDInst { A, L, ZR, ACT, 0, &launch_coopd<A, L, ZR, ACT, 0>, 0, nullptr }
#define CD_INST_LDS(A, L, ZR, ACT) CoopDInst { A, L, ZR, ACT, 0, &launch_coopd<A, L, ZR, ACT, 0>, kCdLdsRows, &launch_coopd<A, L, ZR, ACT, 0, kCdLdsRows> }
#define CD_EXACT(A, ZR, ACT) CoopDInst { A, 2, ZR, ACT, 1, &launch_coopd<A, 2, ZR, ACT, 1>, 0, nullptr }
// (A, ZR) pairs of the reference's default architecture H = 4 (D + 1): D <= 48 with 9 .. 12 hidden tiles (nvariables 16 .. 23),
// D <= 64 with 13 .. 16 (nvariables 24 .. 31); the same pairs serve any flow of those sizes
#define CD_SHAPES(L, ACT) CD_INST_LDS(2, L, 12, ACT), CD_INST(3, L, 12, ACT), CD_INST(3, L, 16, ACT)
static const CoopDInst kCoopD[] = {
    CD_SHAPES(2, CNF_ACT_SOFTPLUS),
    // other flows of 8 .. 15 hidden tiles: tanh, three hidden layers (A = 2: act' of two layers parked), D <= 32
    CD_INST(2, 2, 8, CNF_ACT_SOFTPLUS), CD_INST(3, 2, 8, CNF_ACT_SOFTPLUS), CD_INST(2, 3, 8, CNF_ACT_SOFTPLUS), CD_INST(2, 3, 12, CNF_ACT_SOFTPLUS),
    CD_INST(2, 2, 8, CNF_ACT_TANH_PRESCALED), CD_INST(3, 2, 8, CNF_ACT_TANH_PRESCALED), CD_SHAPES(2, CNF_ACT_TANH_PRESCALED),
    CD_INST(2, 3, 8, CNF_ACT_TANH_PRESCALED), CD_INST(3, 3, 8, CNF_ACT_TANH_PRESCALED), CD_INST(2, 3, 12, CNF_ACT_TANH_PRESCALED),
    CD_EXACT(2, 12, CNF_ACT_SOFTPLUS), CD_EXACT(3, 12, CNF_ACT_SOFTPLUS), CD_EXACT(3, 16, CNF_ACT_SOFTPLUS),   // TestMode of the same flows
};

static const CoopDInst* cd_find(int HT_real, int L, int KZ, int ACT, int MODE) {
    const int A = HT_real / 4;
    const CoopDInst* best = nullptr;
    for (const CoopDInst& c : kCoopD) {
        const bool act_ok = c.ACT == ACT || (c.ACT == CNF_ACT_TANH_PRESCALED && ACT == CNF_ACT_TANH);
        if (c.A == A && c.L == L && c.MODE == MODE && c.ZR >= KZ && act_ok && (!best || c.ZR < best->ZR)) best = &c;
    }
    return best;
}

// H = widest hidden layer, D = state rows of the configuration; (HT, ZR) = the plan's layout
bool coopd_supported(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int exact, int C) {
    if (C < 0 || C > 16) return false;   // (conditions: one k-group of layer 1)
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    if (HT_real < 8 || HT_real > HT_lay || KZ > ZR_lay) return false;
    const CoopDInst* c = cd_find(HT_real, L, KZ, ACT, exact ? 1 : 0);
    if (!c) return coopd2_supported(HT_real, L, KZ, ACT, C, exact);   // 16 .. 24 hidden tiles: the 32-sample form (cnf_coop_d2.hip)
    // state registers beyond the plan's k-steps would read image k-groups that do not exist
    if ((c->ZR + 3) / 4 > (ZR_lay + 3) / 4) return false;
    // LDS: exchange buffer(s) + state / probe images + partial tiles (+ conditions) + C vectors; when that exceeds 160 KB the partial
    // tiles alias the exchange buffer, which they must then fit (coopd_launch makes the same decision)
    const int DT = c->ZR / 4, cvn = (1 + L) * 16 * HT_lay + 16 * ((ZR_lay + 3) / 4);   // b_1, w_1t, b_2 .. b_L, b_N
    if (coopd_lds_bytes(HT_real, L, DT, false, cvn, C > 0) <= 160 * 1024) return true;
    return coopd_lds_bytes(HT_real, L, DT, true, cvn, C > 0) <= 160 * 1024 && 4 * DT * 4 <= (L == 2 ? 1 : 2) * HT_real * 4;
}

// samples per super-tile of the form that serves the shape (64, or 32 for 16 .. 24 hidden tiles): the checkpoint arrays of the
// cooperative gradient are sized by it
int coopd_supertile(int H, int D, int L, int ACT, int exact) {
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    return cd_find(HT_real, L, KZ, ACT, exact ? 1 : 0) ? 64 : 32;
}

// the run-time view of a plan's packed image (layout (HT_lay, L, ZR_lay, CR_lay), backward images included) for a configuration with
// H hidden units and D state rows whose hidden tiles are dealt as 4 A + b
void dimg_fill(DImg& G, int H, int D, int L, int HT_lay, int ZR_lay, int CR_lay, int A_inst, int C) {
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    const MfmaLayout Y(HT_lay, L, ZR_lay, CR_lay, true);
    G.f1z = Y.f1z; G.fh = Y.fh; G.fN = Y.fN; G.bN = Y.bN; G.bh = Y.bh; G.b1 = Y.b1;
    G.v_b1 = Y.v_b1; G.v_w1t = Y.v_w1t; G.v_bh = Y.v_bh; G.v_bN = Y.v_bN;
    G.KPZ = Y.KGZ; G.HTP = Y.HT; G.imgH = MfmaLayout::imgA(Y.HT, Y.HT); G.vecH = MfmaLayout::vecC(Y.HT);
    G.b = HT_real - 4 * A_inst;
    const int ksH = (H + 3) / 4;
    G.KGH = HT_real; G.remH = ksH - 4 * (HT_real - 1);
    G.KGZ = (KZ + 3) / 4; G.remZ = KZ - 4 * (G.KGZ - 1);
    G.ckzr = ZR_lay;
    G.ck_ls = ZR_lay; G.ck_qs = 4;
    G.q_off = 0;
    G.f1y = Y.f1y; G.KPC = Y.KGC > 0 ? Y.KGC : 1; G.remC = (C + 3) / 4;   // condition k-steps (<= 4)
    G.cvn = Y.v_bN + MfmaLayout::vecC(Y.DT) - Y.v_b1;
    G.xalias = 0;
}

size_t coopd_rk_floats(int H, int D, int L, int ACT, int exact, int num_cus) {
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    if (const CoopDInst* c = cd_find(HT_real, L, KZ, ACT, exact ? 1 : 0)) return cd_rk_in_ring(c->A, c->L, c->ZR) ? (size_t)num_cus * 4 * 6 * c->ZR * 64 : 0;
    return L == 2 ? coopd2_rk_floats(HT_real, KZ, ACT, num_cus, exact) : 0;
}

hipError_t coopd_launch(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int CR_lay, const KArgs& k, int num_cus, hipStream_t st) {
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    const bool exact = k.exact == 1;
    if (exact && k.q_off <= 0) return hipErrorNotSupported;
    const CoopDInst* c = cd_find(HT_real, L, KZ, ACT, exact ? 1 : 0);
    const bool form32 = !c && coopd2_supported(HT_real, L, KZ, ACT, k.C, exact);
    if (!c && !form32) return hipErrorNotSupported;
    const int A_inst = c ? c->A : HT_real / 4;
    DArgs a{};
    a.k = k;
    DImg& G = a.g;
    dimg_fill(G, H, D, L, HT_lay, ZR_lay, CR_lay, A_inst, k.C);
    G.q_off = exact ? k.q_off : 0;
    if (k.C > 16 || (k.C > 0 && CR_lay < G.remC)) return hipErrorNotSupported;
    if ((k.ckpt || k.ckpt_k) && ZR_lay % 4 != 0) return hipErrorNotSupported;   // (checkpoint rows leave as 16-byte stores)
    if (form32 && k.ck_tiles) return hipErrorNotSupported;   // (the 32-sample form writes [tile][lane][ZR] rows only)
    if (form32) return coopd2_launch(HT_real, L, KZ, ACT, a, num_cus, st);
    const int DT = c->ZR / 4;
    G.xalias = coopd_lds_bytes(HT_real, L, DT, false, G.cvn, k.C > 0) <= 160 * 1024 ? 0 : 1;
    const int lds = coopd_lds_bytes(HT_real, L, DT, G.xalias != 0, G.cvn, k.C > 0);
    if (lds > 160 * 1024 || (G.xalias && 4 * DT * 4 > (L == 2 ? 1 : 2) * HT_real * 4)) return hipErrorNotSupported;
    const long long nst = (k.B + 63) / 64;
    const int nblocks = (int)(nst < num_cus ? nst : num_cus);
    // rows of the Runge-Kutta state in LDS where the instance has the form and the configuration's images leave the room
    if (c->fn_lds && !G.xalias && tuning().coopd != 3) {
        const int lds2 = (lds + 15) / 16 * 16 + c->NL * DT * 4 * 64 * 16;
        if (lds2 <= 160 * 1024) return c->fn_lds(a, lds2, nblocks, st);
    }
    if (cd_rk_in_ring(c->A, c->L, c->ZR) && !k.rk) return hipErrorNotSupported;   // the ring of the Runge-Kutta sums (coopd_rk_floats: plan-owned)
    return c->fn(a, lds, nblocks, st);
}

}  // namespace cnf
