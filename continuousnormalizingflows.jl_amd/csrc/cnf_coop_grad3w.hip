// cnf_coop_grad3w.hip - the second-order reverse sweep of the cooperative gradient's second form, TWO WORKGROUPS PER CU
// (round 6).
//
// Reference: the parameter gradient of `loss` through the fixed-step solve (Zygote through SciMLBase.solve with QuadratureAdjoint +
// ZygoteVJP, src/core/icnf.jl:90-99, driven by src/exts/mlj_ext/core_icnf.jl:42-51); the mathematics is DESIGN.md section 8 /
// cnf_coop_grad3.hip's header (same chains, same operands in, same tiles out).
//
// Why a second kernel for the same step: cnf_coop_grad3.hip runs one wave per SIMD with the whole register file (h_l and dbar_l
// parked in accumulation registers, the next stage's operands requested a stage ahead, the owners' costate rows in registers).
// Its s_memtime trace on the reference's default architecture (profiles/r6/r6i_sweep_phase_trace.txt, nvariables = 20: two hidden
// layers, D ~ H / 4) shows a stage of 45.5 k cycles carrying 22.6 k cycles of MFMA issue: the owners' dense phase, four barriers,
// five elementwise phases and the prologues of three short products are serial on a SIMD that has nothing else to run (40 %
// MFMA-busy in profiles/r6/r6z_nv20_grad_pmc.txt).  Two hidden layers do not need the whole register file: with nothing parked and
// nothing requested ahead a wave's live state is five tile sets.  This kernel is that form - at most 256 registers per wave, at
// most 80 KB of LDS per workgroup - so that TWO workgroups are resident per CU and one's elementwise phases, barriers and round
// trips run under the other's products:
//   * h_1 is read twice per stage from the stage store (at the start for vbar_1, behind the last H x H product's fragment
//     requests for sbar_1) instead of being parked; dbar_1 and dbar_2 stay in architectural registers;
//   * the owners keep no rows: z_n and the costate of the step are re-read per stage from the checkpoint / costate arrays (L2;
//     coalesced where the forward solve wrote them as tiles, KArgs::ck_tiles), Zbar_j of the running step's stages lives in the
//     kernel's global scratch (CGArgs::zb, L2); every row of a stage's dense phase is requested up front, unconditionally;
//   * nothing is requested a stage ahead: the other workgroup's products cover the round trips.
// Two hidden layers, HT = 4 A + b real hidden tiles (A = 2, 3), 32-sample super-tiles, tanh and softplus.  Same chains and the same
// summation order as cnf_coop_grad3.hip: the two sweeps agree bit for bit (tests/test_parity_gpu.py).
// Measured (default architecture, B = 32 768, 40 Tsit5 steps; profiles/r6/): 0.512 -> 0.451 ms per launch at nvariables = 20,
// 45.7 % MFMA-busy (39.8 % one per CU); what was tried on top of it and lost is in DESIGN.md section 8.6.
#include "cnf_coop_grad3_dev.h"

namespace cnf {

// LDS: the exchange buffers X0 (the partial tiles of Zbar alias it) and X1 [HT][2][64], the gbar and kbar images [DT][2][64]
constexpr int coop_grad3w_lds_bytes(int HT, int DT) {
    constexpr int NC = 2;
    const int part = 4 * DT * NC;
    const int x0 = HT * NC > part ? HT * NC : part;
    return (x0 + HT * NC + 2 * DT * NC) * 64 * 16;
}

// A: shared hidden tiles per wave (HT = 4 A + b, b run-time: the left-over tiles go one each to waves 3, 2, 1);
// KZ: state registers per lane (D <= 4 KZ, whole M-tiles); NS: stages of the instance
// H1L: h_1 of the stage waits in LDS between its two uses (this wave's own units: no barrier) where 80 KB have the room - A = 2 -
// instead of being read from the stage store a second time (one of a stage's five tile-set reads)
template <int A, int KZ, int ACT, int NS, bool H1L>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
coop_grad3w_step_kernel(G3Args ga) {
    constexpr int NC = 2;
    constexpr bool LO = true;
    using U = G3U<A, NC>;
    static_assert(ACT == CNF_ACT_TANH_PRESCALED || ACT == CNF_ACT_SOFTPLUS, "act' and act'' are rebuilt from h: tanh and softplus");
    static_assert(KZ % 4 == 0, "state registers in whole M-tiles");
    const CGArgs& a = ga.c.c;
    const CG3Args& q3 = ga.c;
    const DImg& G = ga.g;
    constexpr int DT = KZ / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HT = 4 * A + G.b;
    const int X0N = (HT * NC > 4 * DT * NC ? HT * NC : 4 * DT * NC) * 64;
    f32x4* X0 = reinterpret_cast<f32x4*>(smem);        // [HT][NC][64]; the partial tiles [4 waves][DT][NC][64] alias it
    f32x4* X1 = X0 + X0N;                              // [HT][NC][64]
    f32x4* gbuf = X1 + HT * NC * 64;                   // [DT][NC][64]: gbar
    f32x4* kbuf = gbuf + DT * NC * 64;                 // [DT][NC][64]: kbar
    f32x4* hbuf = kbuf + DT * NC * 64;                 // (H1L) [HT][NC][64]: h_1 of the stage
    f32x4* pbuf = X0;
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool owner = wave < NC;
    const int D = a.D;
    const long long B = a.B;
    const long long nst = a.ntiles_pad / NC;
    const DRs R0{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, 0x7fffffff, 0x00020000), (unsigned)lane * 16u};
    const float inv_fs = ACT == CNF_ACT_TANH_PRESCALED ? 1.f / kTanhPrescale : 1.f;   // the forward images of tanh nets carry the pre-scale
    const int ns = a.T.ns < NS ? a.T.ns : NS;
    const float dt = a.dt, tn = a.tn;
    const int ckzr = G.ckzr, ckls = G.ck_ls, ckqs = G.ck_qs;   // (cnf_coop_d_dev.h: the two layouts of the checkpoint rows)
    const int un = 3 - wave;
    const bool v0 = LO && un < G.b;
    const int tR = (4 * A + un < HT - 1) ? 4 * A + un : HT - 1;
    const int mtS0 = wave * A;
    const G3Off<A> TZ = g3_offsets<A>(R0, G.KPZ, mtS0, tR);
    const G3Off<A> TH = g3_offsets<A>(R0, G.HTP, mtS0, tR);
    unsigned vd[DT];
#pragma unroll
    for (int dm = 0; dm < DT; ++dm) { vd[dm] = R0.lane16 + (unsigned)(dm * G.HTP) * 1024u; asm volatile("" : "+v"(vd[dm])); }
    const unsigned F1Z = (unsigned)G.f1z * 4u, FH = (unsigned)G.fh * 4u, BN = (unsigned)G.bN * 4u, BH = (unsigned)G.bh * 4u, B1 = (unsigned)G.b1 * 4u;
    const int HTs = q3.HTs, DTZ = q3.DTZ;
    const long long ntp = a.ntiles_pad;
    // byte offsets of this wave's units inside a column-tile PAIR of an [..][ntp][HTs] tile array (sample tile c, hidden tile mt)
    unsigned uo[A][NC], uoR[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int m = 0; m < A; ++m) { uo[m][c] = (unsigned)lane * 16u + (unsigned)((c * HTs + mtS0 + m) * 1024); asm volatile("" : "+v"(uo[m][c])); }
        uoR[c] = (unsigned)lane * 16u + (unsigned)((c * HTs + tR) * 1024); asm volatile("" : "+v"(uoR[c]));
    }
    const unsigned arr_bytes = (unsigned)((long long)ns * ntp * HTs * 1024);
    auto rsrc = [&](const float* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)arr_bytes, 0x00020000); };
    auto load_units = [&](const float* arr, unsigned so, U& u) {
        const __amdgpu_buffer_rsrc_t r = rsrc(arr);
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) u.S[m][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)uo[m][c], (int)so, 0));
#pragma unroll
        for (int c = 0; c < NC; ++c) u.R[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)uoR[c], (int)so, 0));
    };
    auto store_units = [&](float* arr, unsigned so, const U& u) {
        const __amdgpu_buffer_rsrc_t r = rsrc(arr);
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, u.S[m][c]), r, (int)uo[m][c], (int)so, 2);
                CNF_STORE_DATA_HAZARD(u.S[m][c]);
            }
        if (v0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, u.R[c]), r, (int)uoR[c], (int)so, 2);
                CNF_STORE_DATA_HAZARD(u.R[c]);
            }
        }
    };
    auto publish = [&](f32x4* __restrict__ xb, const U& v) {
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) xb[((mtS0 + m) * NC + c) * 64 + lane] = v.S[m][c];
        if (v0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) xb[(tR * NC + c) * 64 + lane] = v.R[c];
        }
    };
    auto zero_u = [&](U& u) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) u.S[m][c] = z;
#pragma unroll
        for (int c = 0; c < NC; ++c) u.R[c] = z;
    };
    // vbar = (acc / fs) .* act'(h); acc <- dbar
    auto up_ew = [&](U& acc, const U& h, U& vb) {
        auto one = [&](f32x4& ac, const f32x4& hh, f32x4& v) {
            ac = ac * inv_fs;
            v = ac * dact_from_h<ACT>(hh);
        };
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) one(acc.S[m][c], h.S[m][c], vb.S[m][c]);
#pragma unroll
        for (int c = 0; c < NC; ++c) one(acc.R[c], h.R[c], vb.R[c]);
    };
    // sbar = hbar .* act' + dbar .* G(h, delta): in place of hbar
    auto down_ew = [&](U& hb, const U& h, const U& dl, const U& db) {
        auto one = [&](f32x4& x, const f32x4& hh, const f32x4& dd, const f32x4& bb) {
            const f32x4 d = dact_from_h<ACT>(hh);
            x = g3_sbar(x, d, bb, g3_G<ACT>(hh, dd, d));
        };
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) one(hb.S[m][c], h.S[m][c], dl.S[m][c], db.S[m][c]);
#pragma unroll
        for (int c = 0; c < NC; ++c) one(hb.R[c], h.R[c], dl.R[c], db.R[c]);
    };

    for (long long st = blockIdx.x; st < nst; st += gridDim.x) {
        const long long smp0 = st * (16 * NC);
        const long long smp = smp0 + (owner ? wave : 0) * 16 + n;
        const bool valid = owner && smp < B;
        const long long sc = smp < B ? smp : B - 1;
        const long long tile = st * NC + (owner ? wave : 0);
        // the owners' rows: eps in registers (fixed for the solve), z_n and the costate re-read per stage (f32x4 per 16-row group)
        float eps[KZ];
#pragma unroll
        for (int s = 0; s < KZ; ++s) eps[s] = 0.f;
        const float* znp = a.ckpt + ((long long)a.step * ntp + tile) * 64 * ckzr + lane * ckls;
        f32x4* lamp = reinterpret_cast<f32x4*>(a.lam + (tile * 64 + lane) * KZ);
        // Zbar_j of this wave's sample tile: [NS][DT] f32x4 per lane in the kernel's global scratch (L2; written and read by the same lane)
        f32x4* zbt = reinterpret_cast<f32x4*>(a.zb) + ((tile * NS * DT) * 64 + lane);
        if (owner) {
#pragma unroll
            for (int s = 0; s < KZ; ++s) {
                const int f = 4 * s + g;
                eps[s] = f < D ? a.eps[sc * D + f] : 0.f;
            }
            if (a.step == a.nsteps - 1) {
                // the costate at t_1 (cotangent of the loss terms of the final state): made here, kept where every later step keeps it
                float lam[KZ];
#pragma unroll
                for (int s = 0; s < KZ; ++s) lam[s] = valid ? a.ckpt[((long long)a.nsteps * ntp + tile) * 64 * ckzr + lane * ckls + (s >> 2) * ckqs + (s & 3)] : 0.f;
                if (a.lam3 != 0.f) {
                    float sa = 0.f;
#pragma unroll
                    for (int s = 0; s < KZ; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) sa = fmaf(lam[s], lam[s], sa); }
                    sa = group_sum(sa);
                    const float inv = sa > 0.f ? a.lam3 * rsqrtf(sa) : 0.f;
#pragma unroll
                    for (int s = 0; s < KZ; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) lam[s] = fmaf(inv, lam[s], lam[s]); }
                }
#pragma unroll
                for (int q = 0; q < DT; ++q) lamp[q] = f32x4{lam[4 * q], lam[4 * q + 1], lam[4 * q + 2], lam[4 * q + 3]};
            }
        }
        __syncthreads();                 // the previous super-tile's readers of the LDS images are done
        f32x4 aS[A], aR = {0.f, 0.f, 0.f, 0.f};
        auto stage_off = [&](int is) { return (unsigned)(((long long)is * ntp + st * NC) * HTs * 1024); };

#pragma clang loop unroll(disable)
        for (int i = ns - 1; i >= 0; --i) {
            // (the image's buffer resource is rebuilt per stage from the kernel argument made scalar by hand: see cnf_coop_dgrad.hip)
            const unsigned long long pimg = (unsigned long long)a.packed;
            const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pimg), phi = __builtin_amdgcn_readfirstlane((unsigned)(pimg >> 32));
            const DRs R{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)phi << 32) | plo), 0, 0x7fffffff, 0x00020000), R0.lane16};
            const float bi = a.T.b[i];
            const float cl = valid ? dt * bi : 0.f;      // cotangent of ldot (dL/d dlogp = +1 per column); zero for padding columns
            const unsigned soH = stage_off(i);
#ifdef G3_TRACE
            unsigned long long tr[16];
#endif
            G3_T(0);
            U h1;
            if (owner) {
                // ---- dense phase: stage state z_s, kbar, gbar (cnf_coop_grad3.hip's algebra) ----
                // Every row the phase needs is requested up front, unconditionally (one round trip; a load behind a branch makes the
                // compiler's wait-count insertion drain the memory counter at the join - the first build of this phase took 27 k
                // cycles that way): slot j of a 16-row group is the stage derivative k_j where the tableau's a[i][j] multiplies it
                // (j < i) and Zbar_{j+1} where a[j+1][i] does (j >= i) - NS - 1 rows, not 2 (NS - 1).
                const long long rowb = (long long)a.step * ns * ntp + tile, rstride = ntp * 64 * (long long)ckzr;
                const float* kbase = a.ckpt_k + rowb * 64 * ckzr + lane * ckls;
                const float* gbase = (a.lam2 != 0.f ? a.ckpt_g : a.ckpt_k) + rowb * 64 * ckzr + lane * ckls;
                f32x4 row[NS - 1][DT], zn[DT], lm[DT], ki[DT], gi[DT];
                float ca[NS - 1], ck[NS - 1];
#pragma unroll
                for (int j = 0; j < NS - 1; ++j) {
                    const bool lower = j < i;                         // a[i][j] = 0 for j >= i
                    const bool upper = !lower && j + 1 < ns;          // Zbar_j exists for i < j < ns only
                    ca[j] = lower ? a.T.a[i][j] : 0.f;
                    ck[j] = upper ? a.T.a[j + 1][i] : 0.f;
                    const f32x4* src = upper ? zbt + ((j + 1) * DT) * 64 : reinterpret_cast<const f32x4*>(kbase + (j < ns ? j : ns - 1) * rstride);
                    const int qs = upper ? 64 : (ckqs >> 2);          // f32x4 stride between the 16-row groups of the source
#pragma unroll
                    for (int q = 0; q < DT; ++q) row[j][q] = src[q * qs];
                }
#pragma unroll
                for (int q = 0; q < DT; ++q) {
                    zn[q] = *reinterpret_cast<const f32x4*>(znp + q * ckqs);
                    lm[q] = lamp[q];
                    ki[q] = *reinterpret_cast<const f32x4*>(kbase + i * rstride + q * ckqs);
                    gi[q] = *reinterpret_cast<const f32x4*>(gbase + i * rstride + q * ckqs);
                }
                load_units(q3.fh[0], soH, h1);               // h_1 of this stage: needed behind the first product
                // gbar = cotangent of g = eps^T J: -c_l eps (+ c_n g / |g|);  kbar += c_E zdot / |zdot|  (src/core/icnf.jl:184-251)
                float inv1 = 0.f, inv2 = 0.f;
                if (a.lam1 != 0.f) {
                    float e2 = 0.f;
#pragma unroll
                    for (int q = 0; q < DT; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) e2 = fmaf(ki[q][e], ki[q][e], e2);
                    e2 = group_sum(e2);
                    inv1 = e2 > 0.f ? cl * a.lam1 * rsqrtf(e2) : 0.f;
                }
                if (a.lam2 != 0.f) {
                    float n2 = 0.f;
#pragma unroll
                    for (int q = 0; q < DT; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) n2 = fmaf(gi[q][e], gi[q][e], n2);
                    n2 = group_sum(n2);
                    inv2 = n2 > 0.f ? cl * a.lam2 * rsqrtf(n2) : 0.f;
                }
                const long long ctile = (long long)i * ntp + tile;
                f32x4* gbp = reinterpret_cast<f32x4*>(q3.gb) + (ctile * DTZ) * 64 + lane;
                f32x4* ztp = reinterpret_cast<f32x4*>(q3.zt) + (ctile * DTZ) * 64 + lane;
                const int DTs = q3.DTs;              // tiles of the D-row arrays: the configuration's (<= the instance's DT)
                f32x4* epp = reinterpret_cast<f32x4*>(q3.ep) + (ctile * DTs) * 64 + lane;
                f32x4* kbp = reinterpret_cast<f32x4*>(q3.kb) + (ctile * DTs) * 64 + lane;
                const int kt = D >> 4, et = (D & 15) >> 2, gt = D & 3;   // where the time row (feature D) sits
                const float tt = tn + a.T.c[i] * dt;
#pragma unroll
                for (int q = 0; q < DT; ++q) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    f32x4 kb = lm[q] * bi;
#pragma unroll
                    for (int j = 0; j < NS - 1; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[e] = fmaf(ca[j], row[j][q][e], acc[e]);
                            kb[e] = fmaf(ck[j], row[j][q][e], kb[e]);
                        }
                    f32x4 zs, kbar, gbar;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        zs[e] = fmaf(dt, acc[e], zn[q][e]);
                        kbar[e] = valid ? dt * kb[e] : 0.f;
                        gbar[e] = -cl * eps[4 * q + e];
                        kbar[e] = fmaf(inv1, ki[q][e], kbar[e]);         // (inv = 0 without the regulariser)
                        gbar[e] = fmaf(inv2, gi[q][e], gbar[e]);
                    }
                    gbuf[(q * NC + wave) * 64 + lane] = gbar;
                    kbuf[(q * NC + wave) * 64 + lane] = kbar;
                    // the D-row operands of Wbar_1 (= delta_1 gbar^T + sbar_1 [z; t]^T) and Wbar_N (= eps cbar^T + kbar h_L^T) as tiles
                    if (q < DTZ) {
                        f32x4 zv = zs;
                        if (!a.autonomous && q == kt && g == gt) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (e == et) zv[e] = tt;
                        }
                        if (!valid) zv = f32x4{0.f, 0.f, 0.f, 0.f};
                        gbp[q * 64] = gbar;
                        ztp[q * 64] = zv;
                    }
                    if (q < DTs) {
                        epp[q * 64] = f32x4{eps[4 * q], eps[4 * q + 1], eps[4 * q + 2], eps[4 * q + 3]};
                        kbp[q * 64] = kbar;
                    }
                }
                // (the input side has one 16-row group more than the state when the time row opens it: D = 16 k)
                for (int q = DT; q < DTZ; ++q) {
                    f32x4 zv = {0.f, 0.f, 0.f, 0.f};
                    if (!a.autonomous && q == kt && g == gt && valid) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (e == et) zv[e] = tt;
                    }
                    gbp[q * 64] = f32x4{0.f, 0.f, 0.f, 0.f};
                    ztp[q * 64] = zv;
                }
            } else {
                load_units(q3.fh[0], soH, h1);
            }
            U acc;
            zero_u(acc);
            g3_load_a<A, LO>(R, TZ, F1Z, 0, aS, aR);
            G3_T(1);
            G3_SYNC();                                                                     // gbar / kbar published
            G3_T(2);
            // ================= up 1: dbar_1 = W_1[:,0:D] gbar =================
            U h2;
            g3_gemm<A, LO, NC>(R, TZ, F1Z, G.KGZ, G.remZ, v0, gbuf, lane, aS, aR, acc, [&]() { load_units(q3.fh[1], soH, h2); });
            G3_T(3);
            g3_load_a<A, LO>(R, TH, FH, 0, aS, aR);
            U d1, d2;
            {
                U vb;
                up_ew(acc, h1, vb);                      // acc <- dbar_1, vb <- vbar_1
                publish(X0, vb);
                if constexpr (H1L) {
#pragma unroll
                    for (int m = 0; m < A; ++m)
#pragma unroll
                        for (int c = 0; c < NC; ++c) hbuf[((mtS0 + m) * NC + c) * 64 + lane] = h1.S[m][c];
#pragma unroll
                    for (int c = 0; c < NC; ++c) hbuf[(tR * NC + c) * 64 + lane] = h1.R[c];   // (a wave without a left-over tile holds a clamped copy of another wave's: the same values into the same slot)
                }
                d1 = acc;
                zero_u(acc);
                G3_T(4);
                G3_SYNC();
                G3_T(5);
                // ================= up 2: dbar_2 = W_2 vbar_1 =================
                g3_gemm<A, LO, NC>(R, TH, FH, G.KGH, G.remH, v0, X0, lane, aS, aR, acc, [&]() {});
                G3_T(6);
                g3_load_a<A, LO>(R, TZ, BN, 0, aS, aR);
                store_units(q3.sv[0], soH, vb);
            }
            U dl2;
            {
                U vb;
                up_ew(acc, h2, vb);                      // acc <- dbar_2, vb <- vbar_2 (= cbar)
                d2 = acc;
                zero_u(acc);
                G3_T(7);
                // ================= the top: hbar_2 = W_N^T kbar =================
                g3_gemm<A, LO, NC>(R, TZ, BN, G.KGZ, G.remZ, v0, kbuf, lane, aS, aR, acc, [&]() { load_units(q3.fd[1], soH, dl2); });
                G3_T(8);
                g3_load_a<A, LO>(R, TH, BH, 0, aS, aR);
                store_units(q3.sv[1], soH, vb);
            }
            down_ew(acc, h2, dl2, d2);                   // acc <- sbar_2
            publish(X1, acc);
            U s2 = acc;
            zero_u(acc);
            G3_T(9);
            G3_SYNC();
            G3_T(10);
            // ================= down: hbar_1 = W_2^T sbar_2 =================
            U dl1;
            g3_gemm<A, LO, NC>(R, TH, BH, G.KGH, G.remH, v0, X1, lane, aS, aR, acc,
                               [&]() { if constexpr (!H1L) load_units(q3.fh[0], soH, h1); load_units(q3.fd[0], soH, dl1); });
            if constexpr (H1L) {
#pragma unroll
                for (int m = 0; m < A; ++m)
#pragma unroll
                    for (int c = 0; c < NC; ++c) h1.S[m][c] = hbuf[((mtS0 + m) * NC + c) * 64 + lane];
#pragma unroll
                for (int c = 0; c < NC; ++c) h1.R[c] = hbuf[(tR * NC + c) * 64 + lane];
            }
            G3_T(11);
            // the fragments of the Zbar product (this wave's own k-groups of W_1[:,0:D]^T)
            constexpr int NZ = A + 1;
            f32x4 fz[NZ][DT];
#pragma unroll
            for (int m = 0; m < NZ; ++m)
#pragma unroll
                for (int dm = 0; dm < DT; ++dm) fz[m][dm] = dloadv(R, vd[dm], B1 + (unsigned)(m < A ? mtS0 + m : tR) * 1024u);
            store_units(q3.ss[1], soH, s2);
            down_ew(acc, h1, dl1, d1);                   // acc <- sbar_1
            G3_T(12);
            // ================= Zbar_i = W_1[:,0:D]^T sbar_1: partial tiles over this wave's own k-groups, from registers =================
            f32x4 part[DT][NC];
#pragma unroll
            for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                for (int c = 0; c < NC; ++c) part[dm][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < NZ; ++m) {
                if (m < A) {
                    const int js = (mtS0 + m == G.KGH - 1) ? G.remH : 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < js) {
#pragma unroll
                            for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                                for (int c = 0; c < NC; ++c) part[dm][c] = mfma4(fz[m][dm][j], acc.S[m][c][j], part[dm][c]);
                        }
                } else if (v0) {
                    // (a left-over tile may be the last k-group: its k-steps beyond `rem` multiply zero columns of the image)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                            for (int c = 0; c < NC; ++c) part[dm][c] = mfma4(fz[m][dm][j], acc.R[c][j], part[dm][c]);
                }
            }
            store_units(q3.ss[0], soH, acc);
            // (X0 - the partial tiles alias it - was last read by the product before the last one of the stage, and every wave has passed
            // the barrier in front of the last one since: no barrier here)
#pragma unroll
            for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                for (int c = 0; c < NC; ++c) pbuf[((wave * DT + dm) * NC + c) * 64 + lane] = part[dm][c];
            G3_T(13);
            G3_SYNC();
            G3_T(14);
            if (owner) {
#pragma unroll
                for (int dm = 0; dm < DT; ++dm) {
                    f32x4 v = pbuf[((0 * DT + dm) * NC + wave) * 64 + lane];
#pragma unroll
                    for (int w = 1; w < 4; ++w) v += pbuf[((w * DT + dm) * NC + wave) * 64 + lane];
                    zbt[(i * DT + dm) * 64] = v;
                }
            }
#ifdef G3_TRACE
            G3_T(15);
            if (blockIdx.x == 3 && st == 3 && a.step == 5 && i == 2 && lane == 0) {
#define G3_D(k) (int)(tr[k] - tr[k - 1])
                printf("w%d: dense %d B0 %d up1 %d ew1 %d B1 %d up2 %d ew2 %d top %d ewtop %d B %d down %d ewdown %d zbar %d B %d red %d | stage %d\n", wave, G3_D(1), G3_D(2), G3_D(3),
                       G3_D(4), G3_D(5), G3_D(6), G3_D(7), G3_D(8), G3_D(9), G3_D(10), G3_D(11), G3_D(12), G3_D(13), G3_D(14), G3_D(15), (int)(tr[15] - tr[0]));
            }
#endif
            // (the next stage's first LDS writes - gbar / kbar images - touch neither exchange buffer; its first publish into X0 comes
            // behind its first barrier, which the owners reach after this sum)
        }
        if (owner) {
            float lam[KZ];
#pragma unroll
            for (int q = 0; q < DT; ++q) {
                f32x4 acc = lamp[q];
                for (int j = 0; j < ns; ++j) acc += zbt[(j * DT + q) * 64];
                lamp[q] = acc;
#pragma unroll
                for (int e = 0; e < 4; ++e) lam[4 * q + e] = acc[e];
            }
            if (a.step == 0 && a.grad_x && valid) {
#pragma unroll
                for (int s = 0; s < KZ; ++s) {
                    const int f = 4 * s + g;
                    if (f < a.nvars) a.grad_x[smp * a.nvars + f] = lam[s];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
template <int A, int KZ, int ACT, int NS, bool H1L>
static hipError_t launch_g3w(const G3Args& a, int lds, int nblocks, hipStream_t st) {
    auto kern = coop_grad3w_step_kernel<A, KZ, ACT, NS, H1L>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (!once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

struct G3WInst {
    int A, KZ, ACT;
    hipError_t (*fn[2])(const G3Args&, int, int, hipStream_t);   // [0] RK4 (4 stages), [1] Tsit5 (6 stages)
    hipError_t (*fn_h1l[2])(const G3Args&, int, int, hipStream_t);   // the same with h_1 kept in LDS (A = 2), or null
};
#define G3W_INST(A, KZ, ACT) G3WInst { A, KZ, ACT, { &launch_g3w<A, KZ, ACT, 4, false>, &launch_g3w<A, KZ, ACT, 6, false> }, \
                                       { A == 2 ? &launch_g3w<2, KZ, ACT, 4, true> : nullptr, A == 2 ? &launch_g3w<2, KZ, ACT, 6, true> : nullptr } }
static const G3WInst kG3W[] = {
    G3W_INST(2, 8, CNF_ACT_SOFTPLUS), G3W_INST(2, 12, CNF_ACT_SOFTPLUS),
    G3W_INST(2, 8, CNF_ACT_TANH_PRESCALED), G3W_INST(2, 12, CNF_ACT_TANH_PRESCALED),
    G3W_INST(3, 8, CNF_ACT_SOFTPLUS), G3W_INST(3, 12, CNF_ACT_SOFTPLUS), G3W_INST(3, 16, CNF_ACT_SOFTPLUS),
    G3W_INST(3, 8, CNF_ACT_TANH_PRESCALED), G3W_INST(3, 12, CNF_ACT_TANH_PRESCALED), G3W_INST(3, 16, CNF_ACT_TANH_PRESCALED),
};
static const G3WInst* g3w_find(int HT_real, int L, int KZ, int ACT) {
    if (L != 2) return nullptr;
    const int A = HT_real / 4;
    const G3WInst* best = nullptr;
    for (const G3WInst& c : kG3W) {
        const bool act_ok = c.ACT == ACT || (c.ACT == CNF_ACT_TANH_PRESCALED && ACT == CNF_ACT_TANH);
        if (c.A == A && c.KZ >= KZ && act_ok && (!best || c.KZ < best->KZ)) best = &c;
    }
    return best;
}

// Does the two-per-CU sweep serve the shape (two hidden layers; the checkpoint rows are read KZ registers wide; 80 KB of LDS)?
bool coop_grad3w_supported(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int CR_lay) {
    if (CR_lay != 0) return false;
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    if (HT_real > HT_lay || KZ > ZR_lay) return false;
    const G3WInst* c = g3w_find(HT_real, L, KZ, ACT);
    if (!c || c->KZ > ZR_lay) return false;
    return coop_grad3w_lds_bytes(HT_real, c->KZ / 4) <= 80 * 1024;
}

hipError_t coop_grad3w_step_launch(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, const CG3Args& a, int num_cus, hipStream_t st) {
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    const G3WInst* c = g3w_find(HT_real, L, KZ, ACT);
    if (!c) return hipErrorNotSupported;
    G3Args ga{};
    ga.c = a;
    dimg_fill(ga.g, H, D, L, HT_lay, ZR_lay, 0, c->A, 0);
    if (a.ck_tiles) { ga.g.ck_ls = 4; ga.g.ck_qs = 256; }
    const int lds = coop_grad3w_lds_bytes(HT_real, c->KZ / 4);
    if (lds > 80 * 1024) return hipErrorNotSupported;
    const long long nst = a.c.ntiles_pad / 2;
    const int nblocks = (int)(nst < 2LL * num_cus ? nst : 2LL * num_cus);
    const int si = a.c.T.ns <= 4 ? 0 : 1;
    const int lds_h1 = lds + HT_real * 2 * 64 * 16;
    if (c->fn_h1l[si] && lds_h1 <= 80 * 1024) return c->fn_h1l[si](ga, lds_h1, nblocks, st);
    return c->fn[si](ga, lds, nblocks, st);
}

}  // namespace cnf
