// cnf_coop_grad3.hip - the SECOND-ORDER reverse sweep of the cooperative gradient (round 6).
//
// Reference: the parameter gradient of `loss` through the fixed-step solve (Zygote through SciMLBase.solve with QuadratureAdjoint +
// ZygoteVJP, src/core/icnf.jl:90-99, driven by src/exts/mlj_ext/core_icnf.jl:42-51); the mathematics is DESIGN.md section 8 /
// cnf_coop_grad.hip's header.
//
// The sweeps of rounds 3 - 5 (cnf_coop_grad.hip, cnf_coop_dgrad.hip) RECOMPUTE, per stage, the forward chain h_l and the
// first-order pullback delta_l - the two chains the forward solve has just run - beside the two second-order chains that need
// them: half of a stage's H x H products are repeats, and every operand of the deferred weight-cotangent products (X_l =
// [delta_l | sbar_l], Y_l = [vbar_l | h_l]) leaves through an LDS transposition.  Here the forward solve STORES h_l and delta_l of
// every stage as it computes them (tile-native, cnf_tiles.h: 2 L H floats per sample and stage - 32 GB at cfg4, a sixth of one
// GPU's HBM, written under a compute-bound kernel), and this kernel runs the second-order chains alone:
//     up:    dbar_1 = W_1[:,0:D] gbar,   vbar_l = dbar_l .* act'(h_l),   dbar_{l+1} = W_{l+1} vbar_l
//     top:   hbar_L = W_N^T kbar
//     down:  sbar_l = hbar_l .* act'_l + dbar_l .* G(h_l, delta_l),   hbar_{l-1} = W_l^T sbar_l,   Zbar = W_1[:,0:D]^T sbar_1
//            (G = delta act'' / act':  tanh -2 h delta,  softplus delta (1 - act'))
// Every product is ONE chain over a 32-sample super-tile (two column tiles per weight fragment, as the two-chain products of the
// older sweeps), no activation is evaluated (act' comes from h: an FMA for tanh, one v_exp for softplus), vbar_l and sbar_l leave
// for the cotangent products straight from the registers that hold them (cnf_wgrad_tiles.hip reads tiles), and the other half of
// those products' operands is the forward solve's store.  Per stage and sample: 4 (L - 1) H^2 + 8 H D multiply-adds instead of
// 8 (L - 1) H^2 + 16 H D.
//
// Organisation (cnf_coop_dgrad.hip's): a workgroup of four waves, one per SIMD with the whole register file, owns a super-tile;
// the real hidden tiles HT = 4 A + b are dealt - wave w takes tiles [w A, (w + 1) A) with both column tiles, the b left-over
// tiles go one each to waves 3, 2, 1; waves 0 and 1 own the two sample tiles (dense D-row state, Runge-Kutta adjoint); Zbar is
// split along K by ownership (every wave multiplies the sbar_1 tiles it has just produced); h_l and dbar_l of the lower layers wait
// in accumulation registers between the way up and the way down.  Barriers order LDS traffic only.
#include "cnf_coop_grad3_dev.h"

namespace cnf {

// LDS: two exchange buffers [HT][2][64] (the partial tiles of Zbar alias the first), the gbar and kbar images [DT][2][64], C vectors
// are not needed (no bias enters a second-order chain)
constexpr int coop_grad3_lds_bytes(int HT, int DT, int NC, int NS = 6) {
    const int part = 4 * DT * NC;                      // [4 waves][DT][NC] partial tiles
    const int x0 = HT * NC > part ? HT * NC : part;
    return (x0 + HT * NC + 2 * DT * NC + (NC == 2 ? 2 * NS * DT : 0)) * 64 * 16;   // ... + (NC = 2) Zbar_j of the running step: [2 owners][NS][DT]
}

// A: shared hidden tiles per wave; LO: the instance serves left-over tiles (HT = 4 A + b, b run-time); L: hidden layers;
// KZ: state registers per lane (D <= 4 KZ, whole M-tiles); NS: stages of the instance
// PF: what a stage needs from HBM at its start (h_1, delta_L, the checkpoint rows) is requested a stage ahead (the A = 4, L = 3 instance
// has no registers for it: 472 -> 512 with spills, 0.77 -> 0.86 ms per launch at cfg4; two hidden layers gain 11 %)
// NC: sample tiles of a super-tile = column tiles of every product (2: 32 samples, waves 0 and 1 own the sample tiles; 4: 64 samples,
// every wave owns one - a stage's fixed costs (barriers, the dense phase, prologues of the short products) are paid per twice the
// MFMAs, and no wave idles through the dense phase: the form of the two-hidden-layer nets, whose stages are short)
template <int A, bool LO, int L, int KZ, int ACT, int NS, int NC, bool PF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
coop_grad3_step_kernel(G3Args ga) {
    using U = G3U<A, NC>;
    constexpr bool ZL = NC == 2;                     // Zbar_j of the running step in LDS (else in the global scratch CGArgs::zb: L2)
    static_assert(ACT == CNF_ACT_TANH_PRESCALED || ACT == CNF_ACT_SOFTPLUS, "act' and act'' are rebuilt from h: tanh and softplus");
    static_assert(KZ % 4 == 0, "state registers in whole M-tiles");
    static_assert(L == 2 || L == 3, "two or three hidden layers");
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
    f32x4* zbL = kbuf + DT * NC * 64;                  // (ZL) [2 owners][NS][DT][64]: Zbar_j of the running step's stages (owner-private)
    f32x4* pbuf = X0;
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool owner = wave < NC;
    const int D = a.D;
    const long long B = a.B;
    const long long nst = a.ntiles_pad / NC;   // every group of NC of the checkpoint arrays' tiles: the groups behind the batch write zeros the cotangent products read
    const DRs R0{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, 0x7fffffff, 0x00020000), (unsigned)lane * 16u};
    const float inv_fs = ACT == CNF_ACT_TANH_PRESCALED ? 1.f / kTanhPrescale : 1.f;   // the forward images of tanh nets carry the pre-scale
    const int ns = a.T.ns < NS ? a.T.ns : NS;
    const float dt = a.dt, tn = a.tn;
    const int ckzr = G.ckzr;   // (lane-major checkpoint rows only: the tile layout of KArgs::ck_tiles is the two-per-CU sweep's - its two run-time strides cost the 512-register instances here 14 ... 42 spilled registers)
    // this wave's left-over tile (b of them, one each to waves 3, 2, 1; clamped for the loads)
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
    const unsigned IMGH = (unsigned)G.imgH * 4u;
    const int HTs = q3.HTs, DTZ = q3.DTZ;
    const long long ntp = a.ntiles_pad;
    // byte offsets of this wave's units inside a column-tile PAIR of an [..][ntp][HTs] tile array (sample tile q, hidden tile mt)
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
        if constexpr (LO) {
#pragma unroll
            for (int c = 0; c < NC; ++c) u.R[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)uoR[c], (int)so, 0));
        }
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
        if constexpr (LO) {
            if (v0) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, u.R[c]), r, (int)uoR[c], (int)so, 2);
                    CNF_STORE_DATA_HAZARD(u.R[c]);
                }
            }
        }
    };
    auto publish = [&](f32x4* __restrict__ xb, const U& v) {
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) xb[((mtS0 + m) * NC + c) * 64 + lane] = v.S[m][c];
        if constexpr (LO) {
            if (v0) {
#pragma unroll
                for (int c = 0; c < NC; ++c) xb[(tR * NC + c) * 64 + lane] = v.R[c];
            }
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
    auto park_u = [&](const U& u, U& p) {
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) p.S[m][c] = park4(u.S[m][c]);
        if constexpr (LO) {
#pragma unroll
            for (int c = 0; c < NC; ++c) p.R[c] = park4(u.R[c]);
        }
    };
    auto unpark_u = [&](const U& p, U& u) {
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) u.S[m][c] = unpark4(p.S[m][c]);
        if constexpr (LO) {
#pragma unroll
            for (int c = 0; c < NC; ++c) u.R[c] = unpark4(p.R[c]);
        }
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
        if constexpr (LO) {
#pragma unroll
            for (int c = 0; c < NC; ++c) one(acc.R[c], h.R[c], vb.R[c]);
        }
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
        if constexpr (LO) {
#pragma unroll
            for (int c = 0; c < NC; ++c) one(hb.R[c], h.R[c], dl.R[c], db.R[c]);
        }
    };
    auto publish_dense = [&](f32x4* img, int ct, const float (&v)[KZ]) {
#pragma unroll
        for (int kg = 0; kg < DT; ++kg) img[(kg * NC + ct) * 64 + lane] = f32x4{v[4 * kg], v[4 * kg + 1], v[4 * kg + 2], v[4 * kg + 3]};
    };

    for (long long st = blockIdx.x; st < nst; st += gridDim.x) {
        const long long smp0 = st * (16 * NC);
        const long long smp = smp0 + (owner ? wave : 0) * 16 + n;
        const bool valid = owner && smp < B;
        const long long sc = smp < B ? smp : B - 1;
        const long long tile = st * NC + (owner ? wave : 0);
        float eps[KZ], zn[KZ], lam[KZ];
#pragma unroll
        for (int s = 0; s < KZ; ++s) { eps[s] = 0.f; zn[s] = 0.f; lam[s] = 0.f; }
        if (owner) {
#pragma unroll
            for (int s = 0; s < KZ; ++s) {
                const int f = 4 * s + g;
                eps[s] = f < D ? a.eps[sc * D + f] : 0.f;
                zn[s] = a.ckpt[(((long long)a.step * ntp + tile) * 64 + lane) * ckzr + s];
            }
            if (a.step == a.nsteps - 1) {
#pragma unroll
                for (int s = 0; s < KZ; ++s) lam[s] = valid ? a.ckpt[(((long long)a.nsteps * ntp + tile) * 64 + lane) * ckzr + s] : 0.f;
                if (a.lam3 != 0.f) {
                    float sa = 0.f;
#pragma unroll
                    for (int s = 0; s < KZ; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) sa = fmaf(lam[s], lam[s], sa); }
                    sa = group_sum(sa);
                    const float inv = sa > 0.f ? a.lam3 * rsqrtf(sa) : 0.f;
#pragma unroll
                    for (int s = 0; s < KZ; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) lam[s] = fmaf(inv, lam[s], lam[s]); }
                }
            } else {
#pragma unroll
                for (int s = 0; s < KZ; ++s) lam[s] = a.lam[(tile * 64 + lane) * KZ + s];
            }
        }
        __syncthreads();                 // the previous super-tile's readers of the LDS images are done
        // Zbar_j of this wave's sample tile: [NS][DT] f32x4 per lane - LDS, or (NC = 4: no room) the kernel's global scratch, L2-resident
        f32x4* zbt = ZL ? zbL + ((owner ? wave : 0) * NS * DT) * 64 + lane
                        : reinterpret_cast<f32x4*>(a.zb) + ((tile * NS * DT) * 64 + lane);
        f32x4 aS[A], aR = {0.f, 0.f, 0.f, 0.f};
        // What a stage needs from HBM at its very start - h_1 and delta_L of the stage (stage store), the checkpoint rows of the dense
        // phase - is requested one stage AHEAD, under the previous stage's last H x H product: with one wave per SIMD nothing else
        // hides a round trip to HBM, and the two D-sized products these loads used to sit in front of are a tenth of its length.
        U hN, dN;
        f32x4 krN[NS - 1][DT], kiN[DT], giN[DT];
        auto load_rows = [&](int is) {
            const long long rowb = (long long)a.step * ns * ntp + tile, rstride = ntp * 64 * (long long)ckzr;
            const float* kbase = a.ckpt_k + (rowb * 64 + lane) * ckzr;
            const float* gbase = (a.lam2 != 0.f ? a.ckpt_g : a.ckpt_k) + (rowb * 64 + lane) * ckzr;
#pragma unroll
            for (int j = 0; j < NS - 1; ++j) {
                const int jj = j < ns ? j : ns - 1;
#pragma unroll
                for (int q = 0; q < DT; ++q) krN[j][q] = *reinterpret_cast<const f32x4*>(kbase + jj * rstride + 4 * q);
            }
#pragma unroll
            for (int q = 0; q < DT; ++q) {
                kiN[q] = *reinterpret_cast<const f32x4*>(kbase + is * rstride + 4 * q);
                giN[q] = *reinterpret_cast<const f32x4*>(gbase + is * rstride + 4 * q);
            }
        };
        auto stage_off = [&](int is) { return (unsigned)(((long long)is * ntp + st * NC) * HTs * 1024); };
        if constexpr (PF) {
            load_units(q3.fh[0], stage_off(ns - 1), hN);
            load_units(q3.fd[L - 1], stage_off(ns - 1), dN);
        }
        if (owner) load_rows(ns - 1);   // (the rows: every instance)

#pragma clang loop unroll(disable)
        for (int i = ns - 1; i >= 0; --i) {
            // (the image's buffer resource is rebuilt per stage from the kernel argument made scalar by hand: see cnf_coop_dgrad.hip)
            const unsigned long long pimg = (unsigned long long)a.packed;
            const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pimg), phi = __builtin_amdgcn_readfirstlane((unsigned)(pimg >> 32));
            const DRs R{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)phi << 32) | plo), 0, 0x7fffffff, 0x00020000), R0.lane16};
            const float bi = a.T.b[i];
            const float cl = valid ? dt * bi : 0.f;      // cotangent of ldot (dL/d dlogp = +1 per column); zero for padding columns
            const float tt = tn + a.T.c[i] * dt;
            // byte offset of this super-tile's column-tile pair of stage i in every [ns][ntp][tiles] array
            const unsigned soH = stage_off(i);
#ifdef G3_TRACE
            unsigned long long tr[16];
#endif
            G3_T(0);
            if constexpr (!PF) load_units(q3.fh[0], soH, hN);
            U hcur = hN;                            // h_1 of this stage (PF: requested a stage ago)
            if (owner) {
                // ---- dense phase: stage state, kbar, gbar (cnf_coop_dgrad.hip's) on rows requested a stage ago ----
                float zs[KZ], kbar[KZ], gbar[KZ];
                f32x4 kr[NS - 1][DT], ki[DT], gi[DT], zr[NS - 1][DT];
#pragma unroll
                for (int j = 0; j < NS - 1; ++j)
#pragma unroll
                    for (int q = 0; q < DT; ++q) { kr[j][q] = krN[j][q]; zr[j][q] = zbt[((j + 1) * DT + q) * 64]; }
#pragma unroll
                for (int q = 0; q < DT; ++q) { ki[q] = kiN[q]; gi[q] = giN[q]; }
#pragma unroll
                for (int s = 0; s < KZ; ++s) {
                    float acc = 0.f, kb = bi * lam[s];
#pragma unroll
                    for (int j = 0; j < NS - 1; ++j) {
                        acc = fmaf(a.T.a[i][j], kr[j][s >> 2][s & 3], acc);                         // a[i][j] = 0 for j >= i
                        kb = fmaf(a.T.a[j + 1][i], (j + 1 > i && j + 1 < ns) ? zr[j][s >> 2][s & 3] : 0.f, kb);    // Zbar_j exists for i < j < ns only
                    }
                    zs[s] = fmaf(dt, acc, zn[s]);
                    kbar[s] = valid ? dt * kb : 0.f;
                    gbar[s] = -cl * eps[s];
                }
                // gbar = cotangent of g = eps^T J: -c_l eps (+ c_n g / |g|);  kbar += c_E zdot / |zdot|  (src/core/icnf.jl:184-251)
                if (a.lam1 != 0.f) {
                    float e2 = 0.f;
#pragma unroll
                    for (int s = 0; s < KZ; ++s) e2 = fmaf(ki[s >> 2][s & 3], ki[s >> 2][s & 3], e2);
                    e2 = group_sum(e2);
                    const float inv = e2 > 0.f ? cl * a.lam1 * rsqrtf(e2) : 0.f;
#pragma unroll
                    for (int s = 0; s < KZ; ++s) kbar[s] = fmaf(inv, ki[s >> 2][s & 3], kbar[s]);
                }
                if (a.lam2 != 0.f) {
                    float n2 = 0.f;
#pragma unroll
                    for (int s = 0; s < KZ; ++s) n2 = fmaf(gi[s >> 2][s & 3], gi[s >> 2][s & 3], n2);
                    n2 = group_sum(n2);
                    const float inv = n2 > 0.f ? cl * a.lam2 * rsqrtf(n2) : 0.f;
#pragma unroll
                    for (int s = 0; s < KZ; ++s) gbar[s] = fmaf(inv, gi[s >> 2][s & 3], gbar[s]);
                }
                publish_dense(gbuf, wave, gbar);
                publish_dense(kbuf, wave, kbar);
                // the D-row operands of Wbar_1 (= delta_1 gbar^T + sbar_1 [z; t]^T) and Wbar_N (= eps cbar^T + kbar h_L^T) as tiles
                {
                    const long long ctile = (long long)i * ntp + tile;
                    f32x4* gbp = reinterpret_cast<f32x4*>(q3.gb) + (ctile * DTZ) * 64 + lane;
                    f32x4* ztp = reinterpret_cast<f32x4*>(q3.zt) + (ctile * DTZ) * 64 + lane;
                    const int DTs = q3.DTs;              // tiles of the D-row arrays: the configuration's (<= the instance's DT)
                    f32x4* epp = reinterpret_cast<f32x4*>(q3.ep) + (ctile * DTs) * 64 + lane;
                    f32x4* kbp = reinterpret_cast<f32x4*>(q3.kb) + (ctile * DTs) * 64 + lane;
                    const int kt = D >> 4, et = (D & 15) >> 2, gt = D & 3;   // where the time row (feature D) sits
                    for (int kg = 0; kg < DTZ; ++kg) {
                        f32x4 zv = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k2 = 0; k2 < DT; ++k2)
                            if (k2 == kg) {
                                zv = f32x4{zs[4 * k2], zs[4 * k2 + 1], zs[4 * k2 + 2], zs[4 * k2 + 3]};
                                gv = f32x4{gbar[4 * k2], gbar[4 * k2 + 1], gbar[4 * k2 + 2], gbar[4 * k2 + 3]};
                            }
                        if (!a.autonomous && kg == kt && g == gt) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (e == et) zv[e] = tt;
                        }
                        if (!valid) zv = f32x4{0.f, 0.f, 0.f, 0.f};
                        gbp[kg * 64] = gv;
                        ztp[kg * 64] = zv;
                    }
#pragma unroll
                    for (int kg = 0; kg < DT; ++kg)
                        if (kg < DTs) {
                            epp[kg * 64] = f32x4{eps[4 * kg], eps[4 * kg + 1], eps[4 * kg + 2], eps[4 * kg + 3]};
                            kbp[kg * 64] = f32x4{kbar[4 * kg], kbar[4 * kg + 1], kbar[4 * kg + 2], kbar[4 * kg + 3]};
                        }
                }
            }
            U acc;
            U hP[L - 1], dbP[L - 1];               // h_l and dbar_l of the lower layers, parked until the way down
            zero_u(acc);
            g3_load_a<A, LO>(R, TZ, F1Z, 0, aS, aR);
            G3_T(1);
            G3_SYNC();                                                                     // gbar / kbar published
            G3_T(2);
            // Every global access of the stage is PLACED (s_memtime trace of the first build, profiles/r6/r6i_sweep_phase_trace.txt: the
            // memory counter retires in order, so eight tile stores or loads issued in FRONT of a product made its second k-group wait
            // for their acknowledgements / round trips to HBM - the D-sized product at the top took 7.8 k cycles for 2.1 k of MFMAs,
            // every H x H product 23 - 24 k for 16.4 k at cfg4): a tile set is LOADED behind the last fragment request of the product
            // before the one whose elementwise phase needs it, and STORED behind the product that follows the phase that made it.
            // ================= up 1: dbar_1 = W_1[:,0:D] gbar =================
            U hnext;
            g3_gemm<A, LO, NC>(R, TZ, F1Z, G.KGZ, G.remZ, v0, gbuf, lane, aS, aR, acc, [&]() { if constexpr (L > 1) load_units(q3.fh[1], soH, hnext); });
            G3_T(3);
            g3_load_a<A, LO>(R, TH, FH, 0, aS, aR);
            U hlast, dblast;                        // h_L, dbar_L: used at the top, in registers
            U vbl;                                   // vbar of the layer just finished, until the product behind it has been issued
#pragma unroll
            for (int l = 0; l < L; ++l) {
                up_ew(acc, hcur, vbl);                   // acc <- dbar_{l+1} (1-based), vbl <- vbar_{l+1}
                if (l + 1 < L) {
                    publish((l & 1) ? X1 : X0, vbl);
                    park_u(hcur, hP[l]);
                    park_u(acc, dbP[l]);
                    hcur = hnext;
                    zero_u(acc);
                    if (l == 0) G3_T(4);
                    G3_SYNC();
                    if (l == 0) G3_T(5);
                    // ================= up l + 2: dbar = W_{l+2} vbar_{l+1} =================
                    g3_gemm<A, LO, NC>(R, TH, FH + (unsigned)l * IMGH, G.KGH, G.remH, v0, (l & 1) ? X1 : X0, lane, aS, aR, acc,
                                       [&]() {
                                           if (l + 2 < L) load_units(q3.fh[l + 2 < L ? l + 2 : 0], soH, hnext);
                                           else if (!PF) load_units(q3.fd[L - 1], soH, dN);      // delta_L for the top
                                       });
                    if (l == 0) G3_T(6);
                    if (l + 2 < L) g3_load_a<A, LO>(R, TH, FH + (unsigned)(l + 1) * IMGH, 0, aS, aR);
                    else g3_load_a<A, LO>(R, TZ, BN, 0, aS, aR);
                    store_units(q3.sv[l], soH, vbl);
                } else {
                    hlast = hcur;
                    dblast = acc;
                }
            }
            // ================= the top: hbar_L = W_N^T kbar =================
            G3_T(7);
            U dlt, dln;
            zero_u(acc);
            g3_gemm<A, LO, NC>(R, TZ, BN, G.KGZ, G.remZ, v0, kbuf, lane, aS, aR, acc, [&]() { load_units(q3.fd[L - 2], soH, dln); });
            G3_T(8);
            g3_load_a<A, LO>(R, TH, BH + (unsigned)(L - 2) * IMGH, 0, aS, aR);
            store_units(q3.sv[L - 1], soH, vbl);
            down_ew(acc, hlast, dN, dblast);             // acc <- sbar_L  (dN: delta_L of this stage)
            // buffers: vbar_1 -> X0, (vbar_2 -> X1,) sbar_L -> the buffer the last up product did not read, alternating downwards
            constexpr int topbuf = (L - 1) & 1;          // L = 2: X1; L = 3: X0
            publish(topbuf ? X1 : X0, acc);
            U sprev = acc;                               // sbar_{l+1}: stored behind the product that reads it
            // the fragments of the Zbar product: ALL requested behind the last H x H product where they are few (<= 12 f32x4), else the first
            // k-group's there and the others one k-group ahead inside the product
            constexpr int NZ = A + (LO ? 1 : 0);
            constexpr bool ZALL = NZ * DT <= 12;
            f32x4 fz[ZALL ? NZ : 2][DT];
#pragma unroll
            for (int l = L - 1; l >= 1; --l) {           // hbar_l = W_{l+1}^T sbar_{l+1}  (1-based l)
                const int rb = ((L - 1 - l) & 1) ^ topbuf;   // the buffer sbar_{l+1} was published in
                dlt = dln;
                zero_u(acc);
                if (l == 1) G3_T(9);
                G3_SYNC();
                if (l == 1) G3_T(10);
                g3_gemm<A, LO, NC>(R, TH, BH + (unsigned)(l - 1) * IMGH, G.KGH, G.remH, v0, rb ? X1 : X0, lane, aS, aR, acc,
                                   [&]() { if (l > 1) load_units(q3.fd[l > 1 ? l - 2 : 0], soH, dln); });
                if (l == 1) G3_T(11);
                if (l > 1) g3_load_a<A, LO>(R, TH, BH + (unsigned)(l - 2) * IMGH, 0, aS, aR);
                else {
#pragma unroll
                    for (int m = 0; m < (ZALL ? NZ : 1); ++m)
#pragma unroll
                        for (int dm = 0; dm < DT; ++dm) fz[m][dm] = dloadv(R, vd[dm], B1 + (unsigned)(m < A ? mtS0 + m : tR) * 1024u);
                }
                store_units(q3.ss[l], soH, sprev);
                if (l == 1) {                            // the next stage's (i - 1) first needs: they travel under the rest of this stage
                    const int inx = i > 0 ? i - 1 : 0;
                    if constexpr (PF) {
                        load_units(q3.fh[0], stage_off(inx), hN);
                        load_units(q3.fd[L - 1], stage_off(inx), dN);
                    }
                    if (owner) load_rows(inx);
                }
                U hh, db;
                unpark_u(hP[l - 1], hh);
                unpark_u(dbP[l - 1], db);
                down_ew(acc, hh, dlt, db);               // acc <- sbar_l
                if (l > 1) { publish(rb ? X0 : X1, acc); sprev = acc; }
            }
            // ================= Zbar_i = W_1[:,0:D]^T sbar_1: partial tiles over this wave's own k-groups, from registers =================
            G3_T(12);
            f32x4 part[DT][NC];
#pragma unroll
            for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                for (int c = 0; c < NC; ++c) part[dm][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < NZ; ++m) {
                f32x4(&cur)[DT] = fz[ZALL ? m : (m & 1)];
                if (!ZALL && m + 1 < NZ) {
#pragma unroll
                    for (int dm = 0; dm < DT; ++dm) fz[(m + 1) & 1][dm] = dloadv(R, vd[dm], B1 + (unsigned)(m + 1 < A ? mtS0 + m + 1 : tR) * 1024u);
                }
                if (m < A) {
                    const int js = (mtS0 + m == G.KGH - 1) ? G.remH : 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < js) {
#pragma unroll
                            for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                                for (int c = 0; c < NC; ++c) part[dm][c] = mfma4(cur[dm][j], acc.S[m][c][j], part[dm][c]);
                        }
                } else if (v0) {
                    // (a left-over tile may be the last k-group: its k-steps beyond `rem` multiply zero columns of the image)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                            for (int c = 0; c < NC; ++c) part[dm][c] = mfma4(cur[dm][j], acc.R[c][j], part[dm][c]);
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
                printf("w%d: dense %d B0 %d up1 %d ew1 %d B1 %d up2 %d ew2(+up3) %d top %d ewtop %d B %d down %d ewdown %d zbar %d B %d red %d | stage %d\n", wave, G3_D(1), G3_D(2), G3_D(3),
                       G3_D(4), G3_D(5), G3_D(6), G3_D(7), G3_D(8), G3_D(9), G3_D(10), G3_D(11), G3_D(12), G3_D(13), G3_D(14), G3_D(15), (int)(tr[15] - tr[0]));
            }
#endif
            // (the next stage's first LDS writes - gbar / kbar images - touch neither exchange buffer; its first publish into X0 comes
            // behind its first barrier, which the owners reach after this sum)
        }
        if (owner) {
#pragma unroll
            for (int s = 0; s < KZ; ++s) {
                float acc = lam[s];
                for (int j = 0; j < ns; ++j) acc += zbt[(j * DT + (s >> 2)) * 64][s & 3];
                lam[s] = acc;
                a.lam[(tile * 64 + lane) * KZ + s] = acc;
            }
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

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
template <int A, bool LO, int L, int KZ, int ACT, int NS, int NC, bool PF>
static hipError_t launch_g3(const G3Args& a, int lds, int nblocks, hipStream_t st) {
    auto kern = coop_grad3_step_kernel<A, LO, L, KZ, ACT, NS, NC, PF>;
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

struct G3Inst {
    int A, LO, L, KZ, ACT, NC;
    hipError_t (*fn[2])(const G3Args&, int, int, hipStream_t);   // [0] RK4 (4 stages), [1] Tsit5 (6 stages)
};
// three hidden layers: nothing requested ahead (the registers hold h_l and dbar_l of two layers); two: the next stage's first needs
// requested a stage ahead.  Every instance runs 32-sample super-tiles: the 64-sample form (NC = 4: every wave an owner, a stage's
// fixed costs per twice the MFMAs) is written, but its two-hidden-layer instances either crash this compiler's AGPR-copy rewrite
// pass (A = 3; A = 2 with 8 state registers) or, built without -amdgpu-mfma-vgpr-form, spill 68 registers to scratch.
#define G3_INSTX(A, LO, L, KZ, ACT, NC, PF) G3Inst { A, LO, L, KZ, ACT, NC, { &launch_g3<A, LO, L, KZ, ACT, 4, NC, PF>, &launch_g3<A, LO, L, KZ, ACT, 6, NC, PF> } }
#define G3_INST(A, LO, L, KZ, ACT) G3_INSTX(A, LO, L, KZ, ACT, 2, (L == 2))
static const G3Inst kG3[] = {
    G3_INST(4, false, 3, 8, CNF_ACT_TANH_PRESCALED),   // cfg4: 3 x 256, D <= 32
    G3_INST(2, false, 3, 8, CNF_ACT_TANH_PRESCALED),   // 3 x 128
    // the flows whose forward solve runs on the dealt kernel (cnf_coop_d.hip: 8 .. 15 hidden tiles): the reference's default
    // architecture at nvariables = 16 .. 29 (two softplus layers), tanh nets of those widths
    G3_INST(2, true, 2, 8, CNF_ACT_SOFTPLUS), G3_INST(2, true, 2, 12, CNF_ACT_SOFTPLUS), G3_INST(3, true, 2, 12, CNF_ACT_SOFTPLUS), G3_INST(3, true, 2, 16, CNF_ACT_SOFTPLUS),
    G3_INST(3, true, 2, 8, CNF_ACT_SOFTPLUS),
    G3_INST(2, true, 2, 8, CNF_ACT_TANH_PRESCALED), G3_INST(2, true, 2, 12, CNF_ACT_TANH_PRESCALED), G3_INST(3, true, 2, 8, CNF_ACT_TANH_PRESCALED),
    G3_INST(3, true, 2, 12, CNF_ACT_TANH_PRESCALED), G3_INST(3, true, 2, 16, CNF_ACT_TANH_PRESCALED),
    G3_INST(2, true, 3, 8, CNF_ACT_TANH_PRESCALED), G3_INST(3, true, 3, 8, CNF_ACT_TANH_PRESCALED), G3_INST(2, true, 3, 12, CNF_ACT_TANH_PRESCALED),
    G3_INST(2, true, 3, 8, CNF_ACT_SOFTPLUS), G3_INST(2, true, 3, 12, CNF_ACT_SOFTPLUS),
};
static const G3Inst* g3_find(int HT_real, int L, int KZ, int ACT) {
    const int A = HT_real / 4, b = HT_real - 4 * A;
    const G3Inst* best = nullptr;
    for (const G3Inst& c : kG3) {
        const bool act_ok = c.ACT == ACT || (c.ACT == CNF_ACT_TANH_PRESCALED && ACT == CNF_ACT_TANH);
        if (c.A == A && (c.LO || b == 0) && c.L == L && c.KZ >= KZ && act_ok && (!best || c.KZ < best->KZ || (c.KZ == best->KZ && !c.LO && best->LO))) best = &c;
    }
    return best;
}

bool coop_grad3_supported(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int CR_lay) {
    if (CR_lay != 0) return false;
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    if (HT_real > HT_lay || KZ > ZR_lay) return false;
    const G3Inst* c = g3_find(HT_real, L, KZ, ACT);
    if (!c || c->KZ > ZR_lay) return false;   // the checkpoint rows are read KZ registers wide
    return coop_grad3_lds_bytes(HT_real, c->KZ / 4, c->NC) <= 160 * 1024;   // (sized for six stages)
}

hipError_t coop_grad3_step_launch(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, const CG3Args& a, int num_cus, hipStream_t st) {
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    const G3Inst* c = g3_find(HT_real, L, KZ, ACT);
    if (!c) return hipErrorNotSupported;
    G3Args ga{};
    ga.c = a;
    dimg_fill(ga.g, H, D, L, HT_lay, ZR_lay, 0, c->A, 0);
    if (a.ck_tiles) return hipErrorNotSupported;   // (this sweep reads lane-major checkpoint rows)
    const int lds = coop_grad3_lds_bytes(HT_real, c->KZ / 4, c->NC);
    if (lds > 160 * 1024) return hipErrorNotSupported;
    const long long nst = a.c.ntiles_pad / c->NC;
    const int nblocks = (int)(nst < num_cus ? nst : num_cus);
    return c->fn[a.c.T.ns <= 4 ? 0 : 1](ga, lds, nblocks, st);
}

}  // namespace cnf
