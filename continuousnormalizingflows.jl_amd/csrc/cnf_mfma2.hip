// cnf_mfma2.hip — the fixed-step solve of one-probe Hutchinson-VJP flows, hand-scheduled (round 5; gfx950).
//
// Same mathematics, operand image, data layout and arithmetic as mfma_solve_kernel<HT, L, ZR, 0, ACT, ENG_VJP, 1, ..> (cnf_mfma_kernel.h:
// read its header first; SURVEY.md section 8 A2 - A5, A9 - A10: inference_prob's u0 = [x; 0], augmented_f in TrainMode with the VJP
// trace estimate, base_sol with a fixed-step RK4 / Tsit5, inference_sol's epilogue - src/core/base_icnf.jl:134-172,247-296,
// src/core/icnf.jl:517-559) - every product multiplies the same fragments in the same order, every elementwise step is the same
// expression, so a solve is bit-identical to that kernel's.  What changes is the instruction order, laid out by hand the way
// cnf_grad2.hip's stage is (cnf_sched_dev.h):
//   * fragments are requested one k-group ahead of the MFMAs that use them, the first k-group of the NEXT product - across the
//     stage boundary too - behind the current product's last k-group, bias vectors with them;
//   * MFMA runs, activation phases and the Runge-Kutta update are fenced apart (f32 MFMAs hide no VALU work on gfx950 and every
//     MFMA -> VALU -> MFMA round trip costs ~9 issue cycles);
//   * the tableau entries of a stage come from an LDS table, not from scalar loads at a run-time index (those share the
//     out-of-order lgkmcnt counter with every LDS read that follows: each fragment wait becomes lgkmcnt(0));
//   * the checkpoint stores of a gradient's forward pass (z_n per step, zdot per stage) are issued behind the stage's last product.
// Whole solves only (nsteps >= 1), no conditions, no kfull rows: everything else stays on mfma_solve_kernel.
#include "cnf_sched_dev.h"

namespace cnf {

template <int HT, int L, int ZR, int ACT, int NTHREADS>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(NTHREADS / 256, NTHREADS / 256)))
mfma_solve2_kernel(KArgs a) {
    constexpr MfmaLayout LAY(HT, L, ZR, 0, true, 0);
    constexpr int DT = (ZR + 3) / 4, NH = L - 1, ZJ = ZR < 4 ? ZR : 4;
    constexpr int TAB = (LAY.lds_total + 3) / 4 * 4;   // [6 stages][8]: c, b, acol[0..4]
    static_assert(DT == 1, "D <= 16");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stage_image<NTHREADS>(a.packed, smem, LAY.lds_total / 4);
    if (threadIdx.x < 48) {
        const int st = threadIdx.x >> 3, k = threadIdx.x & 7;
        smem[TAB + threadIdx.x] = k == 0 ? a.T.c[st] : k == 1 ? a.T.b[st] : k < 7 ? a.acol[st][k - 2] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, n = lane & 15;
    const int wave = threadIdx.x >> 6;
    constexpr int WPB = NTHREADS / 64;
    const long long ntiles = (a.B + 15) / 16;
    const int D = a.D, S = D + 3;
    const bool reg_z = a.reg_z, reg_j = a.reg_j, autonomous = a.autonomous;
    const long long total_waves = (long long)gridDim.x * WPB;
    const float dt = a.dt;
    const int ns = a.T.ns, nsteps = a.nsteps;

    for (long long tile = (long long)blockIdx.x + (long long)gridDim.x * wave; tile < ntiles; tile += total_waves) {
        const long long smp = tile * 16 + n;
        const bool valid = smp < a.B;
        const long long sc = valid ? smp : a.B - 1;   // clamp loads, mask stores
        float z[ZR], eps[ZR];
        float lacc = 0.f, eacc = 0.f, nacc = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            if (a.x) z[s] = f < a.nvars ? a.x[sc * a.nvars + f] : 0.f;   // u0 = [x; 0]
            else z[s] = f < D ? a.u0[sc * S + f] : 0.f;
            eps[s] = (f < D && a.eps) ? a.eps[sc * D + f] : 0.f;
        }
        if (!a.x) { lacc = a.u0[sc * S + D]; eacc = a.u0[sc * S + D + 1]; nacc = a.u0[sc * S + D + 2]; }
        // solve-invariant products: c = W_N^T eps, q = W_1[:,0:D] eps (the forward image carries the tanh pre-scale)
        f32x4 pre_c[HT], pre_q[HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) pre_c[mt] = pre_q[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps}, pre_c);
        gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{eps}, pre_q);
        if constexpr (ACT == CNF_ACT_TANH_PRESCALED) {
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) pre_q[mt] *= (1.f / kTanhPrescale);
        }
        f32x4 nf[HT], nz[1];   // first fragments of the next product (across stages)
        afrag<HT, ZJ>(smem + LAY.f1z, lane, LAY.KGZ, 0, nf);
        // layer 1's bias and time column are the same in every stage: kept (32 registers) instead of read back per stage in front
        // of the first product; the tableau entries of a stage are requested one stage ahead
        // (one wave per SIMD only: with two, the partner wave covers those reads and the registers cost more than they save)
        constexpr bool KEEP_B1 = NTHREADS == 256;
        f32x4 b1v[HT], w1tv[HT];
        if constexpr (KEEP_B1) {
            load_cvec<HT>(smem + LAY.v_b1, g, b1v);
            load_cvec<HT>(smem + LAY.v_w1t, g, w1tv);
        }
        f32x4 tq0 = reinterpret_cast<const f32x4*>(smem + TAB)[0], tq1 = reinterpret_cast<const f32x4*>(smem + TAB)[1];

        float P[5][ZR], zsum[ZR], lsum, esum, nsum;
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
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                const float* sm = smem + opaque;
                auto IMG_F = [&](int l) { return sm + LAY.fh + l * MfmaLayout::imgA(HT, HT); };   // W_{l+2}
                auto IMG_B = [&](int l) { return sm + LAY.bh + l * MfmaLayout::imgA(HT, HT); };   // W_{l+2}^T
                const float* const no_img = nullptr;
                // the stage's tableau entries (LDS table: two uniform 16-byte reads, requested a stage ahead)
                const f32x4 t0 = tq0, t1 = tq1;
                {
                    const int sn = st + 1 < ns ? st + 1 : 0;
                    const f32x4* tb = reinterpret_cast<const f32x4*>(sm + TAB + sn * 8);
                    tq0 = tb[0]; tq1 = tb[1];
                }
                const float cst = t0[0], bst = t0[1];
                const float acol[5] = {t0[2], t0[3], t1[0], t1[1], t1[2]};
                float zs[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) zs[s] = fmaf(dt, P[0][s], z[s]);
                const float t = tn + cst * dt;
                // ---- layer 1: a = W1z z + w1t t + b1 ----
                f32x4 acc[HT], h[HT], d[L][HT];
                // C vectors: one lane base a stage, at the vectors' own region (the image is larger than a 16-bit immediate reaches)
                const float* smg = sm + 4 * g + LAY.v_b1;
                asm volatile("" : "+v"(smg));
                if constexpr (KEEP_B1) {
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) acc[mt] = autonomous ? b1v[mt] : tile_fma(w1tv[mt], t, b1v[mt]);
                } else {
                    load_cvec_g<HT>(smg, 0, acc);
                    if (!autonomous) {
                        f32x4 wt[HT];
                        load_cvec_g<HT>(smg, LAY.v_w1t - LAY.v_b1, wt);
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) acc[mt] = tile_fma(wt[mt], t, acc[mt]);
                    }
                }
                G2_FENCE();
                gemm_pf<HT, ZR, HT>(sm + LAY.f1z, lane, RegIn<ZR>{zs}, nf, acc, no_img, 0, nf);
                static_for<0, L>([&](auto lc) {
                    constexpr int l = decltype(lc)::value;
                    f32x4 accn[HT], accz[DT];
                    if constexpr (l + 1 < L) { afrag<HT>(IMG_F(l), lane, HT, 0, nf); load_cvec_g<HT>(smg, LAY.v_bh - LAY.v_b1 + l * MfmaLayout::vecC(HT), accn); }
                    else { afrag<DT>(sm + LAY.fN, lane, HT, 0, nz); load_cvec_g<DT>(smg, LAY.v_bN - LAY.v_b1, accz); }
                    G2_FENCE();
                    if constexpr (HT == 4 && ACT == CNF_ACT_TANH_PRESCALED) {
                        tanh_tiles4<true, true>(acc, h, d[l]);
                    } else {
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) act_tile<ACT>(acc[mt], h[mt], d[l][mt]);
                    }
                    G2_FENCE();
                    if constexpr (l + 1 < L) {
                        gemm_pf<HT, 4 * HT, HT>(IMG_F(l), lane, TileIn<HT>{h}, nf, accn, no_img, 0, nf);
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) acc[mt] = accn[mt];
                    } else {
                        // last layer (identity): zdot; the first pullback fragments behind it
                        gemm_pf<DT, 4 * HT, HT>(sm + LAY.fN, lane, TileIn<HT>{h}, nz, accz, IMG_B(NH - 1), HT, nf);
#pragma unroll
                        for (int mt = 0; mt < DT; ++mt) acc[mt] = accz[mt];
                    }
                });
                float zd[ZR], ld = 0.f, ed = 0.f, nd = 0.f;
#pragma unroll
                for (int s = 0; s < ZR; ++s) zd[s] = acc[s >> 2][s & 3];
                if (reg_z) {   // Edot = |zdot|_2   (src/core/icnf.jl:184-199)
                    float e2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) e2 = fmaf(zd[s], zd[s], e2);
                    ed = sqrtf(group_sum(e2));
                }
                // ---- pullback: delta_L = c .* act'_L, delta_l = (W_{l+1}^T delta_{l+1}) .* act'_l ----
                f32x4 dl[HT];
                tiles_mul<HT>(pre_c, d[L - 1], dl);
                static_for<0, NH>([&](auto lc) {
                    constexpr int l = L - 1 - decltype(lc)::value;   // L-1 .. 1
                    f32x4 u[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) u[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    G2_FENCE();
                    // (the last product requests the NEXT stage's layer-1 fragments)
                    gemm_pf<HT, 4 * HT, HT, (l > 1 ? 4 : ZR)>(IMG_B(l - 1), lane, TileIn<HT>{dl}, nf, u, l > 1 ? IMG_B(l > 1 ? l - 2 : 0) : sm + LAY.f1z, l > 1 ? HT : LAY.KGZ, nf);
                    G2_FENCE();
                    tiles_mul<HT>(u, d[l - 1], dl);
                });
                if (!reg_j) {
                    // <eps^T J, eps> = <delta_1, W_1[:,0:D] eps>: a dot with the hoisted q replaces the last product
                    const float qd = tiles_dot<HT>(dl, pre_q);
                    ld -= group_sum(qd);
                } else {
                    f32x4 gacc[DT];
#pragma unroll
                    for (int dt_ = 0; dt_ < DT; ++dt_) gacc[dt_] = f32x4{0.f, 0.f, 0.f, 0.f};
                    gemm_tiles<DT, 4 * HT>(sm + LAY.b1, lane, TileIn<HT>{dl}, gacc);   // W_1[:,0:D]^T delta_1
                    float dot = 0.f, n2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) {
                        const float gv = gacc[s >> 2][s & 3];
                        dot = fmaf(gv, eps[s], dot);
                        n2 = fmaf(gv, gv, n2);
                    }
                    ld -= group_sum(dot);
                    nd += sqrtf(group_sum(n2));   // ndot = |eps^T J|_2 (icnf.jl:229-245)
                }
                if (a.ckpt_k) {
#pragma unroll
                    for (int s = 0; s < ZR; ++s)
                        a.ckpt_k[((((long long)step * ns + st) * ntiles + tile) * 64 + lane) * ZR + s] = zd[s];
                }
                // ---- fold the stage derivative into the step update and into the partial sums of the stages still to come ----
                lsum = fmaf(bst, ld, lsum); esum = fmaf(bst, ed, esum); nsum = fmaf(bst, nd, nsum);
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    zsum[s] = fmaf(bst, zd[s], zsum[s]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) P[i][s] = fmaf(acol[i], zd[s], P[i + 1][s]);
                    P[4][s] = acol[4] * zd[s];
                }
            }
            lacc = fmaf(dt, lsum, lacc); eacc = fmaf(dt, esum, eacc); nacc = fmaf(dt, nsum, nacc);
#pragma unroll
            for (int s = 0; s < ZR; ++s) z[s] = fmaf(dt, zsum[s], z[s]);
        }
        if (a.ckpt) {
#pragma unroll
            for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)nsteps * ntiles + tile) * 64 + lane) * ZR + s] = z[s];
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
    }
}

// ---------------------------------------------------------------------------------------
// The same solve on TWO waves per 16-sample tile, for batches that leave most of the chip idle (cfg1 - BASELINE's CPU-runnable
// configuration - is 64 tiles on 1024 SIMDs; a tile's solve is one dependent chain of 240 dynamics calls, 2 570 cycles each on one
// wave: 0.257 ms whatever the batch).  The pullback (eps^T J eps, |eps^T J|) of a stage needs the forward pass's act' tiles and
// nothing else, and the forward pass of the next stage needs nothing of the pullback (the state does not depend on the trace
// estimates): wave 0 of the workgroup runs the forward chain - products, activations, zdot, |zdot|, the Runge-Kutta update of z -
// and leaves act' of every layer in one of two LDS slots; wave 1, one stage behind on another SIMD, multiplies the transposed images
// and folds the trace terms.  One barrier per stage (the forward wave writes slot i & 1 in front of barrier i, the pullback wave
// reads it behind; slot i & 1 is rewritten behind barrier i + 1, which the pullback wave reaches with stage i's reads done).
// A stage costs max(forward, pullback) + a barrier instead of their sum.  Same fragments, same products in the same order, same
// elementwise expressions as mfma_solve2_kernel / mfma_solve_kernel: bit-identical (test-enforced).
// (the pair's barrier waits for the wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. the checkpoint stores of a
// gradient's forward pass, once per stage)
#define S2P_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int HT, int L, int ZR, int ACT>
__global__ void __launch_bounds__(128)
mfma_solve2p_kernel(KArgs a) {
    constexpr MfmaLayout LAY(HT, L, ZR, 0, true, 0);
    constexpr int DT = (ZR + 3) / 4, NH = L - 1, ZJ = ZR < 4 ? ZR : 4;
    constexpr int TAB = (LAY.lds_total + 3) / 4 * 4;   // [6 stages][8]: c, b, acol[0..4]
    constexpr int XCH = TAB + 48;                      // [2 slots][L][HT][64 lanes] x 4 floats: act' of the stage
    constexpr int FIN = XCH + 2 * L * HT * 256;        // [64 lanes][2]: the pullback wave's logp and n accumulators at the end
    static_assert(DT == 1, "D <= 16");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stage_image<128>(a.packed, smem, LAY.lds_total / 4);
    if (threadIdx.x < 48) {
        const int st = threadIdx.x >> 3, k = threadIdx.x & 7;
        smem[TAB + threadIdx.x] = k == 0 ? a.T.c[st] : k == 1 ? a.T.b[st] : k < 7 ? a.acol[st][k - 2] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long ntiles = (a.B + 15) / 16;
    const long long tile = blockIdx.x;                 // one tile per workgroup (grid = tiles)
    const int D = a.D, S = D + 3;
    const bool reg_z = a.reg_z, reg_j = a.reg_j, autonomous = a.autonomous;
    const float dt = a.dt;
    const int ns = a.T.ns, nsteps = a.nsteps;
    const long long smp = tile * 16 + n;
    const bool valid = smp < a.B;
    const long long sc = valid ? smp : a.B - 1;        // clamp loads, mask stores
    f32x4* const xch = reinterpret_cast<f32x4*>(smem + XCH) + lane;

    if (wave == 0) {
        // ---------------- forward chain ----------------
        float z[ZR];
        float eacc = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            if (a.x) z[s] = f < a.nvars ? a.x[sc * a.nvars + f] : 0.f;   // u0 = [x; 0]
            else z[s] = f < D ? a.u0[sc * S + f] : 0.f;
        }
        if (!a.x) eacc = a.u0[sc * S + D + 1];
        f32x4 nf[HT], nz[1];
        afrag<HT, ZJ>(smem + LAY.f1z, lane, LAY.KGZ, 0, nf);
        f32x4 b1v[HT], w1tv[HT];
        load_cvec<HT>(smem + LAY.v_b1, g, b1v);
        load_cvec<HT>(smem + LAY.v_w1t, g, w1tv);
        f32x4 tq0 = reinterpret_cast<const f32x4*>(smem + TAB)[0], tq1 = reinterpret_cast<const f32x4*>(smem + TAB)[1];
        float P[5][ZR], zsum[ZR], esum;
        int it = 0;
#pragma clang loop unroll(disable)
        for (int step = 0; step < nsteps; ++step) {
            const float tn = a.t0 + (float)step * dt;
            if (a.ckpt) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)step * ntiles + tile) * 64 + lane) * ZR + s] = z[s];
            }
            esum = 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                zsum[s] = 0.f;
#pragma unroll
                for (int i = 0; i < 5; ++i) P[i][s] = 0.f;
            }
#pragma clang loop unroll(disable)
            for (int st = 0; st < ns; ++st, ++it) {
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                const float* sm = smem + opaque;
                auto IMG_F = [&](int l) { return sm + LAY.fh + l * MfmaLayout::imgA(HT, HT); };   // W_{l+2}
                const float* const no_img = nullptr;
                const f32x4 t0 = tq0, t1 = tq1;
                {
                    const int sn = st + 1 < ns ? st + 1 : 0;
                    const f32x4* tb = reinterpret_cast<const f32x4*>(sm + TAB + sn * 8);
                    tq0 = tb[0]; tq1 = tb[1];
                }
                const float cst = t0[0], bst = t0[1];
                const float acol[5] = {t0[2], t0[3], t1[0], t1[1], t1[2]};
                float zs[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) zs[s] = fmaf(dt, P[0][s], z[s]);
                const float t = tn + cst * dt;
                f32x4 acc[HT], h[HT], d[HT];
                const float* smg = sm + 4 * g + LAY.v_b1;
                asm volatile("" : "+v"(smg));
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) acc[mt] = autonomous ? b1v[mt] : tile_fma(w1tv[mt], t, b1v[mt]);
                G2_FENCE();
                gemm_pf<HT, ZR, HT>(sm + LAY.f1z, lane, RegIn<ZR>{zs}, nf, acc, no_img, 0, nf);
                f32x4* const slot = xch + (it & 1) * (L * HT * 64);
                static_for<0, L>([&](auto lc) {
                    constexpr int l = decltype(lc)::value;
                    f32x4 accn[HT], accz[DT];
                    if constexpr (l + 1 < L) { afrag<HT>(IMG_F(l), lane, HT, 0, nf); load_cvec_g<HT>(smg, LAY.v_bh - LAY.v_b1 + l * MfmaLayout::vecC(HT), accn); }
                    else { afrag<DT>(sm + LAY.fN, lane, HT, 0, nz); load_cvec_g<DT>(smg, LAY.v_bN - LAY.v_b1, accz); }
                    G2_FENCE();
                    if constexpr (HT == 4 && ACT == CNF_ACT_TANH_PRESCALED) {
                        tanh_tiles4<true, true>(acc, h, d);
                    } else {
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) act_tile<ACT>(acc[mt], h[mt], d[mt]);
                    }
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) slot[(l * HT + mt) * 64] = d[mt];     // act' of layer l + 1 for the pullback wave
                    G2_FENCE();
                    if constexpr (l + 1 < L) {
                        gemm_pf<HT, 4 * HT, HT>(IMG_F(l), lane, TileIn<HT>{h}, nf, accn, no_img, 0, nf);
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) acc[mt] = accn[mt];
                    } else {
                        // last layer (identity): zdot; the NEXT stage's layer-1 fragments behind it
                        gemm_pf<DT, 4 * HT, HT, ZR>(sm + LAY.fN, lane, TileIn<HT>{h}, nz, accz, sm + LAY.f1z, LAY.KGZ, nf);
#pragma unroll
                        for (int mt = 0; mt < DT; ++mt) acc[mt] = accz[mt];
                    }
                });
                S2P_BARRIER();                       // barrier `it`: the stage's act' tiles are in their slot
                float zd[ZR], ed = 0.f;
#pragma unroll
                for (int s = 0; s < ZR; ++s) zd[s] = acc[s >> 2][s & 3];
                if (reg_z) {   // Edot = |zdot|_2   (src/core/icnf.jl:184-199)
                    float e2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) e2 = fmaf(zd[s], zd[s], e2);
                    ed = sqrtf(group_sum(e2));
                }
                if (a.ckpt_k) {
#pragma unroll
                    for (int s = 0; s < ZR; ++s)
                        a.ckpt_k[((((long long)step * ns + st) * ntiles + tile) * 64 + lane) * ZR + s] = zd[s];
                }
                esum = fmaf(bst, ed, esum);
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    zsum[s] = fmaf(bst, zd[s], zsum[s]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) P[i][s] = fmaf(acol[i], zd[s], P[i + 1][s]);
                    P[4][s] = acol[4] * zd[s];
                }
            }
            eacc = fmaf(dt, esum, eacc);
#pragma unroll
            for (int s = 0; s < ZR; ++s) z[s] = fmaf(dt, zsum[s], z[s]);
        }
        if (a.ckpt) {
#pragma unroll
            for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)nsteps * ntiles + tile) * 64 + lane) * ZR + s] = z[s];
        }
        S2P_BARRIER();                               // the pullback wave's accumulators are in FIN
        const float lacc = smem[FIN + 2 * lane], nacc = smem[FIN + 2 * lane + 1];
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
    } else {
        // ---------------- pullback, one stage behind ----------------
        float eps[ZR];
        float lacc = 0.f, nacc = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            eps[s] = (f < D && a.eps) ? a.eps[sc * D + f] : 0.f;
        }
        if (!a.x) { lacc = a.u0[sc * S + D]; nacc = a.u0[sc * S + D + 2]; }
        // solve-invariant products: c = W_N^T eps, q = W_1[:,0:D] eps (the forward image carries the tanh pre-scale)
        f32x4 pre_c[HT], pre_q[HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) pre_c[mt] = pre_q[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps}, pre_c);
        gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{eps}, pre_q);
        if constexpr (ACT == CNF_ACT_TANH_PRESCALED) {
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) pre_q[mt] *= (1.f / kTanhPrescale);
        }
        const bool use_q = a.use_q != 0 && !reg_j;
        f32x4 nf[HT];
        if constexpr (NH > 0) afrag<HT>(smem + LAY.bh + (NH - 1) * MfmaLayout::imgA(HT, HT), lane, HT, 0, nf);
        float lsum, nsum;
        int it = 0;
#pragma clang loop unroll(disable)
        for (int step = 0; step < nsteps; ++step) {
            lsum = nsum = 0.f;
#pragma clang loop unroll(disable)
            for (int st = 0; st < ns; ++st, ++it) {
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                const float* sm = smem + opaque;
                auto IMG_B = [&](int l) { return sm + LAY.bh + l * MfmaLayout::imgA(HT, HT); };   // W_{l+2}^T
                const float bst = sm[TAB + st * 8 + 1];
                S2P_BARRIER();                       // barrier `it`: the forward wave has left the stage's act' tiles
                const f32x4* const slot = reinterpret_cast<const f32x4*>(sm + XCH) + lane + (it & 1) * (L * HT * 64);
                f32x4 d[L][HT];
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) d[l][mt] = slot[(l * HT + mt) * 64];
                float ld = 0.f, nd = 0.f;
                // ---- pullback: delta_L = c .* act'_L, delta_l = (W_{l+1}^T delta_{l+1}) .* act'_l ----
                f32x4 dl[HT];
                tiles_mul<HT>(pre_c, d[L - 1], dl);
                static_for<0, NH>([&](auto lc) {
                    constexpr int l = L - 1 - decltype(lc)::value;   // L-1 .. 1
                    f32x4 u[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) u[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    G2_FENCE();
                    // (the last product requests the NEXT stage's first pullback fragments)
                    gemm_pf<HT, 4 * HT, HT>(IMG_B(l - 1), lane, TileIn<HT>{dl}, nf, u, IMG_B(l > 1 ? l - 2 : NH - 1), HT, nf);
                    G2_FENCE();
                    tiles_mul<HT>(u, d[l - 1], dl);
                });
                if (use_q) {
                    // <eps^T J, eps> = <delta_1, W_1[:,0:D] eps>: a dot with the hoisted q replaces the last product
                    const float qd = tiles_dot<HT>(dl, pre_q);
                    ld -= group_sum(qd);
                } else {
                    f32x4 gacc[DT];
#pragma unroll
                    for (int dt_ = 0; dt_ < DT; ++dt_) gacc[dt_] = f32x4{0.f, 0.f, 0.f, 0.f};
                    gemm_tiles<DT, 4 * HT>(sm + LAY.b1, lane, TileIn<HT>{dl}, gacc);   // W_1[:,0:D]^T delta_1
                    float dot = 0.f, n2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) {
                        const float gv = gacc[s >> 2][s & 3];
                        dot = fmaf(gv, eps[s], dot);
                        n2 = fmaf(gv, gv, n2);
                    }
                    ld -= group_sum(dot);
                    if (reg_j) nd += sqrtf(group_sum(n2));   // ndot = |eps^T J|_2 (icnf.jl:229-245)
                }
                lsum = fmaf(bst, ld, lsum); nsum = fmaf(bst, nd, nsum);
            }
            lacc = fmaf(dt, lsum, lacc); nacc = fmaf(dt, nsum, nacc);
        }
        smem[FIN + 2 * lane] = lacc;
        smem[FIN + 2 * lane + 1] = nacc;
        S2P_BARRIER();
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
namespace {
typedef hipError_t (*Solve2Launch)(const KArgs&, int, int, hipStream_t);
template <int HT, int L, int ZR, int ACT, int NT>
hipError_t solve2_launch_inst(const KArgs& a, int lds, int nblocks, hipStream_t st) {
    auto kern = mfma_solve2_kernel<HT, L, ZR, ACT, NT>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (!once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), lds, st, a);
    return hipGetLastError();
}
struct Solve2Inst { int HT, L, ZR, ACT, NT; Solve2Launch fn; };
#define S2_INST(HT, L, ZR, ACT, NT) Solve2Inst { HT, L, ZR, ACT, NT, &solve2_launch_inst<HT, L, ZR, ACT, NT> }
const Solve2Inst kSolve2[] = {
    S2_INST(4, 3, 2, CNF_ACT_TANH_PRESCALED, 256),   // cfg2 / cfg2': D = 8, 3 x 64, one wave per SIMD
    S2_INST(4, 3, 2, CNF_ACT_TANH_PRESCALED, 512),   // ... two
    S2_INST(2, 2, 1, CNF_ACT_TANH_PRESCALED, 256),   // cfg1: D = 2, 2 x 32 (B = 1024: one tile per CU, latency-bound)
    S2_INST(2, 2, 1, CNF_ACT_TANH_PRESCALED, 512),
};
const Solve2Inst* s2_find(int HT, int L, int ZR, int ACT, int NT) {
    for (const Solve2Inst& s : kSolve2)
        if (s.HT == HT && s.L == L && s.ZR == ZR && s.ACT == ACT && (NT == 0 || s.NT == NT)) return &s;
    return nullptr;
}
}  // namespace

bool solve2_supported(int HT, int L, int ZR, int ACT) { return s2_find(HT, L, ZR, ACT, 0) != nullptr; }

// the two-waves-per-tile form: nets of two hidden tiles (a tile-split over four waves does not pay there)
namespace {
template <int HT, int L, int ZR, int ACT>
hipError_t solve2p_launch_inst(const KArgs& a, int lds, int nblocks, hipStream_t st) {
    auto kern = mfma_solve2p_kernel<HT, L, ZR, ACT>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (!once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(128), lds, st, a);
    return hipGetLastError();
}
struct Solve2pInst { int HT, L, ZR, ACT; Solve2Launch fn; };
const Solve2pInst kSolve2p[] = {
    {2, 2, 1, CNF_ACT_TANH_PRESCALED, &solve2p_launch_inst<2, 2, 1, CNF_ACT_TANH_PRESCALED>},   // cfg1: D = 2, 2 x 32
    {2, 2, 4, CNF_ACT_TANH_PRESCALED, &solve2p_launch_inst<2, 2, 4, CNF_ACT_TANH_PRESCALED>},   // the zero-padded instances' layout: D <= 16
    {2, 2, 4, CNF_ACT_SOFTPLUS, &solve2p_launch_inst<2, 2, 4, CNF_ACT_SOFTPLUS>},               // e.g. the default architecture at nvariables = 3
};
const Solve2pInst* s2p_find(int HT, int L, int ZR, int ACT) {
    for (const Solve2pInst& s : kSolve2p)
        if (s.HT == HT && s.L == L && s.ZR == ZR && s.ACT == ACT) return &s;
    return nullptr;
}
}  // namespace

// batches of at most this many tiles run two waves per tile (every tile on its own pair of SIMDs: 2 x 256 CUs); beyond it the
// per-wave form's throughput wins
bool solve2p_supported(int HT, int L, int ZR, int ACT, long long ntiles, int num_cus) {
    return s2p_find(HT, L, ZR, ACT) != nullptr && ntiles <= 2ll * num_cus;
}

hipError_t solve2p_launch(int HT, int L, int ZR, int ACT, const KArgs& a, hipStream_t st) {
    const Solve2pInst* s = s2p_find(HT, L, ZR, ACT);
    if (!s) return hipErrorNotSupported;
    const MfmaLayout lay(HT, L, ZR, 0, true, 0);
    const int lds = ((lay.lds_total + 3) / 4 * 4 + 48 + 2 * L * HT * 256 + 128) * (int)sizeof(float);
    const long long ntiles = (a.B + 15) / 16;
    return s->fn(a, lds, (int)ntiles, st);
}

// nthreads: 256 / 512 (one / two waves per SIMD), 0 = the instance table's first
hipError_t solve2_launch(int HT, int L, int ZR, int ACT, int nthreads, const KArgs& a, int num_cus, hipStream_t st) {
    const Solve2Inst* s = s2_find(HT, L, ZR, ACT, nthreads);
    if (!s) s = s2_find(HT, L, ZR, ACT, 0);
    if (!s) return hipErrorNotSupported;
    const MfmaLayout lay(HT, L, ZR, 0, true, 0);
    const int lds = ((lay.lds_total + 3) / 4 * 4 + 48) * (int)sizeof(float);
    const long long ntiles = (a.B + 15) / 16;
    const long long cap = num_cus;   // one workgroup per CU (the image takes more than half the LDS)
    const int nblocks = (int)(ntiles < cap ? ntiles : cap);
    return s->fn(a, lds, nblocks, st);
}

}  // namespace cnf
