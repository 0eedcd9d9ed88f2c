// cnf_grad.hip — parameter gradient of the FFJORD loss through the fixed-step solve (gfx950).
//
// SURVEY.md §8(f) rank 2.  The reference differentiates `loss` through SciMLBase.solve with
// QuadratureAdjoint + ZygoteVJP (src/core/icnf.jl:90-99; src/exts/mlj_ext/core_icnf.jl:42-51).
// With a fixed-step solver the exact gradient of the discrete loss is reverse mode through the RK
// steps (discretise-then-optimise): grad = sum_j d(-logp_j)/dp over the batch columns given.
//
// Regularised objective (TrainMode{true}: + l1 |zdot| + l2 |eps^T J| + l3 |z_aug|, icnf.jl:184-251,628-637):
//   kbar += c_E zdot/|zdot|;  gbar = c_n g/|g| - c_l eps with g = W_1[:,0:D]^T delta_1;  dbar_1 = W_1[:,0:D] gbar
//   (the hoisted-q shortcut is the special case c_n = 0);  lambda_N += l3 z_aug/|z_aug|.
//
// Two launches per gradient: the forward solve kernel (cnf_mfma_kernel.h) with step checkpoints
// z_n, then this reverse sweep.  Per wave: one 16-sample tile, steps in reverse; per stage
//   recompute   h_l, act'_l                                   (forward images)
//   pullback    delta_L = c .* act'_L, u_l = W_{l+1}^T delta_{l+1}, delta_l = u_l .* act'_l
//   reverse of  Phi = kbar^T zdot - c_l <delta_1, q>          (c = W_N^T eps, q = W_1[:,0:D] eps hoisted)
//     bottom-up: dbar_1 = -c_l q; ubar_l = dbar_l .* act'_l; abar''_l = dbar_l .* u_l;
//                dbar_{l+1} = W_{l+1} ubar_l;  Wbar_{l+1} += delta_{l+1} ubar_l^T
//     top-down : hbar_L = W_N^T kbar; abar_l = hbar_l .* act'_l + abar''_l .* act''_l;
//                Wbar_l += abar_l h_{l-1}^T; bbar_l += abar_l; hbar_{l-1} = W_l^T abar_l;  Zbar = W_1[:,0:D]^T abar_1
// Weight cotangents are outer products summed over the tile's 16 samples: MFMAs with the SAMPLE
// index on K, so each operand tile is transposed once through wave-private LDS scratch
// (ds_write_b128 into a padded tile + 4 conflict-free ds_read_b32).  Accumulation: the cotangent tiles
// live in registers for the whole launch, partitioned over the workgroup's four waves (wave w owns row
// block w of every hidden matrix and of W_1, column tile w of W_N); the waves run in lockstep and
// publish their operand tiles through an LDS exchange buffer once per matrix per stage; biases are
// outer products with a ones column.  At the end each wave deposits its tiles in a slab and a second
// kernel sums the slabs in a fixed order into the Lux-layout gradient: no atomics, bit-reproducible.
// (History: LDS float atomics - ds_add_f32 retires ~1 lane per 3.4 cycles per CU - 89 ms at cfg2;
// wave-private slabs updated by load/MFMA/store - 64 GB of fabric traffic per gradient - 19 ms.)
#include "cnf_grad_dev.h"

#ifdef G_TRACE
#define G_T(k) do { asm volatile("" ::: "memory"); tr[k] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#else
#define G_T(k)
#endif

namespace cnf {

template <int HT, int L, int ZR, int CR, int ACT>
__global__ void __launch_bounds__(256)
mfma_grad_kernel(GArgs a) {
    using G = GradLds<HT, L, ZR, CR, ACT>;
    constexpr MfmaLayout LAY(HT, L, ZR, CR, true, 0);
    constexpr int DT = G::DT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.packed);
        f32x4* dst = reinterpret_cast<f32x4*>(smem);
        for (int i = threadIdx.x; i < LAY.total / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    using SL = GradSlab<HT, L, ZR, CR>;
    float* slab = a.slab + ((long long)blockIdx.x * 4 + wave) * SL::TOTAL;
    // B fragment of the ones column (feature 0 = 1 for every sample): bias cotangents as outer products
    float onesf[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) onesf[q4] = (lane & 15) == 0 ? 1.f : 0.f;
    const long long ntiles = (a.B + 15) / 16;
    const int D = a.D;
    const bool autonomous = a.autonomous;
    const float dt0 = a.dt;
    const int ns = a.T.ns;

    // Hidden-matrix cotangents live in registers for the whole launch: wave w owns row block w
    // (W̄[16w .. 16w+15][:]) of every hidden matrix and accumulates the outer products of ALL four
    // waves' sample tiles for it; the operand tiles travel through the LDS exchange buffer once per
    // hidden matrix per stage.  (Slab read-modify-write of these tiles cost 64 GB of fabric traffic
    // per gradient at cfg2.)  The four waves therefore run the tile loop in lockstep.
    f32x4 Wh[L > 1 ? L - 1 : 1][HT], Bh[L > 1 ? L - 1 : 1], W1acc[SL::NT1], WNacc, BNacc;
#pragma unroll
    for (int l = 0; l < (L > 1 ? L - 1 : 1); ++l) { zero_tiles<HT>(Wh[l]); Bh[l] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    zero_tiles<SL::NT1>(W1acc);
    WNacc = BNacc = f32x4{0.f, 0.f, 0.f, 0.f};
    float* xch = smem + G::XCH;
    float* xmine = xch + wave * G::XCH_W;
    const long long ngroups = (ntiles + 3) / 4;
    for (long long tg = blockIdx.x; tg < ngroups; tg += gridDim.x) {
        const long long tile_raw = tg * 4 + wave;
        const bool tile_ok = tile_raw < ntiles;
        const long long tile = tile_ok ? tile_raw : ntiles - 1;   // idle waves replay the last tile with zero cotangents
        const long long smp = tile * 16 + n;
        const bool valid = tile_ok && smp < a.B;
        const long long sc = smp < a.B ? smp : a.B - 1;
        float eps[ZR], lam[ZR];
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            eps[s] = f < D ? a.eps[sc * D + f] : 0.f;
            // dL/dz_N = z_N  (L = sum_j -logp_j, -log N(z) = |z|^2/2 + const); zero for padding columns
            lam[s] = valid ? a.ckpt[(((long long)a.nsteps * ntiles + tile) * 64 + lane) * a.ckpt_zr + s] : 0.f;
        }
        if (a.lam3 != 0.f) {   // + l3 |z_aug|_2 at the final time (src/core/base_icnf.jl:106-122)
            float sa = 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) sa = fmaf(lam[s], lam[s], sa); }
            sa = group_sum(sa);
            const float inv = sa > 0.f ? a.lam3 * rsqrtf(sa) : 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) lam[s] = fmaf(inv, lam[s], lam[s]); }
        }
        float y[CR > 0 ? CR : 1];
        y[0] = 0.f;
        if constexpr (CR > 0) {
#pragma unroll
            for (int s = 0; s < CR; ++s) { const int f = 4 * s + g; y[s] = f < a.C ? a.ys[sc * a.C + f] : 0.f; }
        }
        f32x4 y_tile[1];
        y_tile[0] = dense_tile<(CR > 0 ? CR : 1)>(y);   // condition rows as an accumulator-layout tile
        f32x4 cvec[HT], qvec[HT];   // c = W_N^T eps, q = W_1[:,0:D] eps: constant over the solve
        zero_tiles<HT>(cvec);
        zero_tiles<HT>(qvec);
        gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps}, cvec);
        gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{eps}, qvec);
        // eps as an accumulator-layout pseudo tile (rows = state features)
        f32x4 eps_tile[1];
        eps_tile[0] = dense_tile<ZR>(eps);

#pragma clang loop unroll(disable)
        for (int step = a.nsteps - 1; step >= 0; --step) {
            float tn = a.t0 + (float)step * dt0, dt = dt0;
            if (a.tgrid) { tn = a.tgrid[step]; dt = a.tgrid[step + 1] - tn; }
            float zn[ZR];
#pragma unroll
            for (int s = 0; s < ZR; ++s) zn[s] = a.ckpt[(((long long)step * ntiles + tile) * 64 + lane) * a.ckpt_zr + s];
            // ---- forward sweep of the step: stage derivatives kz_i (z rows only) ----
            float kz[6][ZR];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int s = 0; s < ZR; ++s) kz[j][s] = 0.f;
            if (a.ckpt_k) {
                // stage derivatives were checkpointed by the forward kernel
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    if (j < ns) {
#pragma unroll
                        for (int s = 0; s < ZR; ++s)
                            kz[j][s] = a.ckpt_k[((((long long)step * ns + j) * ntiles + tile) * 64 + lane) * a.ckpt_zr + s];
                    }
            } else {
#pragma clang loop unroll(disable)
            for (int st = 0; st < ns; ++st) {
                float zs[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc = fmaf(a.T.a[st][j], kz[j][s], acc);
                    zs[s] = fmaf(dt, acc, zn[s]);
                }
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                const float* sm = smem + opaque;
                f32x4 h[L][HT], d[L][HT];
                grad_forward<HT, L, ZR, CR, ACT>(sm, lane, tn + a.T.c[st] * dt, autonomous, zs, y, h, d);
                f32x4 zacc[DT];
                load_cvec<DT>(sm + LAY.v_bN, g, zacc);
                gemm_tiles<DT, 4 * HT>(sm + LAY.fN, lane, TileIn<HT>{h[L - 1]}, zacc);
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int s = 0; s < ZR; ++s) kz[j][s] = (j == st) ? zacc[s >> 2][s & 3] : kz[j][s];
            }
            }
            // ---- reverse sweep over the stages ----
            float Zb[6][ZR];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int s = 0; s < ZR; ++s) Zb[j][s] = 0.f;
#pragma clang loop unroll(disable)
            for (int st = ns - 1; st >= 0; --st) {
#ifdef G_TRACE
                unsigned long long tr[28];
#endif
                G_T(0);
                float zs[ZR], kbar[ZR];
                const float bi = a.T.b[st];
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    float acc = 0.f, kb = bi * lam[s];
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc = fmaf(a.T.a[st][j], kz[j][s], acc);
#pragma unroll
                    for (int j = 1; j < 6; ++j) kb = fmaf(a.T.a[j][st], Zb[j][s], kb);   // a[j][st] != 0 only for j > st
                    zs[s] = fmaf(dt, acc, zn[s]);
                    kbar[s] = dt * kb;
                }
                const float cl = valid ? dt * bi : 0.f;   // cotangent of ldot: dL/d(dlogp) = +1
                const float cE = cl * a.lam1, cn = cl * a.lam2;   // cotangents of Edot, ndot
                const bool regz = a.lam1 != 0.f, regj = a.lam2 != 0.f;   // wave-uniform
                const float tt = tn + a.T.c[st] * dt;
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                const float* sm = smem + opaque;

                // (1) recompute, (2) first-order pullback
                f32x4 h[L][HT], d[L][HT], dl[L][HT], u[L][HT];
                grad_forward<HT, L, ZR, CR, ACT>(sm, lane, tt, autonomous, zs, y, h, d);
                G_T(1);
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) { u[L - 1][mt] = cvec[mt]; dl[L - 1][mt] = cvec[mt] * d[L - 1][mt]; }
#pragma unroll
                for (int l = L - 1; l >= 1; --l) {
                    zero_tiles<HT>(u[l - 1]);
                    gemm_tiles<HT, 4 * HT>(sm + LAY.bh + (l - 1) * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{dl[l]}, u[l - 1]);
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) dl[l - 1][mt] = u[l - 1][mt] * d[l - 1][mt];
                }
                G_T(2);
                // (3) bottom-up through the pullback: dbar, second-order terms, Wbar_{l+1} += delta_{l+1} ubar_l^T
                if (regz) {   // Edot = |zdot|: kbar += c_E zdot / |zdot|
                    f32x4 zacc[DT];
                    load_cvec<DT>(sm + LAY.v_bN, g, zacc);
                    gemm_tiles<DT, 4 * HT>(sm + LAY.fN, lane, TileIn<HT>{h[L - 1]}, zacc);
                    float e2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) e2 = fmaf(zacc[s >> 2][s & 3], zacc[s >> 2][s & 3], e2);
                    e2 = group_sum(e2);
                    const float inv = e2 > 0.f ? cE * rsqrtf(e2) : 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) kbar[s] = fmaf(inv, zacc[s >> 2][s & 3], kbar[s]);
                }
                // gbar = cotangent of g = eps^T J (dense layout): -c_l eps (+ c_n g/|g|)
                float gbar[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) gbar[s] = -cl * eps[s];
                f32x4 db[HT], a2[L][HT];   // a2_l = dbar_l .* u_l  (multiplies act''_l later)
                if (regj) {
                    f32x4 gacc[DT];
                    zero_tiles<DT>(gacc);
                    gemm_tiles<DT, 4 * HT>(sm + LAY.b1, lane, TileIn<HT>{dl[0]}, gacc);   // g = W_1[:,0:D]^T delta_1
                    float n2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) n2 = fmaf(gacc[s >> 2][s & 3], gacc[s >> 2][s & 3], n2);
                    n2 = group_sum(n2);
                    const float inv = n2 > 0.f ? cn * rsqrtf(n2) : 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) gbar[s] = fmaf(inv, gacc[s >> 2][s & 3], gbar[s]);
                    zero_tiles<HT>(db);
                    gemm_tiles<HT, ZR>(sm + LAY.f1z, lane, RegIn<ZR>{gbar}, db);              // dbar_1 = W_1[:,0:D] gbar
                } else {
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) db[mt] = qvec[mt] * (-cl);                // = W_1[:,0:D] (-c_l eps)
                }
                f32x4 gb_tile[1];
                gb_tile[0] = dense_tile<ZR>(gbar);
                f32x4 ubs[L > 1 ? L - 1 : 1][HT];   // ubar_l kept: its outer product is merged with the top-down one
#pragma unroll
                for (int l = 0; l < L - 1; ++l) {
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) { ubs[l][mt] = db[mt] * d[l][mt]; a2[l][mt] = db[mt] * u[l][mt]; }
                    zero_tiles<HT>(db);
                    gemm_tiles<HT, 4 * HT>(sm + LAY.fh + l * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{ubs[l]}, db);   // W_{l+2} ubar
                }
                f32x4 cb[HT];   // cbar = dbar_L .* act'_L: Wbar_N[i, f] += eps_i cbar_f
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) { cb[mt] = db[mt] * d[L - 1][mt]; a2[L - 1][mt] = db[mt] * cvec[mt]; }
                G_T(3);
                // (4) top-down through the forward chain
                f32x4 kb_tile[1];
                kb_tile[0] = dense_tile<ZR>(kbar);
                {   // Wbar_N += eps cbar^T + kbar h_L^T;  bbar_N += kbar x ones.  Wave w owns column tile w.
                    tile_store(xmine + 0 * TS, lane, eps_tile[0]);
                    tile_store(xmine + 1 * TS, lane, kb_tile[0]);
                    tiles_store<HT>(xmine + 2 * TS, lane, cb);
                    tiles_store<HT>(xmine + (2 + HT) * TS, lane, h[L - 1]);
                    G_T(4);
                    __syncthreads();
                    G_T(5);
                    if (HT >= 4 || wave < HT) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float* xv = xch + v * G::XCH_W;
                            float a1[4], a2f[4], b1[4], b2[4];
                            read_frag_A(xv + 0 * TS, lane, a1);
                            read_frag_A(xv + 1 * TS, lane, a2f);
                            read_frag_B(xv + (2 + wave) * TS, lane, b1);
                            read_frag_B(xv + (2 + HT + wave) * TS, lane, b2);
                            WNacc = outer4(a1, b1, WNacc);
                            WNacc = outer4(a2f, b2, WNacc);
                            if (wave == 0) BNacc = outer4(a2f, onesf, BNacc);
                        }
                    }
                    G_T(6);
                    __syncthreads();
                    G_T(7);
                }
                f32x4 hb[HT];
                zero_tiles<HT>(hb);
                gemm_tiles<HT, ZR>(sm + LAY.bN, lane, RegIn<ZR>{kbar}, hb);   // W_N^T kbar
                G_T(8);
                float Zbar[ZR];
#pragma unroll
                for (int l = L - 1; l >= 0; --l) {
                    f32x4 ab[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) {
                        // act'': tanh -> -2 h (1 - h^2);  softplus -> s (1 - s) with s = act' = sigmoid(a)
                        const f32x4 d2 = ACT == CNF_ACT_TANH ? h[l][mt] * d[l][mt] * -2.f : d[l][mt] * (1.f - d[l][mt]);
                        ab[mt] = hb[mt] * d[l][mt] + a2[l][mt] * d2;
                    }
                    G_T(9 + 6 * (L - 1 - l));
                    if (l > 0) {
                        // Wbar_{l+1} += abar_l h_{l-1}^T + delta_l ubar_{l-1}^T ;  bbar_{l+1} += abar_l x ones.
                        // Publish this wave's operand tiles; wave w accumulates row block w over all 4 waves.
                        tiles_store<HT>(xmine + 0 * HT * TS, lane, ab);           // A1 = abar_l
                        tiles_store<HT>(xmine + 1 * HT * TS, lane, h[l - 1]);     // B1 = h_{l-1}
                        tiles_store<HT>(xmine + 2 * HT * TS, lane, dl[l]);        // A2 = delta_l
                        tiles_store<HT>(xmine + 3 * HT * TS, lane, ubs[l - 1]);   // B2 = ubar_{l-1}
                        G_T(10 + 6 * (L - 1 - l));
                        __syncthreads();
                        G_T(11 + 6 * (L - 1 - l));
                        if (HT >= 4 || wave < HT) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float* xv = xch + v * G::XCH_W;
                                float a1[4], a2f[4];
                                read_frag_A(xv + (0 * HT + wave) * TS, lane, a1);
                                read_frag_A(xv + (2 * HT + wave) * TS, lane, a2f);
                                Bh[l - 1] = outer4(a1, onesf, Bh[l - 1]);
#pragma unroll
                                for (int nt = 0; nt < HT; ++nt) {
                                    float b1[4], b2[4];
                                    read_frag_B(xv + (1 * HT + nt) * TS, lane, b1);
                                    read_frag_B(xv + (3 * HT + nt) * TS, lane, b2);
                                    Wh[l - 1][nt] = outer4(a1, b1, Wh[l - 1][nt]);
                                    Wh[l - 1][nt] = outer4(a2f, b2, Wh[l - 1][nt]);
                                }
                            }
                        }
                        G_T(12 + 6 * (L - 1 - l));
                        __syncthreads();   // exchange buffer is reused by the next hidden matrix / stage
                        G_T(13 + 6 * (L - 1 - l));
                        zero_tiles<HT>(hb);
                        gemm_tiles<HT, 4 * HT>(sm + LAY.bh + (l - 1) * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{ab}, hb);   // W_{l+1}^T abar
                        G_T(14 + 6 * (L - 1 - l));
                    } else {
                        // input pseudo tile [z (D rows); t; ...; 1 at feature 15]: feature j <-> (register j>>2, lane group j&3)
                        f32x4 in_tile;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = r < ZR ? zs[r < ZR ? r : 0] : 0.f;
                            if (!autonomous && 4 * r + g == D) v = tt;
                            if (4 * r + g > D || (autonomous && 4 * r + g == D)) v = 0.f;
                            if (4 * r + g == 15) v = 1.f;
                            in_tile[r] = v;
                        }
                        // Wbar_1 += abar_1 [z; t; 1]^T + delta_1 [gbar; 0]^T (+ abar_1 y^T);  wave w owns row block w
                        tiles_store<HT>(xmine + 0 * HT * TS, lane, ab);
                        tiles_store<HT>(xmine + 1 * HT * TS, lane, dl[0]);
                        tile_store(xmine + (2 * HT + 0) * TS, lane, in_tile);
                        tile_store(xmine + (2 * HT + 1) * TS, lane, gb_tile[0]);
                        if constexpr (CR > 0) tile_store(xmine + (2 * HT + 2) * TS, lane, y_tile[0]);
                        G_T(10 + 6 * (L - 1 - l));
                        __syncthreads();
                        G_T(11 + 6 * (L - 1 - l));
                        if (HT >= 4 || wave < HT) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float* xv = xch + v * G::XCH_W;
                                float a1[4], a2f[4], b1[4], b2[4];
                                read_frag_A(xv + (0 * HT + wave) * TS, lane, a1);
                                read_frag_A(xv + (1 * HT + wave) * TS, lane, a2f);
                                read_frag_B(xv + (2 * HT + 0) * TS, lane, b1);
                                read_frag_B(xv + (2 * HT + 1) * TS, lane, b2);
                                W1acc[0] = outer4(a1, b1, W1acc[0]);
                                W1acc[0] = outer4(a2f, b2, W1acc[0]);
                                if constexpr (CR > 0) {
                                    float by[4];
                                    read_frag_B(xv + (2 * HT + 2) * TS, lane, by);
                                    W1acc[SL::NT1 - 1] = outer4(a1, by, W1acc[SL::NT1 - 1]);
                                }
                            }
                        }
                        G_T(12 + 6 * (L - 1 - l));
                        __syncthreads();
                        G_T(13 + 6 * (L - 1 - l));
                        f32x4 zb[DT];
                        zero_tiles<DT>(zb);
                        gemm_tiles<DT, 4 * HT>(sm + LAY.b1, lane, TileIn<HT>{ab}, zb);   // W_1[:,0:D]^T abar_1
#pragma unroll
                        for (int s = 0; s < ZR; ++s) Zbar[s] = zb[s >> 2][s & 3];
                    }
                }
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int s = 0; s < ZR; ++s) Zb[j][s] = (j == st) ? Zbar[s] : Zb[j][s];
#ifdef G_TRACE
                G_T(14 + 6 * (L - 1));
                if (blockIdx.x == 3 && step == 5 && st == 1 && lane == 0 && tg == blockIdx.x) {
                    printf("w%d:", wave);
                    for (int k = 1; k <= 14 + 6 * (L - 1); ++k) printf(" %d", (int)(tr[k] - tr[k - 1]));
                    printf(" | total %d\n", (int)(tr[14 + 6 * (L - 1)] - tr[0]));
                }
#endif
            }
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                float acc = lam[s];
#pragma unroll
                for (int j = 0; j < 6; ++j) acc += Zb[j][s];
                lam[s] = acc;
            }
        }
        if (a.grad_x && valid) {   // costate at t0 = dL/dz_0; its first nvars rows are dL/dx
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                const int f = 4 * s + g;
                if (f < a.nvars) a.grad_x[smp * a.nvars + f] = lam[s];
            }
        }
    }
    // every wave deposits the tiles it owns in its own (zeroed) slab; grad_reduce_kernel sums the slabs
    if (HT >= 4 || wave < HT) {
#pragma unroll
        for (int l = 0; l < L - 1; ++l) {
#pragma unroll
            for (int nt = 0; nt < HT; ++nt)
                *reinterpret_cast<f32x4*>(slab + SL::WH + l * HT * HT * 256 + ((wave * HT + nt) * 64 + lane) * 4) = Wh[l][nt];
            *reinterpret_cast<f32x4*>(slab + SL::BH + l * HT * 256 + (wave * 64 + lane) * 4) = Bh[l];
        }
#pragma unroll
        for (int it = 0; it < SL::NT1; ++it)
            *reinterpret_cast<f32x4*>(slab + SL::W1 + ((wave * SL::NT1 + it) * 64 + lane) * 4) = W1acc[it];
        *reinterpret_cast<f32x4*>(slab + SL::WN + (wave * 64 + lane) * 4) = WNacc;
        if (wave == 0) *reinterpret_cast<f32x4*>(slab + SL::BN + lane * 4) = BNacc;
    }
}

// Sum the waves' slabs in a fixed order and scatter into the Lux-layout gradient (every parameter is
// written by exactly one thread: no atomics).
template <int HT, int L, int ZR, int CR>
__global__ void __launch_bounds__(1024)
grad_reduce_kernel(const float* __restrict__ slab, int nwaves, GArgs a, float* __restrict__ grad) {
    using SL = GradSlab<HT, L, ZR, CR>;
    // 64 elements per block, 16 slab groups per element (fixed partition, fixed combine order; with 4 groups a thread summed 256
    // slabs one load after the other: 97 us at cfg2 for 51 MB)
    __shared__ float part[16][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float acc0 = 0.f, acc1 = 0.f;
    if (e < SL::TOTAL) {
        int w = grp;
        for (; w + 16 < nwaves; w += 32) {   // two independent chains: two loads in flight per thread
            acc0 += slab[(long long)w * SL::TOTAL + e];
            acc1 += slab[(long long)(w + 16) * SL::TOTAL + e];
        }
        if (w < nwaves) acc0 += slab[(long long)w * SL::TOTAL + e];
    }
    part[grp][el] = acc0 + acc1;
    __syncthreads();
    if (grp != 0 || e >= SL::TOTAL) return;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) sum += (part[4 * q][el] + part[4 * q + 1][el]) + (part[4 * q + 2][el] + part[4 * q + 3][el]);
    const int H = a.H, D = a.D;
    const int ln = (e >> 2) & 63, r = e & 3, n = ln & 15, gg = ln >> 4;
    if (e < SL::WH) {                                   // W_1 image: [mt][input tile][lane][r]
        const int tl = (e - SL::W1) / 256, mt = tl / SL::NT1, it = tl % SL::NT1;
        const int out = 16 * mt + 4 * r + gg;
        const int ncore = D + (a.autonomous ? 0 : 1);   // z and time columns live in input tile 0
        if (out < H && it == 0 && n < ncore) grad[a.w_off[0] + out + H * n] = sum;
        if (out < H && it == 0 && n == 15) grad[a.b_off[0] + out] = sum;       // ones column
        if (out < H && it == 1 && n < a.C) grad[a.w_off[0] + out + H * (ncore + n)] = sum;   // condition columns
    } else if (e < SL::WN) {                            // hidden images
        const int rel = e - SL::WH, l = rel / (HT * HT * 256), tl = (rel / 256) % (HT * HT);
        const int out = 16 * (tl / HT) + 4 * r + gg, in = 16 * (tl % HT) + n;
        if (out < H && in < H) grad[a.w_off[l + 1] + out + H * in] = sum;
    } else if (e < SL::BH) {                            // W_N image: rows = state features
        const int tl = (e - SL::WN) / 256;
        const int out = 16 * (tl / HT) + 4 * r + gg, in = 16 * (tl % HT) + n;
        if (out < D && in < H) grad[a.w_off[L] + out + D * in] = sum;
    } else if (e < SL::BN) {                            // hidden biases: column 0
        const int rel = e - SL::BH, l = rel / (HT * 256), mt = (rel / 256) % HT;
        const int out = 16 * mt + 4 * r + gg;
        if (out < H && n == 0) grad[a.b_off[l + 1] + out] = sum;
    } else {
        const int mt = (e - SL::BN) / 256;
        const int out = 16 * mt + 4 * r + gg;
        if (out < D && n == 0) grad[a.b_off[L] + out] = sum;
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct GradInst {
    int HT, L, ZR, CR, ACT;
    int lds_bytes, slab_total, packed_floats;
    void (*kern)(GArgs);
    void (*reduce)(const float*, int, GArgs, float*);
};
#define GRAD_INST(HT, L, ZR, CR, ACT)                                                                        \
    GradInst { HT, L, ZR, CR, ACT, GradLds<HT, L, ZR, CR, ACT>::TOTAL * 4, GradSlab<HT, L, ZR, CR>::TOTAL,   \
               MfmaLayout(HT, L, ZR, CR, true, 0).total, &mfma_grad_kernel<HT, L, ZR, CR, ACT>,            \
               &grad_reduce_kernel<HT, L, ZR, CR> }
#define GRAD_HT(HT, CR, ACT)                                                                                 \
    GRAD_INST(HT, 3, 2, CR, ACT), GRAD_INST(HT, 2, 2, CR, ACT), GRAD_INST(HT, 3, 4, CR, ACT), GRAD_INST(HT, 2, 4, CR, ACT)
#define GRAD_SHAPES(CR, ACT) GRAD_HT(1, CR, ACT), GRAD_HT(2, CR, ACT), GRAD_HT(3, CR, ACT), GRAD_HT(4, CR, ACT)
static const GradInst kGrad[] = {GRAD_SHAPES(0, CNF_ACT_TANH), GRAD_SHAPES(0, CNF_ACT_SOFTPLUS),
                                 GRAD_SHAPES(4, CNF_ACT_TANH), GRAD_SHAPES(4, CNF_ACT_SOFTPLUS)};

static const GradInst* grad_find(const cnf_config& c) {
    if (c.mode != CNF_MODE_HUTCH_VJP || c.nprobes < 1 || c.ncond > 16) return nullptr;
    const int N = c.n_layers, L = N - 1;
    if (L < 2 || L > 3 || c.acts[N - 1] != CNF_ACT_IDENTITY) return nullptr;
    const int H = c.widths[1];
    for (int l = 0; l < L; ++l)
        if (c.acts[l] != c.acts[0] || c.widths[l + 1] != H) return nullptr;
    const int D = c.nvars + c.naug, HT = (H + 15) / 16, ZR = (D + 3) / 4;
    if (D + (c.autonomous ? 0 : 1) > 15) return nullptr;   // input tile: 16 columns, the last one is the bias column
    const GradInst* best = nullptr;
    for (const GradInst& g : kGrad)
        if (g.HT == HT && g.L == L && g.ACT == c.acts[0] && g.ZR >= ZR && (g.CR > 0) == (c.ncond > 0) &&
            (!best || g.ZR < best->ZR))
            best = &g;
    return best;
}

bool grad_supported(const cnf_config& c) { return grad_find(c) != nullptr; }
size_t grad_packed_bytes(const cnf_config& c) { return (size_t)grad_find(c)->packed_floats * sizeof(float); }
size_t grad_slab_floats(const cnf_config& c, int num_cus) { return (size_t)num_cus * 4 * grad_find(c)->slab_total; }
void grad_shape(const cnf_config& c, int* HT, int* L, int* ZR, int* CR) {
    const GradInst* g = grad_find(c);
    *HT = g->HT; *L = g->L; *ZR = g->ZR; *CR = g->CR;
}

hipError_t grad_launch(const cnf_config& c, const float* packed_dev, const float* ckpt, const float* ckpt_k,
                       int ckpt_zr, const float* eps, const float* ys,
                       const size_t* w_off, const size_t* b_off, int alg, int nsteps, float t0, float t1, const float* tgrid_dev,
                       float probe_w, long long B, const float lam[3], float* slab, float* grad, float* grad_x, int num_cus, hipStream_t st) {
    const GradInst* gi = grad_find(c);
    if (!gi) return hipErrorNotSupported;
    // > 64 KB of dynamic LDS has to be enabled once per device and kernel
    static DeviceOnce done_mask[sizeof(kGrad) / sizeof(kGrad[0])];
    const int idx = (int)(gi - kGrad);
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    // several probes: same layouts, probe-loop kernel (cnf_grad_probes.hip); it needs the stage checkpoints
    GradKernel kern = gi->kern;
    if (c.nprobes > 1) {
        kern = grad_probes_kernel(gi->HT, gi->L, gi->ZR, gi->CR, gi->ACT);
        if (!kern || !ckpt_k) return hipErrorNotSupported;
    }
    static DeviceOnce done_probes[sizeof(kGrad) / sizeof(kGrad[0])];
    // one probe with stage checkpoints: the barrier-free form (cnf_grad2.hip) where it has an instance; CNF_GRAD_V1=1 keeps the
    // exchange form above (A/B switch)
    static DeviceOnce done_v2[sizeof(kGrad) / sizeof(kGrad[0])];
    bool v2 = false;
    static DeviceOnce done_v2p[sizeof(kGrad) / sizeof(kGrad[0])];
    if (ckpt_k && !tuning().grad_v1) {
        GradKernel k2 = c.nprobes == 1 ? grad2_kernel(gi->HT, gi->L, gi->ZR, gi->CR, gi->ACT) : grad2_probes_kernel(gi->HT, gi->L, gi->ZR, gi->CR, gi->ACT);
        if (k2) { kern = k2; v2 = true; }
    }
    DeviceOnce& done = (v2 ? (c.nprobes > 1 ? done_v2p : done_v2) : c.nprobes > 1 ? done_probes : done_mask)[idx];
    if (!done.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, gi->lds_bytes);
        if (e != hipSuccess) return e;
        done.set(dev);
    }
    GArgs a{};
    a.packed = packed_dev; a.ckpt = ckpt; a.ckpt_k = ckpt_k; a.ckpt_zr = ckpt_zr; a.eps = eps; a.K = c.nprobes; a.ys = ys; a.C = c.ncond; a.slab = slab; a.grad_x = grad_x; a.B = B;
    a.nsteps = nsteps; a.t0 = t0; a.dt = (t1 - t0) / (float)nsteps; a.tgrid = tgrid_dev; a.probe_w = probe_w;
    a.D = c.nvars + c.naug; a.H = c.widths[1]; a.n_in = c.widths[0]; a.autonomous = c.autonomous; a.nvars = c.nvars;
    a.lam1 = lam[0]; a.lam2 = lam[1]; a.lam3 = lam[2];
    for (int l = 0; l < c.n_layers; ++l) { a.w_off[l] = (int)w_off[l]; a.b_off[l] = (int)b_off[l]; }
    a.T = make_tableau(alg);
    const long long ntiles = (B + 15) / 16;
    const long long want = (ntiles + 3) / 4;
    const int nblocks = (int)(want < num_cus ? want : num_cus);
    const int nwaves = nblocks * 4;
    hipError_t e = zero_async(slab, (size_t)nwaves * gi->slab_total * sizeof(float), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), gi->lds_bytes, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gi->reduce, dim3((gi->slab_total + 63) / 64), dim3(1024), 0, st, slab, nwaves, a, grad);
    return hipGetLastError();
}

}  // namespace cnf
