// cnf_grad_probes.hip — parameter gradient with K > 1 Hutchinson probes (RNODE training, BASELINE cfg3).
//
// Same method, data layout, slab and reduce kernel as cnf_grad.hip (read that header first).  With K
// probes the stage objective is
//   Phi = kbar^T zdot + sum_k [ -(c_l/K) <eps_k, g_k> + (c_n/K) |g_k| ],   g_k = W_1[:,0:D]^T delta_1^k
// (src/core/icnf.jl:184-251 with the probe mean of this framework's nprobes extension): the forward
// chain h_l, act'_l is shared, the first-order pullback and its bottom-up reverse run once per probe,
// and the top-down pass runs once with abar''_l = sum_k dbar_l^k .* u_l^k.
//
// Per stage:   recompute h, act'            (once)
//   per probe: c_k = W_N^T eps_k;  u_l^k, delta_l^k;  gbar_k;  dbar_1^k = W_1[:,0:D] gbar_k;
//              bottom-up: ubar_l^k, abar'' += ...;  Wbar_{l+2} += delta_{l+1}^k ubar_l^k^T  (exchange per matrix)
//              Wbar_N += eps_k cbar_k^T;  Wbar_1 += delta_1^k [gbar_k; 0]^T                  (one exchange)
//   top-down:  Wbar_N += kbar h_L^T; Wbar_l += abar_l h_{l-1}^T; biases; Zbar              (as for K = 1)
// The probe loop is rolled (runtime K); nothing in registers is indexed by k: eps_k is re-read from
// HBM/L2 (ZR floats per lane per probe per stage) and c_k is recomputed (HT*ZR MFMAs of ~1000).
#include "cnf_grad_dev.h"

namespace cnf {

// publish A (HT tiles) and B (HT tiles); wave w accumulates  W[nt] += sum_waves A[w] B[nt]^T
template <int HT>
__device__ __forceinline__ void exchange_hidden(float* __restrict__ xch, int xch_w, int lane, int wave,
                                                const f32x4 (&A)[HT], const f32x4 (&Bt)[HT], f32x4 (&W)[HT]) {
    float* xmine = xch + wave * xch_w;
    tiles_store<HT>(xmine, lane, A);
    tiles_store<HT>(xmine + HT * TS, lane, Bt);
    __syncthreads();
    if (HT >= 4 || wave < HT) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float* xv = xch + v * xch_w;
            float a1[4];
            read_frag_A(xv + wave * TS, lane, a1);
#pragma unroll
            for (int nt = 0; nt < HT; ++nt) {
                float b1[4];
                read_frag_B(xv + (HT + nt) * TS, lane, b1);
                W[nt] = outer4(a1, b1, W[nt]);
            }
        }
    }
    __syncthreads();
}

template <int HT, int L, int ZR, int CR, int ACT>
__global__ void __launch_bounds__(256)
mfma_grad_probes_kernel(GArgs a) {
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
    float onesf[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) onesf[q4] = (lane & 15) == 0 ? 1.f : 0.f;
    const long long ntiles = (a.B + 15) / 16;
    const int D = a.D, K = a.K;
    const float invK = a.probe_w > 0.f ? a.probe_w : 1.f / (float)K;
    const bool autonomous = a.autonomous;
    const float dt0 = a.dt;
    const int ns = a.T.ns;

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
        float lam[ZR];
#pragma unroll
        for (int s = 0; s < ZR; ++s)
            lam[s] = valid ? a.ckpt[(((long long)a.nsteps * ntiles + tile) * 64 + lane) * a.ckpt_zr + s] : 0.f;
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
        y_tile[0] = dense_tile<(CR > 0 ? CR : 1)>(y);

#pragma clang loop unroll(disable)
        for (int step = a.nsteps - 1; step >= 0; --step) {
            float tn = a.t0 + (float)step * dt0, dt = dt0;
            if (a.tgrid) { tn = a.tgrid[step]; dt = a.tgrid[step + 1] - tn; }
            float zn[ZR];
#pragma unroll
            for (int s = 0; s < ZR; ++s) zn[s] = a.ckpt[(((long long)step * ntiles + tile) * 64 + lane) * a.ckpt_zr + s];
            // stage derivatives kz_i (z rows), checkpointed by the forward kernel
            float kz[6][ZR];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int s = 0; s < ZR; ++s)
                    kz[j][s] = j < ns ? a.ckpt_k[((((long long)step * ns + j) * ntiles + tile) * 64 + lane) * a.ckpt_zr + s] : 0.f;
            float Zb[6][ZR];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int s = 0; s < ZR; ++s) Zb[j][s] = 0.f;
#pragma clang loop unroll(disable)
            for (int st = ns - 1; st >= 0; --st) {
                float zs[ZR], kbar[ZR];
                const float bi = a.T.b[st];
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    float acc = 0.f, kb = bi * lam[s];
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc = fmaf(a.T.a[st][j], kz[j][s], acc);
#pragma unroll
                    for (int j = 1; j < 6; ++j) kb = fmaf(a.T.a[j][st], Zb[j][s], kb);
                    zs[s] = fmaf(dt, acc, zn[s]);
                    kbar[s] = dt * kb;
                }
                const float cl = valid ? dt * bi : 0.f;
                const float cE = cl * a.lam1, cn = cl * a.lam2;
                const bool regz = a.lam1 != 0.f, regj = a.lam2 != 0.f;
                const float tt = tn + a.T.c[st] * dt;
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                const float* sm = smem + opaque;

                f32x4 h[L][HT], d[L][HT];
                grad_forward<HT, L, ZR, CR, ACT>(sm, lane, tt, autonomous, zs, y, h, d);
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
                f32x4 a2[L][HT];   // sum_k dbar_l^k .* u_l^k  (multiplies act''_l in the top-down pass)
#pragma unroll
                for (int l = 0; l < L; ++l) zero_tiles<HT>(a2[l]);

                // ---- per probe: pullback, its reverse, probe-specific weight cotangents ----
#pragma clang loop unroll(disable)
                for (int k = 0; k < K; ++k) {
                    int opq = 0;
                    asm volatile("" : "+v"(opq));
                    const float* sp = smem + opq;
                    float eps[ZR];
#pragma unroll
                    for (int s = 0; s < ZR; ++s) {
                        const int f = 4 * s + g;
                        eps[s] = f < D ? a.eps[(sc * K + k) * D + f] : 0.f;
                    }
                    f32x4 u[L][HT];
                    zero_tiles<HT>(u[L - 1]);
                    gemm_tiles<HT, ZR>(sp + LAY.bN, lane, RegIn<ZR>{eps}, u[L - 1]);   // c_k = W_N^T eps_k
#pragma unroll
                    for (int l = L - 1; l >= 1; --l) {
                        f32x4 dlt[HT];
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) dlt[mt] = u[l][mt] * d[l][mt];
                        zero_tiles<HT>(u[l - 1]);
                        gemm_tiles<HT, 4 * HT>(sp + LAY.bh + (l - 1) * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{dlt}, u[l - 1]);
                    }
                    f32x4 dl0[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) dl0[mt] = u[0][mt] * d[0][mt];
                    float gbar[ZR];
                    const float clk = cl * invK, cnk = cn * invK;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) gbar[s] = -clk * eps[s];
                    if (regj) {
                        f32x4 gacc[DT];
                        zero_tiles<DT>(gacc);
                        gemm_tiles<DT, 4 * HT>(sp + LAY.b1, lane, TileIn<HT>{dl0}, gacc);   // g_k = W_1[:,0:D]^T delta_1^k
                        float n2 = 0.f;
#pragma unroll
                        for (int s = 0; s < ZR; ++s) n2 = fmaf(gacc[s >> 2][s & 3], gacc[s >> 2][s & 3], n2);
                        n2 = group_sum(n2);
                        const float inv = n2 > 0.f ? cnk * rsqrtf(n2) : 0.f;
#pragma unroll
                        for (int s = 0; s < ZR; ++s) gbar[s] = fmaf(inv, gacc[s >> 2][s & 3], gbar[s]);
                    }
                    f32x4 db[HT];
                    zero_tiles<HT>(db);
                    gemm_tiles<HT, ZR>(sp + LAY.f1z, lane, RegIn<ZR>{gbar}, db);   // dbar_1 = W_1[:,0:D] gbar_k
#pragma unroll
                    for (int l = 0; l < L - 1; ++l) {
                        f32x4 ubs[HT], dln[HT];
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) {
                            ubs[mt] = db[mt] * d[l][mt];
                            a2[l][mt] += db[mt] * u[l][mt];
                            dln[mt] = u[l + 1][mt] * d[l + 1][mt];   // delta_{l+1}^k
                        }
                        zero_tiles<HT>(db);
                        gemm_tiles<HT, 4 * HT>(sp + LAY.fh + l * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{ubs}, db);   // W_{l+2} ubar_l
                        exchange_hidden<HT>(xch, G::XCH_W, lane, wave, dln, ubs, Wh[l]);   // Wbar_{l+2} += delta_{l+1} ubar_l^T
                    }
                    f32x4 cb[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) { cb[mt] = db[mt] * d[L - 1][mt]; a2[L - 1][mt] += db[mt] * u[L - 1][mt]; }
                    {   // Wbar_N += eps_k cbar_k^T (wave w: column tile w);  Wbar_1 += delta_1^k [gbar_k; 0]^T (wave w: row block w)
                        tile_store(xmine + 0 * TS, lane, dense_tile<ZR>(eps));
                        tile_store(xmine + 1 * TS, lane, dense_tile<ZR>(gbar));
                        tiles_store<HT>(xmine + 2 * TS, lane, cb);
                        tiles_store<HT>(xmine + (2 + HT) * TS, lane, dl0);
                        __syncthreads();
                        if (HT >= 4 || wave < HT) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float* xv = xch + v * G::XCH_W;
                                float a1[4], b1[4], a2f[4], b2[4];
                                read_frag_A(xv + 0 * TS, lane, a1);
                                read_frag_B(xv + (2 + wave) * TS, lane, b1);
                                WNacc = outer4(a1, b1, WNacc);
                                read_frag_A(xv + (2 + HT + wave) * TS, lane, a2f);
                                read_frag_B(xv + 1 * TS, lane, b2);
                                W1acc[0] = outer4(a2f, b2, W1acc[0]);
                            }
                        }
                        __syncthreads();
                    }
                }

                // ---- top-down through the forward chain (once per stage) ----
                {   // Wbar_N += kbar h_L^T;  bbar_N += kbar x ones
                    tile_store(xmine + 0 * TS, lane, dense_tile<ZR>(kbar));
                    tiles_store<HT>(xmine + 1 * TS, lane, h[L - 1]);
                    __syncthreads();
                    if (HT >= 4 || wave < HT) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float* xv = xch + v * G::XCH_W;
                            float a1[4], b1[4];
                            read_frag_A(xv + 0 * TS, lane, a1);
                            read_frag_B(xv + (1 + wave) * TS, lane, b1);
                            WNacc = outer4(a1, b1, WNacc);
                            if (wave == 0) BNacc = outer4(a1, onesf, BNacc);
                        }
                    }
                    __syncthreads();
                }
                f32x4 hb[HT];
                zero_tiles<HT>(hb);
                gemm_tiles<HT, ZR>(sm + LAY.bN, lane, RegIn<ZR>{kbar}, hb);   // W_N^T kbar
                float Zbar[ZR];
#pragma unroll
                for (int l = L - 1; l >= 0; --l) {
                    f32x4 ab[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) {
                        const f32x4 d2 = ACT == CNF_ACT_TANH ? h[l][mt] * d[l][mt] * -2.f : d[l][mt] * (1.f - d[l][mt]);
                        ab[mt] = hb[mt] * d[l][mt] + a2[l][mt] * d2;
                    }
                    if (l > 0) {
                        // Wbar_{l+1} += abar_l h_{l-1}^T;  bbar_{l+1} += abar_l x ones
                        tiles_store<HT>(xmine + 0 * HT * TS, lane, ab);
                        tiles_store<HT>(xmine + 1 * HT * TS, lane, h[l - 1]);
                        __syncthreads();
                        if (HT >= 4 || wave < HT) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float* xv = xch + v * G::XCH_W;
                                float a1[4];
                                read_frag_A(xv + wave * TS, lane, a1);
                                Bh[l - 1] = outer4(a1, onesf, Bh[l - 1]);
#pragma unroll
                                for (int nt = 0; nt < HT; ++nt) {
                                    float b1[4];
                                    read_frag_B(xv + (HT + nt) * TS, lane, b1);
                                    Wh[l - 1][nt] = outer4(a1, b1, Wh[l - 1][nt]);
                                }
                            }
                        }
                        __syncthreads();
                        zero_tiles<HT>(hb);
                        gemm_tiles<HT, 4 * HT>(sm + LAY.bh + (l - 1) * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{ab}, hb);   // W_{l+1}^T abar
                    } else {
                        f32x4 in_tile;   // [z (D rows); t; ...; 1 at feature 15]
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = r < ZR ? zs[r < ZR ? r : 0] : 0.f;
                            if (!autonomous && 4 * r + g == D) v = tt;
                            if (4 * r + g > D || (autonomous && 4 * r + g == D)) v = 0.f;
                            if (4 * r + g == 15) v = 1.f;
                            in_tile[r] = v;
                        }
                        // Wbar_1 += abar_1 [z; t; 1]^T (+ abar_1 y^T)
                        tiles_store<HT>(xmine + 0 * HT * TS, lane, ab);
                        tile_store(xmine + (HT + 0) * TS, lane, in_tile);
                        if constexpr (CR > 0) tile_store(xmine + (HT + 1) * TS, lane, y_tile[0]);
                        __syncthreads();
                        if (HT >= 4 || wave < HT) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float* xv = xch + v * G::XCH_W;
                                float a1[4], b1[4];
                                read_frag_A(xv + wave * TS, lane, a1);
                                read_frag_B(xv + (HT + 0) * TS, lane, b1);
                                W1acc[0] = outer4(a1, b1, W1acc[0]);
                                if constexpr (CR > 0) {
                                    float by[4];
                                    read_frag_B(xv + (HT + 1) * TS, lane, by);
                                    W1acc[SL::NT1 - 1] = outer4(a1, by, W1acc[SL::NT1 - 1]);
                                }
                            }
                        }
                        __syncthreads();
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

struct ProbesInst { int HT, L, ZR, CR, ACT; GradKernel kern; };
#define GP_INST(HT, L, ZR, CR, ACT) ProbesInst { HT, L, ZR, CR, ACT, &mfma_grad_probes_kernel<HT, L, ZR, CR, ACT> }
#define GP_HT(HT, CR, ACT) GP_INST(HT, 3, 2, CR, ACT), GP_INST(HT, 2, 2, CR, ACT), GP_INST(HT, 3, 4, CR, ACT), GP_INST(HT, 2, 4, CR, ACT)
#define GP_SHAPES(CR, ACT) GP_HT(1, CR, ACT), GP_HT(2, CR, ACT), GP_HT(3, CR, ACT), GP_HT(4, CR, ACT)
static const ProbesInst kProbes[] = {GP_SHAPES(0, CNF_ACT_TANH), GP_SHAPES(0, CNF_ACT_SOFTPLUS),
                                     GP_SHAPES(4, CNF_ACT_TANH), GP_SHAPES(4, CNF_ACT_SOFTPLUS)};

GradKernel grad_probes_kernel(int HT, int L, int ZR, int CR, int ACT) {
    for (const ProbesInst& p : kProbes)
        if (p.HT == HT && p.L == L && p.ZR == ZR && p.CR == CR && p.ACT == ACT) return p.kern;
    return nullptr;
}

}  // namespace cnf
