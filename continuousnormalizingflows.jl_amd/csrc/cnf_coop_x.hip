// cnf_coop_x.hip — the cooperative wide-layer solve kernel extended to what cnf_coop.hip's instances leave out:
//   * conditioned flows (CondLayer rows [z; t; ys], src/layers/cond_layer.jl:7-31, src/core/base_icnf.jl:272-296),
//   * several Hutchinson probes (this framework's nprobes extension; one forward pass, K pullbacks),
//   * the exact trace (TestMode, src/core/icnf.jl:297-339, src/core/utils.jl:79-88) as the D unit probes e_1 .. e_D through the
//     same pullback: tr J = sum_p (e_p^T J)_p - what the reference's DI variants do with one-hot seeds (utils.jl:35-77),
//   * Hutchinson JVP (LuxJacVecMatrixMode, src/core/icnf.jl:561-603): the probes pushed through the forward images,
// for hidden widths above the per-wave kernels' reach (129 .. 256: their forward + transposed images exceed LDS).  Before this
// file those configurations ran layer-wise (cnf_layered.hip), every product a separate launch with its operands in HBM.
//
// Same organisation as cnf_coop.hip (a workgroup owns a super-tile for the whole solve, wave w the output features
// [w H/4, (w+1) H/4) of every product, exchange through LDS as B images, weight fragments from the L2-resident packed image),
// with 32-sample super-tiles and two workgroups per CU (two waves per SIMD, 256 registers each): act' of EVERY hidden layer
// has to stay in registers across the probes, and at 16 sample tiles per wave that set alone would be 192 registers.
// The tuned single-probe kernel of cnf_coop.hip is left untouched (its schedule is sensitive to its own source text).
#define CNF_NO_PK_ASM 1
#define CNF_NO_PHASE_FENCE 1
#include "cnf_coop_dev.h"

namespace cnf {

// One dynamics evaluation for a 32-sample super-tile.  K probes (or, with `exact`, the D unit vectors) run one after the
// other through the pullback; eps_p of the owner's sample tile is published per probe.
template <int HT, int L, int ZR, int CR, int ACT>
__device__ __forceinline__ void coopx_eval(const float* __restrict__ P, f32x4* xbuf, f32x4* zbuf, f32x4* ebuf, const f32x4* ybuf,
                                           int lane, int wave, float t, bool autonomous, bool reg_z, bool reg_j, bool exact, bool jvp,
                                           int D, int K, const float* __restrict__ eps_col, const float (&zs)[ZR], float (&zd)[ZR],
                                           float& ld, float& ed, float& nd, float* __restrict__ gout = nullptr, int q_off = 0, int KHa = 0) {
    constexpr int NT = 2;
    const int KH = (KHa > 0 && KHa < HT) ? ((KHa + 1) & ~1) : HT;   // hidden k-groups that are not zero padding (even: coop_gemm_rt)
    constexpr MfmaLayout LAY(HT, L, ZR, CR, true);
    constexpr int MTW = HT / 4, DT = (ZR + 3) / 4, KGC = (CR + 3) / 4, XB = HT * NT * 64;
    constexpr int IMG = MfmaLayout::imgA(HT, HT);
    const bool owner = wave < NT;
    const int g = lane >> 4;
    const int mt0 = wave * MTW;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
#define AIMG(X) AImg{rP, (unsigned)(X) * 4u, lane16, nullptr}
    f32x4 afr[MTW], afd[DT];
    f32x4 d[L][MTW][NT];     // act' of every hidden layer: needed by every probe
    f32x4 acc[MTW][NT];
    // ---- layer 1: a = W1z z + w1t t + W1y y + b1 ----
    {
        f32x4 bias[MTW], wt[MTW];
        coop_load_a<MTW>(AIMG(LAY.f1z), mt0, DT, 0, afr);
        gload_cvec<MTW>(P + LAY.v_b1, mt0, g, bias);
        gload_cvec<MTW>(P + LAY.v_w1t, mt0, g, wt);
        if (owner) {
#pragma unroll
            for (int kg = 0; kg < DT; ++kg) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (4 * kg + j < ZR) ? zs[(4 * kg + j) < ZR ? 4 * kg + j : 0] : 0.f;
                zbuf[(kg * NT + wave) * 64 + lane] = v;
            }
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const f32x4 b0 = autonomous ? bias[m] : tile_fma(wt[m], t, bias[m]);
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[m][q] = b0;
        }
        __syncthreads();
        coop_gemm<MTW, NT, NT>(AIMG(LAY.f1z), mt0, DT, zbuf, 0, lane, afr, acc);
        if constexpr (CR > 0) {
            coop_load_a<MTW>(AIMG(LAY.f1y), mt0, KGC, 0, afr);
            coop_gemm<MTW, NT, NT>(AIMG(LAY.f1y), mt0, KGC, ybuf, 0, lane, afr, acc);
        }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int cur = l & 1;
        if (l + 1 < L) coop_load_a<MTW>(AIMG(LAY.fh + l * IMG), mt0, HT, 0, afr);
        else if (owner) coop_load_a<DT>(AIMG(LAY.fN), 0, HT, 0, afd);
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                f32x4 h;
                act_tile<ACT>(acc[m][q], h, d[l][m][q]);
                xbuf[cur * XB + ((mt0 + m) * NT + q) * 64 + lane] = h;
            }
        if (l + 1 < L) {
            f32x4 bnx[MTW];
            gload_cvec<MTW>(P + LAY.v_bh + l * MfmaLayout::vecC(HT), mt0, g, bnx);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q) acc[m][q] = bnx[m];
        }
        __syncthreads();
        if (l + 1 < L) coop_gemm_rt<MTW, NT, NT>(AIMG(LAY.fh + l * IMG), mt0, HT, KH, xbuf + cur * XB, 0, lane, afr, acc);
    }
    constexpr int hbuf = (L - 1) & 1;   // buffer holding h_L
    if (owner) {   // zdot = W_N h_L + b_N for this wave's own sample tile
        f32x4 zacc[DT][1], bias[DT];
        gload_cvec<DT>(P + LAY.v_bN, 0, g, bias);
#pragma unroll
        for (int m = 0; m < DT; ++m) zacc[m][0] = bias[m];
        coop_gemm_rt<DT, 1, NT>(AIMG(LAY.fN), 0, HT, KH, xbuf + hbuf * XB, wave, lane, afd, zacc);
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = zacc[s >> 2][0][s & 3];
    }
    ed = 0.f;
    if (reg_z && owner) {
        float e2 = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) e2 = fmaf(zd[s], zd[s], e2);
        ed = sqrtf(group_sum(e2));   // Edot = |zdot|_2   (src/core/icnf.jl:184-199)
    }
    ld = 0.f;
    nd = 0.f;
    int nseed = exact ? D : K;
    if constexpr (L == 2) {
        if (exact && q_off > 0) {
            // Two hidden layers: tr J = sum_ab act'_2[a] W_2[a][b] act'_1[b] (W_1[:,0:D] W_3)[b][a] = act'_2^T Q act'_1 with the constant
            // Q = W_2 .* (W_1[:,0:D] W_3)^T packed behind the operand images: ONE H x H product and a dot instead of D pullbacks (the
            // batched-Jacobian trace of src/core/utils.jl:79-88, icnf.jl:312, for the reference's default architecture).
            // act'_1 goes out as a B image through buffer 0 (h_1's: every reader passed the barrier behind h_2's publish).
            coop_load_a<MTW>(AIMG(q_off), mt0, HT, 0, afr);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    xbuf[((mt0 + m) * NT + q) * 64 + lane] = d[0][m][q];
                    acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            __syncthreads();
            coop_gemm_rt<MTW, NT, NT>(AIMG(q_off), mt0, HT, KH, xbuf, 0, lane, afr, acc);
            // this wave's rows of Q act'_1 against its rows of act'_2; the four waves' partial traces meet in the probe image's LDS
            float* red = reinterpret_cast<float*>(ebuf);
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                float tr = 0.f;
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tr = fmaf(acc[m][q][r], d[1][m][q][r], tr);
                red[(wave * NT + q) * 64 + lane] = group_sum(tr);
            }
            __syncthreads();
            if (owner) ld = -(red[(0 * NT + wave) * 64 + lane] + red[(1 * NT + wave) * 64 + lane] + red[(2 * NT + wave) * 64 + lane] +
                              red[(3 * NT + wave) * 64 + lane]);
            nseed = 0;
        }
    }
    const float scale = exact ? 1.f : 1.f / (float)K;
#pragma clang loop unroll(disable)
    for (int p = 0; p < nseed; ++p) {
        // the probe of this wave's own sample tile as a B image; the barrier also retires every reader of the exchange buffers
        // (the previous probe's g product, or the zdot product)
        float ep[ZR];
        if (owner) {
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                const int f = 4 * s + g;
                ep[s] = exact ? (f == p ? 1.f : 0.f) : (f < D ? eps_col[p * D + f] : 0.f);
            }
#pragma unroll
            for (int kg = 0; kg < DT; ++kg) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (4 * kg + j < ZR) ? ep[(4 * kg + j) < ZR ? 4 * kg + j : 0] : 0.f;
                ebuf[(kg * NT + wave) * 64 + lane] = v;
            }
        }
        if (jvp) {
            // Hutchinson JVP (LuxJacVecMatrixMode, src/core/icnf.jl:561-603, utils.jl:161-170): the tangent eps_p pushed through
            // the FORWARD images - tau_1 = act'_1 .* (W_1[:,0:D] eps), tau_{l+1} = act'_{l+1} .* (W_{l+1} tau_l), J eps = W_N tau_L;
            // ldot = -<eps, J eps> / K (the scalar the VJP form gives), ndot = |J eps|_2 / K.  The forward hidden-layer images
            // carry the tanh pre-scale: undone per product.
            constexpr float inv_fs = ACT == CNF_ACT_TANH_PRESCALED ? 1.f / kTanhPrescale : 1.f;
            coop_load_a<MTW>(AIMG(LAY.f1z), mt0, DT, 0, afr);
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            coop_gemm<MTW, NT, NT>(AIMG(LAY.f1z), mt0, DT, ebuf, 0, lane, afr, acc);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int wbuf = (l & 1) ^ hbuf ^ 1;   // alternate, starting opposite to the buffer h_L sits in
                if (l + 1 < L) coop_load_a<MTW>(AIMG(LAY.fh + l * IMG), mt0, HT, 0, afr);
                else if (owner) coop_load_a<DT>(AIMG(LAY.fN), 0, HT, 0, afd);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q) xbuf[wbuf * XB + ((mt0 + m) * NT + q) * 64 + lane] = acc[m][q] * d[l][m][q] * inv_fs;
                __syncthreads();
                if (l + 1 < L) {
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    coop_gemm_rt<MTW, NT, NT>(AIMG(LAY.fh + l * IMG), mt0, HT, KH, xbuf + wbuf * XB, 0, lane, afr, acc);
                } else if (owner) {
                    f32x4 gacc[DT][1];   // J eps for this wave's own sample tile
#pragma unroll
                    for (int m = 0; m < DT; ++m) gacc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    coop_gemm_rt<DT, 1, NT>(AIMG(LAY.fN), 0, HT, KH, xbuf + wbuf * XB, wave, lane, afd, gacc);
                    float dot = 0.f, n2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) {
                        const float gv = gacc[s >> 2][0][s & 3];
                        dot = fmaf(gv, ep[s], dot);
                        n2 = fmaf(gv, gv, n2);
                    }
                    ld -= scale * group_sum(dot);
                    if (reg_j) nd += scale * sqrtf(group_sum(n2));   // ndot = |J eps|_2 (icnf.jl:229-245 on the JacVec twin)
                }
            }
            continue;
        }
        coop_load_a<MTW>(AIMG(LAY.bN), mt0, DT, 0, afr);
        __syncthreads();
        {   // c = W_N^T eps_p
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            coop_gemm<MTW, NT, NT>(AIMG(LAY.bN), mt0, DT, ebuf, 0, lane, afr, acc);
            if (L > 1) coop_load_a<MTW>(AIMG(LAY.bh + (L - 2) * IMG), mt0, HT, 0, afr);
        }
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            const int wbuf = ((L - 1 - l) & 1) ^ hbuf ^ 1;
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                f32x4 dl[NT];
                tiles_mul<NT>(acc[m], d[l][m], dl);
#pragma unroll
                for (int q = 0; q < NT; ++q) xbuf[wbuf * XB + ((mt0 + m) * NT + q) * 64 + lane] = dl[q];
            }
            if (l == 0 && owner) coop_load_a<DT>(AIMG(LAY.b1), 0, HT, 0, afd);
            __syncthreads();
            if (l > 0) {
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                coop_gemm_rt<MTW, NT, NT>(AIMG(LAY.bh + (l - 1) * IMG), mt0, HT, KH, xbuf + wbuf * XB, 0, lane, afr, acc);
                if (l > 1) coop_load_a<MTW>(AIMG(LAY.bh + (l - 2) * IMG), mt0, HT, 0, afr);
            } else if (owner) {
                f32x4 gacc[DT][1];   // g = W_1[:,0:D]^T delta_1 = eps_p^T J for this wave's own sample tile
#pragma unroll
                for (int m = 0; m < DT; ++m) gacc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                coop_gemm_rt<DT, 1, NT>(AIMG(LAY.b1), 0, HT, KH, xbuf + wbuf * XB, wave, lane, afd, gacc);
                float dot = 0.f, n2 = 0.f;
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    const float gv = gacc[s >> 2][0][s & 3];
                    dot = fmaf(gv, ep[s], dot);       // <eps^T J, eps>, or J_pp for a unit probe
                    n2 = fmaf(gv, gv, n2);
                }
                ld -= scale * group_sum(dot);
                if (reg_j) nd += scale * sqrtf(group_sum(n2));   // ndot = |eps^T J|_2 (icnf.jl:229-245)
                if (gout) {   // checkpointing solve (one probe): g = eps^T J of this stage for the reverse sweep
#pragma unroll
                    for (int s = 0; s < ZR; ++s) gout[s] = gacc[s >> 2][0][s & 3];
                }
            }
        }
    }
#undef AIMG
}

constexpr int coopx_lds_bytes(int HT, int ZR, int CR) { return (2 * HT * 2 * 64 + 2 * ((ZR + 3) / 4) * 2 * 64 + ((CR + 3) / 4) * 2 * 64) * 16; }
// (two workgroups per CU, two waves per SIMD at 256 registers, where two sets of exchange buffers fit the LDS; otherwise - 16
// hidden tiles x 16 state k-steps - one workgroup with the whole register file)
template <int HT, int L, int ZR, int CR, int ACT, int NS>
__global__ void __launch_bounds__(256)
    __attribute__((amdgpu_waves_per_eu(2 * coopx_lds_bytes(HT, ZR, CR) <= 160 * 1024 ? 2 : 1, 2 * coopx_lds_bytes(HT, ZR, CR) <= 160 * 1024 ? 2 : 1)))
coopx_solve_kernel(KArgs a) {
    constexpr int NT = 2;
    constexpr int DT = (ZR + 3) / 4, KGC = (CR + 3) / 4, XB = HT * NT * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xbuf = reinterpret_cast<f32x4*>(smem);   // [2][HT][NT][64]
    f32x4* zbuf = xbuf + 2 * XB;                     // [DT][NT][64]
    f32x4* ebuf = zbuf + DT * NT * 64;               // [DT][NT][64]
    f32x4* ybuf = ebuf + DT * NT * 64;               // [KGC][NT][64]: conditions, constant over the solve
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = a.D, S = D + 3, C = a.C;
    const bool exact = a.exact == 1, jvp = a.exact == 2;   // (KArgs::exact = 2: Hutchinson JVP, set for this kernel family only)
    const int K = exact ? 1 : a.K;
    const int Kd = K * D;
    const bool reg_z = a.reg_z, reg_j = a.reg_j, autonomous = a.autonomous;
    constexpr int SUP = 16 * NT;
    const long long nst = (a.B + SUP - 1) / SUP;
    const bool owner = wave < NT;

    for (long long st = blockIdx.x; st < nst; st += gridDim.x) {
        const long long smp = st * SUP + (owner ? wave : 0) * 16 + n;
        const bool valid = owner && smp < a.B;
        const long long sc = smp < a.B ? smp : a.B - 1;
        float z[ZR];
        float lacc = 0.f, eacc = 0.f, nacc = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            if (a.x) z[s] = f < a.nvars ? a.x[sc * a.nvars + f] : 0.f;
            else z[s] = f < D ? a.u0[sc * S + f] : 0.f;
        }
        if (!a.x) { lacc = a.u0[sc * S + D]; eacc = a.u0[sc * S + D + 1]; nacc = a.u0[sc * S + D + 2]; }
        const float* eps_col = a.eps ? a.eps + sc * Kd : nullptr;     // this lane's column of probes (read per probe; L2-hot)
        __syncthreads();   // the previous super-tile's readers of the LDS images are done
        if constexpr (CR > 0) {
            if (owner) {
#pragma unroll
                for (int kg = 0; kg < KGC; ++kg) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = 4 * (4 * kg + j) + g;
                        v[j] = (4 * kg + j < CR && f < C) ? a.ys[sc * C + f] : 0.f;
                    }
                    ybuf[(kg * NT + wave) * 64 + lane] = v;
                }
            }
        }

        constexpr int NP = NS - 1;
        float Pz[NP][ZR], zsum[ZR], lsum, esum, nsum;
        float zd[ZR], ld = 0.f, ed = 0.f, nd = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = 0.f;
        const float dt0 = a.dt;
        const bool single = a.nsteps == 0;
        const int ns = single ? 1 : (a.T.ns < NS ? a.T.ns : NS);
        const int nsteps = single ? 1 : a.nsteps;
        // checkpoints for the cooperative gradient (KArgs::ckpt / ckpt_k / ckpt_g; tile = 16-sample tile index, nst NT of them)
        const long long cktile = st * NT + (owner ? wave : 0), ckntp = nst * NT;
#pragma clang loop unroll(disable)
        for (int step = 0; step < nsteps; ++step) {
            // uniform steps, or the caller's grid (KArgs::tgrid: the frozen steps of an adaptive solve, for the gradient)
            const float tn = a.tgrid ? a.tgrid[step] : a.t0 + (float)step * dt0;
            const float dt = a.tgrid ? a.tgrid[step + 1] - tn : dt0;
            if (a.ckpt && owner && !single) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)step * ckntp + cktile) * 64 + lane) * ZR + s] = z[s];
            }
            lsum = esum = nsum = 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                zsum[s] = 0.f;
#pragma unroll
                for (int i = 0; i < NP; ++i) Pz[i][s] = 0.f;
            }
#pragma clang loop unroll(disable)
            for (int sg = 0; sg < ns; ++sg) {
                float zs[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) zs[s] = fmaf(dt, Pz[0][s], z[s]);
                float* gout = (a.ckpt_g && owner && !single) ? a.ckpt_g + ((((long long)step * ns + sg) * ckntp + cktile) * 64 + lane) * ZR : nullptr;
                coopx_eval<HT, L, ZR, CR, ACT>(a.packed, xbuf, zbuf, ebuf, ybuf, lane, wave, tn + a.T.c[sg] * dt, autonomous, reg_z, reg_j,
                                                exact, jvp, D, K, eps_col, zs, zd, ld, ed, nd, gout, a.q_off, a.KH);
                if (a.ckpt_k && owner && !single) {
#pragma unroll
                    for (int s = 0; s < ZR; ++s) a.ckpt_k[((((long long)step * ns + sg) * ckntp + cktile) * 64 + lane) * ZR + s] = zd[s];
                }
                const float bst = a.T.b[sg];
                lsum = fmaf(bst, ld, lsum); esum = fmaf(bst, ed, esum); nsum = fmaf(bst, nd, nsum);
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    zsum[s] = fmaf(bst, zd[s], zsum[s]);
#pragma unroll
                    for (int i = 0; i < NP - 1; ++i) Pz[i][s] = fmaf(a.acol[sg][i], zd[s], Pz[i + 1][s]);
                    Pz[NP - 1][s] = a.acol[sg][NP - 1] * zd[s];
                }
            }
            if (single) break;
            lacc = fmaf(dt, lsum, lacc); eacc = fmaf(dt, esum, eacc); nacc = fmaf(dt, nsum, nacc);
#pragma unroll
            for (int s = 0; s < ZR; ++s) z[s] = fmaf(dt, zsum[s], z[s]);
        }
        if (a.ckpt && owner && !single) {
#pragma unroll
            for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)nsteps * ckntp + cktile) * 64 + lane) * ZR + s] = z[s];
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
// host side
// ---------------------------------------------------------------------------------------
template <int HT, int L, int ZR, int CR, int ACT, int NS>
static hipError_t launch_coopx(const KArgs& a, int num_cus, hipStream_t st) {
    constexpr int lds = coopx_lds_bytes(HT, ZR, CR);
    static_assert(lds <= 160 * 1024, "exchange buffers exceed LDS");
    auto kern = coopx_solve_kernel<HT, L, ZR, CR, ACT, NS>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (!once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    const long long nst = (a.B + 31) / 32;
    const long long cap = 2LL * num_cus;
    const int nblocks = (int)(nst < cap ? nst : cap);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

struct CoopXInst {
    int HT, L, ZR, CR, ACT;
    hipError_t (*fn[2])(const KArgs&, int, hipStream_t);   // [0] <= 4 stages (RK4), [1] <= 6 stages (Tsit5)
};
#define CX_INST(HT, L, ZR, CR, ACT) \
    CoopXInst { HT, L, ZR, CR, ACT, { &launch_coopx<HT, L, ZR, CR, ACT, 4>, &launch_coopx<HT, L, ZR, CR, ACT, 6> } }
// zero-padded instances: hidden tiles 8 / 12 / 16 (H <= 128 / 192 / 256), 8 or 16 state k-steps (D <= 32 / 64 - the reference's
// default architecture has D = 2 nvariables + 1 and H = 4 (D + 1): nvariables 16 .. 30 land here), 0 or 4 condition k-steps
// (C <= 16); tanh instances run pre-scaled pre-activations (mfma_pack folds -2 log2 e into the forward images)
#define CX_SHAPES(HT, ACT) CX_INST(HT, 3, 8, 0, ACT), CX_INST(HT, 2, 8, 0, ACT), CX_INST(HT, 3, 8, 4, ACT), CX_INST(HT, 2, 8, 4, ACT), \
                           CX_INST(HT, 3, 16, 0, ACT), CX_INST(HT, 2, 16, 0, ACT), CX_INST(HT, 3, 16, 4, ACT), CX_INST(HT, 2, 16, 4, ACT)
// beyond 256 hidden units / 64 state rows: 20 or 24 hidden tiles (H <= 320 / 384) x 24 state k-steps (D <= 96), unconditioned; one
// workgroup per CU (120 KB of exchange buffers), one wave per SIMD with the whole register file
#define CX_BIG(HT, ACT) CX_INST(HT, 3, 24, 0, ACT), CX_INST(HT, 2, 24, 0, ACT)
static const CoopXInst kCoopX[] = {
    // exact twins of cnf_coop.hip's two narrow-state instances: their plans checkpoint on a caller's grid (the frozen steps of an
    // adaptive solve) through this kernel ON THEIR OWN packed image, so hidden tiles and state k-steps must coincide
    CX_INST(8, 3, 2, 0, CNF_ACT_TANH_PRESCALED), CX_INST(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED),
    CX_SHAPES(8, CNF_ACT_TANH_PRESCALED), CX_SHAPES(12, CNF_ACT_TANH_PRESCALED), CX_SHAPES(16, CNF_ACT_TANH_PRESCALED),
    CX_SHAPES(8, CNF_ACT_SOFTPLUS), CX_SHAPES(12, CNF_ACT_SOFTPLUS), CX_SHAPES(16, CNF_ACT_SOFTPLUS),
    CX_BIG(20, CNF_ACT_TANH_PRESCALED), CX_BIG(24, CNF_ACT_TANH_PRESCALED), CX_BIG(20, CNF_ACT_SOFTPLUS), CX_BIG(24, CNF_ACT_SOFTPLUS),
};

static const CoopXInst* cx_find(int HT, int L, int ZR, int CR, int ACT) {
    const CoopXInst* best = nullptr;
    for (const CoopXInst& c : kCoopX) {
        const bool act_ok = c.ACT == ACT || (c.ACT == CNF_ACT_TANH_PRESCALED && ACT == CNF_ACT_TANH);
        if (c.HT >= HT && c.L == L && c.ZR >= ZR && c.CR >= CR && (CR > 0 || c.CR == 0) && act_ok &&
            (!best || c.HT < best->HT || (c.HT == best->HT && c.ZR < best->ZR)))
            best = &c;
    }
    return best;
}
// the instance compiled for exactly this layout family MfmaLayout(HT, L, ZR, CR, true): the only kind that may run on an image
// another plan packed (the kernel reads the image through that layout's offsets and strides its checkpoints by ZR)
static const CoopXInst* cx_find_exact(int HT, int L, int ZR, int CR, int ACT) {
    for (const CoopXInst& c : kCoopX) {
        const bool act_ok = c.ACT == ACT || (c.ACT == CNF_ACT_TANH_PRESCALED && ACT == CNF_ACT_TANH);
        if (c.HT == HT && c.L == L && c.ZR == ZR && c.CR == CR && act_ok) return &c;
    }
    return nullptr;
}

bool coopx_supported(int HT, int L, int ZR, int CR, int ACT, int* HT_inst, int* ZR_inst, int* CR_inst) {
    const CoopXInst* c = cx_find(HT, L, ZR, CR, ACT);
    if (!c) return false;
    if (HT_inst) *HT_inst = c->HT;
    if (ZR_inst) *ZR_inst = c->ZR;
    if (CR_inst) *CR_inst = c->CR;
    return true;
}

hipError_t coopx_launch(int HT, int L, int ZR, int CR, int ACT, const KArgs& a, int num_cus, hipStream_t st) {
    const CoopXInst* c = cx_find(HT, L, ZR, CR, ACT);
    if (!c) return hipErrorNotSupported;
    return c->fn[a.T.ns <= 4 ? 0 : 1](a, num_cus, st);
}

bool coopx_exact_supported(int HT, int L, int ZR, int CR, int ACT) { return cx_find_exact(HT, L, ZR, CR, ACT) != nullptr; }
hipError_t coopx_launch_exact(int HT, int L, int ZR, int CR, int ACT, const KArgs& a, int num_cus, hipStream_t st) {
    const CoopXInst* c = cx_find_exact(HT, L, ZR, CR, ACT);
    if (!c) return hipErrorNotSupported;
    return c->fn[a.T.ns <= 4 ? 0 : 1](a, num_cus, st);
}

}  // namespace cnf
