// cnf_coop.hip — workgroup-cooperative fused solve kernel for wide hidden layers (gfx950).
//
// The per-wave kernel of cnf_mfma.hip keeps a 16-sample tile's whole activation set in one wave's
// registers and the whole operand image in LDS; that stops working around H = 128 (VJP) / 256
// (1.15 MB of images for cfg4: D = 32, 3x256).  Here a 256-thread workgroup (one wave per SIMD)
// owns a 64-sample super-tile for the whole solve:
//   * wave w computes output features [w H/4, (w+1) H/4) of every layer for all 4 sample tiles
//     (MTW x 4 accumulator tiles), and owns the ODE state / RK stage derivatives of sample tile w;
//   * act' of a wave's own features stays in its registers for the pullback (same partition);
//   * activations h_l and cotangents delta_l are exchanged through LDS.  Thanks to the row
//     permutation of cnf_mfma_layout.h, the f32x4 a lane holds for accumulator tile (mt, nt) is
//     exactly the B-operand fragment of k-group mt for sample tile nt, so the exchange is one
//     lane-linear ds_write_b128 per tile and one conflict-free ds_read_b128 per 4 MFMAs;
//   * weight fragments stream from L2 (the packed image stays L2-resident: 1.15 MB vs 4 MB per
//     XCD) with 16-byte-per-lane coalesced loads, each reused by 4 sample tiles (16 MFMAs per
//     load instruction), prefetched one k-group ahead.
// Same math and same reference map as cnf_mfma.hip (src/core/icnf.jl:517-559, utils.jl:150-159).
// With ONE wave per SIMD the packed-f32 asm statements (each carries its own wait states and is volatile) and the MFMA / VALU
// phase fences of the per-wave kernel cost more than they save here (same-box A/B at cfg4: 29.0 ms with both, 28.5 / 29.1 with
// one, 28.1 with neither): nothing else can issue while this wave sits in an s_nop, and the k-loops need the scheduler's freedom
// to place the L2 / LDS fragment loads.  This translation unit takes the plain forms.
#define CNF_NO_PK_ASM 1
#define CNF_NO_PHASE_FENCE 1
#include "cnf_coop_dev.h"
#include "cnf_coop_grad.h"

namespace cnf {

// One wave per SIMD means nothing hides a stall, so the structure below is about never waiting:
//   * the FIRST weight fragment of every product (an L2 read, 1-2 k cycles) is requested before the activation / exchange /
//     barrier phase that precedes the product, not after it (the A images do not depend on the barrier); biases likewise;
//   * no stage derivative is stored: running sums, as in the per-wave kernel; act' of the last hidden layer is rebuilt from the
//     activations still sitting in the exchange buffer instead of being held in 64 registers.
// (Tried and dropped: hoisting the solve-invariant products c = W_N^T eps and q = W_1[:,0:D] eps out of the RK loop, as the
// per-wave kernel does, through a per-workgroup global workspace - LDS and registers are full.  5.5 % fewer MFMAs, but 128 KB per
// workgroup per dynamics call does not stay in the 4 MB L2 of an XCD beside the weight image (PMC: 10 GB of HBM reads per launch
// when streamed), and because vmcnt retires in order the reads cannot hide under the product they are issued before: its own
// fragment loads wait for them.  cfg4: 25.7 ms without, 28.7 ms with c alone, 27.6 - 29.0 ms with both.)

// NT: sample tiles per super-tile (4: one workgroup per CU owns 64 samples; 2: two workgroups per CU own 32 samples each - half
// the exchange buffers and half the accumulators per wave, so two waves share each SIMD and fill each other's stalls, at the
// price of each weight fragment feeding 2 sample tiles instead of 4).  Waves 0 .. NT-1 own the ODE state of one sample tile.
// WL: the packed image sits in LDS at `wl` (see AImg); P is then only the global copy it was staged from
template <int HT, int L, int ZR, int ACT, int NT, bool WL = false, bool GOUT = false>
__device__ __forceinline__ void coop_eval(const float* __restrict__ P, const float* __restrict__ wl, f32x4* __restrict__ xbuf,
                                          f32x4* __restrict__ zbuf, const f32x4* __restrict__ ebuf, f32x4* __restrict__ pbuf,
                                          int lane, int wave, float t, bool autonomous, bool reg_z,
                                          bool reg_j, const float (&zs)[ZR],
                                          float (&zd)[ZR], float& ld, float& ed, float& nd, float* __restrict__ gout = nullptr, int KHa = 0,
                                          f32x4* __restrict__ fsb = nullptr, long long fsl = 0) {
    // fsb (checkpointing form, round 6): the stage store of the second-order reverse sweep (cnf_tiles.h) - every h_l and delta_l tile
    // of this stage leaves for HBM as the wave that computed it holds it (tile-native: one 16-byte store per lane); fsb points at
    // this super-tile's first tile of (kind h, layer 0), `fsl` f32x4 separate consecutive (kind, layer) arrays
    constexpr MfmaLayout LAY(HT, L, ZR, 0, true);
    // hidden k-groups that are not zero padding (even): a width that fills fewer 16-row tiles than the instance has skips the rest
    const int KH = (KHa > 0 && KHa < HT) ? ((KHa + 1) & ~1) : HT;
    constexpr int MTW = HT / 4, DT = (ZR + 3) / 4, XB = HT * NT * 64;   // XB: f32x4 per exchange buffer
    const bool owner = wave < NT;                                        // this wave integrates sample tile `wave`
    const int g = lane >> 4;
    const int mt0 = wave * MTW;
    // image offsets are multiples of 4 floats; fragment loads go through a buffer resource over the packed image
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    const float* __restrict__ PV = WL ? wl : P;   // bias / time-column vectors
#define AIMG(X) AImg{rP, (unsigned)(X) * 4u, lane16, WL ? wl : nullptr}
    f32x4 afr[MTW];                      // first A fragments of the next H-row product, requested one phase ahead
    f32x4 afd[DT];                       // ... of the next D-row product (last layer, W_1^T)
    // act' of this wave's features, kept for the pullback - except the LAST hidden layer's when it can be rebuilt from the
    // activations themselves, which stay in the exchange buffer until delta_{L-1} replaces them (tanh: 1 - h^2): 64 registers
    // NT == 1 (tile-split form): the two D-row products (zdot = W_N h_L and g = W_1[:,0:D]^T delta_1) are split over the waves
    // along K - wave w multiplies the k-groups of its OWN features straight from its registers and publishes a partial tile,
    // the owner adds the four partials in wave order - instead of the owner running all HT k-groups while three SIMDs wait
    constexpr bool SPLITK = NT == 1;
    constexpr bool D_FROM_H = (ACT == CNF_ACT_TANH_PRESCALED || ACT == CNF_ACT_TANH) && !SPLITK;
    constexpr int LD = D_FROM_H ? (L > 1 ? L - 1 : 1) : L;
    f32x4 d[LD][MTW][NT];
    f32x4 acc[MTW][NT];
    // ---- layer 1 ----
    {
        f32x4 bias[MTW], wt[MTW];
        coop_load_a<MTW>(AIMG(LAY.f1z), mt0, DT, 0, afr);
        gload_cvec<MTW>(PV + LAY.v_b1, mt0, g, bias);
        gload_cvec<MTW>(PV + LAY.v_w1t, mt0, g, wt);
        // publish this wave's stage state as the B image of sample tile `wave`
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
        phase_fence();
        coop_gemm<MTW, NT, NT>(AIMG(LAY.f1z), mt0, DT, zbuf, 0, lane, afr, acc);
        phase_fence();
    }
    // cur = exchange buffer holding the current layer's activations (compile-time after unrolling)
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int cur = l & 1;
        // request what the NEXT product needs before this layer's activation / exchange / barrier phase
        if (l + 1 < L) coop_load_a<MTW>(AIMG(LAY.fh + l * MfmaLayout::imgA(HT, HT)), mt0, HT, 0, afr);
        else if (SPLITK) coop_load_a<DT>(AIMG(LAY.fN), 0, HT, mt0, afd);
        else if (owner) coop_load_a<DT>(AIMG(LAY.fN), 0, HT, 0, afd);
        f32x4 hown[SPLITK ? MTW : 1];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                f32x4 h, dd;
                act_tile<ACT>(acc[m][q], h, dd);
                if (!(D_FROM_H && l == L - 1 && L > 1)) d[l < LD ? l : 0][m][q] = dd;
                if (SPLITK && l == L - 1) hown[SPLITK ? m : 0] = h;
                else xbuf[cur * XB + ((mt0 + m) * NT + q) * 64 + lane] = h;
                if constexpr (GOUT) {
                    if (fsb) __builtin_nontemporal_store(h, &fsb[(long long)l * fsl + (q * HT + mt0 + m) * 64 + lane]);
                }
            }
        if constexpr (SPLITK) {
            if (l == L - 1) {   // partial of zdot over this wave's own k-groups
                f32x4 zp[DT][1];
#pragma unroll
                for (int dt_ = 0; dt_ < DT; ++dt_) zp[dt_][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    if (m > 0) coop_load_a<DT>(AIMG(LAY.fN), 0, HT, mt0 + m, afd);
                    f32x4 hb[1] = {hown[m]};
                    coop_frag_mfma<DT, 1>(afd, hb, zp);
                }
#pragma unroll
                for (int dt_ = 0; dt_ < DT; ++dt_) pbuf[(wave * DT + dt_) * 64 + lane] = zp[dt_][0];
            }
        }
        if (l + 1 < L) {
            // the next layer's bias goes straight into the accumulators (dead now); the request overlaps the barrier
            f32x4 bnx[MTW];
            gload_cvec<MTW>(PV + LAY.v_bh + l * MfmaLayout::vecC(HT), mt0, g, bnx);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q) acc[m][q] = bnx[m];
        }
        __syncthreads();
        if (l + 1 < L) {
            phase_fence();
            if constexpr (WL) coop_gemm<MTW, NT, NT>(AIMG(LAY.fh + l * MfmaLayout::imgA(HT, HT)), mt0, HT, xbuf + cur * XB, 0, lane, afr, acc);   // (fragments from LDS: the guarded loop is the faster one)
            else coop_gemm_rt<MTW, NT, NT>(AIMG(LAY.fh + l * MfmaLayout::imgA(HT, HT)), mt0, HT, KH, xbuf + cur * XB, 0, lane, afr, acc);
            phase_fence();
        }
    }
    constexpr int hbuf = (L - 1) & 1;   // buffer holding h_L
    // ---- last layer (identity) for this wave's own sample tile: zdot ----
    if (owner) {
        f32x4 zacc[DT][1];
        f32x4 bias[DT];
        gload_cvec<DT>(PV + LAY.v_bN, 0, g, bias);
#pragma unroll
        for (int m = 0; m < DT; ++m) zacc[m][0] = bias[m];
        if constexpr (SPLITK) {
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int m = 0; m < DT; ++m) zacc[m][0] += pbuf[(w * DT + m) * 64 + lane];
        } else {
            phase_fence();
            if constexpr (WL) coop_gemm<DT, 1, NT>(AIMG(LAY.fN), 0, HT, xbuf + hbuf * XB, wave, lane, afd, zacc);   // (fragments from LDS: the guarded loop is the faster one)
            else coop_gemm_rt<DT, 1, NT>(AIMG(LAY.fN), 0, HT, KH, xbuf + hbuf * XB, wave, lane, afd, zacc);
            phase_fence();
        }
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = zacc[s >> 2][0][s & 3];
    }
    ed = 0.f;
    if (reg_z && owner) {
        float e2 = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) e2 = fmaf(zd[s], zd[s], e2);
        ed = sqrtf(group_sum(e2));
    }
    // ---- pullback: delta_L = c .* act'_L into the buffer h_{L-1} occupied, then W_l^T delta_l .* act'_{l-1} downwards ----
    nd = 0.f;
    {   // c = W_N^T eps
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        coop_load_a<MTW>(AIMG(LAY.bN), mt0, DT, 0, afr);
        coop_gemm<MTW, NT, NT>(AIMG(LAY.bN), mt0, DT, ebuf, 0, lane, afr, acc);
        if (L > 1) coop_load_a<MTW>(AIMG(LAY.bh + (L - 2) * MfmaLayout::imgA(HT, HT)), mt0, HT, 0, afr);
    }
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
        // acc holds W_{l+2}^T delta_{l+2} (or c): multiply by act'_{l+1}; publish it for the next product
        const int wbuf = ((L - 1 - l) & 1) ^ hbuf ^ 1;   // alternate, starting opposite to hbuf
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            f32x4 dm[NT], dl[NT];
            if (D_FROM_H && L > 1 && l == L - 1) {
                // act'_L = 1 - h_L^2 from this wave's own tiles of h_L, still in the exchange buffer
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    const f32x4 h = xbuf[hbuf * XB + ((mt0 + m) * NT + q) * 64 + lane];
                    const f32x2 h0 = {h[0], h[1]}, h1 = {h[2], h[3]}, one = {1.f, 1.f};
                    const f32x2 d0 = __builtin_elementwise_fma(-h0, h0, one), d1 = __builtin_elementwise_fma(-h1, h1, one);
                    dm[q] = f32x4{d0[0], d0[1], d1[0], d1[1]};
                }
            } else {
#pragma unroll
                for (int q = 0; q < NT; ++q) dm[q] = d[l < LD ? l : 0][m][q];
            }
            tiles_mul<NT>(acc[m], dm, dl);
            if constexpr (SPLITK) {
                if (l == 0) {   // partial of g = W_1[:,0:D]^T delta_1 over this wave's own k-groups (accumulated in acc[0][0..])
                    f32x4 gp[DT][1];
                    coop_load_a<DT>(AIMG(LAY.b1), 0, HT, mt0 + m, afd);
#pragma unroll
                    for (int dt_ = 0; dt_ < DT; ++dt_) gp[dt_][0] = m == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : pbuf[(wave * DT + dt_) * 64 + lane];
                    f32x4 db[1] = {dl[0]};
                    coop_frag_mfma<DT, 1>(afd, db, gp);
#pragma unroll
                    for (int dt_ = 0; dt_ < DT; ++dt_) pbuf[(wave * DT + dt_) * 64 + lane] = gp[dt_][0];
                    continue;
                }
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) xbuf[wbuf * XB + ((mt0 + m) * NT + q) * 64 + lane] = dl[q];
            if constexpr (GOUT) {
                if (fsb) {
#pragma unroll
                    for (int q = 0; q < NT; ++q) __builtin_nontemporal_store(dl[q], &fsb[(long long)(L + l) * fsl + (q * HT + mt0 + m) * 64 + lane]);
                }
            }
        }
        if (l == 0 && owner && !SPLITK) coop_load_a<DT>(AIMG(LAY.b1), 0, HT, 0, afd);
        __syncthreads();
        if (l > 0) {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            phase_fence();
            if constexpr (WL) coop_gemm<MTW, NT, NT>(AIMG(LAY.bh + (l - 1) * MfmaLayout::imgA(HT, HT)), mt0, HT, xbuf + wbuf * XB, 0, lane, afr, acc);   // (fragments from LDS: the guarded loop is the faster one)
            else coop_gemm_rt<MTW, NT, NT>(AIMG(LAY.bh + (l - 1) * MfmaLayout::imgA(HT, HT)), mt0, HT, KH, xbuf + wbuf * XB, 0, lane, afr, acc);
            phase_fence();
            if (l > 1) coop_load_a<MTW>(AIMG(LAY.bh + (l - 2) * MfmaLayout::imgA(HT, HT)), mt0, HT, 0, afr);
        } else if (owner) {
            // g = W_1[:,0:D]^T delta_1 for this wave's own sample tile (needed for |eps^T J|)
            f32x4 gacc[DT][1];
#pragma unroll
            for (int m = 0; m < DT; ++m) gacc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLITK) {
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int m = 0; m < DT; ++m) gacc[m][0] += pbuf[(w * DT + m) * 64 + lane];
            } else {
                phase_fence();
                if constexpr (WL) coop_gemm<DT, 1, NT>(AIMG(LAY.b1), 0, HT, xbuf + wbuf * XB, wave, lane, afd, gacc);   // (fragments from LDS: the guarded loop is the faster one)
                else coop_gemm_rt<DT, 1, NT>(AIMG(LAY.b1), 0, HT, KH, xbuf + wbuf * XB, wave, lane, afd, gacc);
                phase_fence();
            }
            // this lane's probe values sit in the B image of eps (k-group kg, own sample tile): no registers held for them
            float dot = 0.f, n2 = 0.f;
#pragma unroll
            for (int kg = 0; kg < DT; ++kg) {
                const f32x4 ev = ebuf[(kg * NT + wave) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (4 * kg + j < ZR) {
                        const float gv = gacc[kg][0][j];
                        dot = fmaf(gv, ev[j], dot);
                        n2 = fmaf(gv, gv, n2);
                    }
                }
            }
            ld = -group_sum(dot);
            nd = reg_j ? sqrtf(group_sum(n2)) : 0.f;
            if constexpr (GOUT) {   // g = eps^T J of this stage for the reverse sweep (cotangent of |eps^T J|)
                if (gout) {
#pragma unroll
                    for (int s = 0; s < ZR; ++s) gout[s] = gacc[s >> 2][0][s & 3];
                }
            }
        }
    }
}

// NS: Runge-Kutta stages of the instance (4: RK4, 6: Tsit5) - the partial sums of the later stages' increments take NS - 1 rows
// WL (with NT = 1): the tile-split form for small batches - one 16-sample tile per workgroup, its hidden width split over the
// four waves (one per SIMD), the operand images staged into LDS once per workgroup.  A batch of <= one tile per compute unit
// otherwise runs on one wave per tile at ~40 % MFMA utilisation with three SIMDs of every CU idle.
// CK: the owner waves also checkpoint z at the start of every step (+ the final state) and every stage derivative, in tile
// layout [..][16-sample tile][lane][ZR] (KArgs::ckpt / ckpt_k): the forward half of the cooperative gradient (cnf_coop_grad.hip)
template <int HT, int L, int ZR, int ACT, int NS, int NT, bool WL = false, bool CK = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, NT == 4 ? 1 : 2)))
coop_vjp_solve_kernel(KArgs a) {
    constexpr int DT = (ZR + 3) / 4, XB = HT * NT * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xbuf = reinterpret_cast<f32x4*>(smem);   // [2][HT][NT sample tiles][64 lanes]
    f32x4* zbuf = xbuf + 2 * XB;                     // [DT][NT][64]
    f32x4* ebuf = zbuf + DT * NT * 64;               // [DT][NT][64]
    f32x4* pbuf = ebuf + DT * NT * 64;               // NT == 1: [4 waves][DT][64] partial tiles of the D-row products
    constexpr int PB = NT == 1 ? 4 * DT * 64 : 0;
    const float* wl = nullptr;
    if constexpr (WL) {
        constexpr MfmaLayout LAYW(HT, L, ZR, 0, true);
        f32x4* dst = pbuf + PB;
        stage_image<256>(a.packed, reinterpret_cast<float*>(dst), LAYW.lds_total / 4);
        wl = reinterpret_cast<const float*>(dst);
        // (the first __syncthreads of the super-tile loop orders these writes before any fragment read)
    }
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = a.D, S = D + 3;
    const bool reg_z = a.reg_z, reg_j = a.reg_j, autonomous = a.autonomous;
    constexpr int SUP = 16 * NT;                      // samples per super-tile
    const long long nst = (a.B + SUP - 1) / SUP;
    const bool owner = wave < NT;

    for (long long st = blockIdx.x; st < nst; st += gridDim.x) {
        const long long smp = st * SUP + (owner ? wave : 0) * 16 + n;
        const bool valid = owner && smp < a.B;
        const long long sc = valid ? smp : a.B - 1;
        float z[ZR], eps[ZR];
        float lacc = 0.f, eacc = 0.f, nacc = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            if (a.x) z[s] = f < a.nvars ? a.x[sc * a.nvars + f] : 0.f;
            else z[s] = f < D ? a.u0[sc * S + f] : 0.f;
            eps[s] = f < D ? a.eps[sc * D + f] : 0.f;
        }
        if (!a.x) { lacc = a.u0[sc * S + D]; eacc = a.u0[sc * S + D + 1]; nacc = a.u0[sc * S + D + 2]; }
        __syncthreads();   // previous super-tile's readers of the exchange buffers are done
        if (owner) {
#pragma unroll
            for (int kg = 0; kg < DT; ++kg) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (4 * kg + j < ZR) ? eps[(4 * kg + j) < ZR ? 4 * kg + j : 0] : 0.f;
                ebuf[(kg * NT + wave) * 64 + lane] = v;
            }
        }

        // ---- fixed-step explicit RK, stage loop rolled; running sums instead of stored stage derivatives (see
        // mfma_solve_kernel in cnf_mfma_kernel.h: same fma chains in the same order) ----
        constexpr int NP = NS - 1;
        float Pz[NP][ZR], zsum[ZR], lsum, esum, nsum;
        float zd[ZR], ld = 0.f, ed = 0.f, nd = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = 0.f;
        const float dt = a.dt;
        const bool single = a.nsteps == 0;
        const int ns = single ? 1 : (a.T.ns < NS ? a.T.ns : NS);
        const int nsteps = single ? 1 : a.nsteps;
#pragma clang loop unroll(disable)
        for (int step = 0; step < nsteps; ++step) {
            const float tn = a.t0 + (float)step * dt;
            if constexpr (CK) {
                if (owner) {
#pragma unroll
                    for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)step * nst * NT + st * NT + wave) * 64 + lane) * ZR + s] = z[s];
                }
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
                float* gout = nullptr;
                if constexpr (CK) {
                    if (a.ckpt_g) gout = a.ckpt_g + ((((long long)step * ns + sg) * nst * NT + st * NT + wave) * 64 + lane) * ZR;
                }
                f32x4* fsb = nullptr;
                long long fsl = 0;
                if constexpr (CK) {
                    if (a.kfull) {   // (the checkpointing form's use of KArgs::kfull: base of the stage store, cnf_tiles.h)
                        fsl = (long long)nsteps * ns * nst * NT * HT * 64;
                        fsb = reinterpret_cast<f32x4*>(a.kfull) + ((((long long)step * ns + sg) * nst + st) * NT) * HT * 64;
                    }
                }
                coop_eval<HT, L, ZR, ACT, NT, WL, CK>(a.packed, wl, xbuf, zbuf, ebuf, pbuf, lane, wave, tn + a.T.c[sg] * dt, autonomous,
                                          reg_z, reg_j, zs, zd, ld, ed, nd, gout, a.KH, fsb, fsl);
                if constexpr (CK) {
                    if (owner) {
#pragma unroll
                        for (int s = 0; s < ZR; ++s)
                            a.ckpt_k[((((long long)step * ns + sg) * nst * NT + st * NT + wave) * 64 + lane) * ZR + s] = zd[s];
                    }
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
        if (single) {
            if (valid) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) a.u_out[smp * S + f] = zd[s]; }
                if (g == 0) { a.u_out[smp * S + D] = ld; a.u_out[smp * S + D + 1] = ed; a.u_out[smp * S + D + 2] = nd; }
            }
            continue;
        }
        if constexpr (CK) {
            if (owner) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) a.ckpt[(((long long)a.nsteps * nst * NT + st * NT + wave) * 64 + lane) * ZR + s] = z[s];
            }
        }
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
template <int HT, int L, int ZR, int ACT, int NS, int NT, bool WL = false, bool CK = false>
static hipError_t launch_coop(const KArgs& a, int nblocks, hipStream_t st) {
    constexpr int DT = (ZR + 3) / 4;
    constexpr int lds = (2 * HT * NT * 64 + 2 * DT * NT * 64 + (NT == 1 ? 4 * DT * 64 : 0)) * 16 +
                        (WL ? MfmaLayout(HT, L, ZR, 0, true).lds_total * 4 : 0);
    static_assert(lds <= 160 * 1024, "exchange buffers + staged image exceed LDS");
    auto kern = coop_vjp_solve_kernel<HT, L, ZR, ACT, NS, NT, WL, CK>;
    static DeviceOnce once;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    if (!once.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        once.set(dev);
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

struct CoopInst {
    int HT, L, ZR, ACT;
    // [0] RK4, [1] Tsit5, 64-sample super-tiles (one workgroup per CU).  (32-sample super-tiles - two workgroups per CU, two waves per
    // SIMD - were measured at cfg4: 28.8 ms against 25.4 ms; each weight fragment then feeds 2 sample tiles instead of 4 and the second
    // wave on the SIMD hides less than that costs.  The A/B build was deleted in round 6.)
    hipError_t (*fn[2])(const KArgs&, int, hipStream_t);
};
#define COOP_INST(HT, L, ZR, ACT) \
    CoopInst { HT, L, ZR, ACT, { &launch_coop<HT, L, ZR, ACT, 4, 4>, &launch_coop<HT, L, ZR, ACT, 6, 4> } }
// tanh instances are compiled for pre-scaled pre-activations (mfma_pack folds -2 log2 e into the
// forward images); they are matched against CNF_ACT_TANH configurations.  First the exact shapes,
// then zero-padded ones (state k-steps padded to 8: D <= 32).
#define COOP_GEN(HT)                                                                              \
    COOP_INST(HT, 3, 8, CNF_ACT_TANH_PRESCALED), COOP_INST(HT, 2, 8, CNF_ACT_TANH_PRESCALED),     \
    COOP_INST(HT, 3, 8, CNF_ACT_SOFTPLUS), COOP_INST(HT, 2, 8, CNF_ACT_SOFTPLUS)
static const CoopInst kCoop[] = {
    COOP_INST(16, 3, 8, CNF_ACT_TANH_PRESCALED),   // cfg4: D=32, 3x256
    COOP_INST(8, 3, 2, CNF_ACT_TANH_PRESCALED),    // D=8, 3x128 Hutchinson VJP
    COOP_INST(4, 3, 2, CNF_ACT_TANH_PRESCALED),    // D=8, 3x64 (cross-check of the per-wave kernel)
    COOP_GEN(16), COOP_GEN(12), COOP_GEN(8),
};

// smallest instance that holds the shape: hidden tiles (zero-padded, e.g. 7 -> 8), then state k-steps
static const CoopInst* coop_find(int HT, int L, int ZR, int ACT) {
    const CoopInst* best = nullptr;
    for (const CoopInst& c : kCoop) {
        const bool act_ok = c.ACT == ACT || (c.ACT == CNF_ACT_TANH_PRESCALED && ACT == CNF_ACT_TANH);
        if (c.HT >= HT && c.L == L && c.ZR >= ZR && act_ok &&
            (!best || c.HT < best->HT || (c.HT == best->HT && c.ZR < best->ZR)))
            best = &c;
    }
    return best;
}

// ---- the tile-split small-batch form: exact shapes only (it shares the per-wave instance's packed image) ----
struct SplitInst {
    int HT, L, ZR, ACT;
    hipError_t (*fn[2])(const KArgs&, int, hipStream_t);   // [0] <= 4 stages (RK4), [1] <= 6 stages (Tsit5)
};
#define SPLIT_INST(HT, L, ZR, ACT) \
    SplitInst { HT, L, ZR, ACT, { &launch_coop<HT, L, ZR, ACT, 4, 1, true>, &launch_coop<HT, L, ZR, ACT, 6, 1, true> } }
static const SplitInst kSplit[] = {
    SPLIT_INST(4, 3, 2, CNF_ACT_TANH_PRESCALED),   // D <= 8, 3x64 tanh (cfg2 / cfg2' at small batches)
    SPLIT_INST(4, 2, 2, CNF_ACT_TANH_PRESCALED),
    SPLIT_INST(4, 3, 2, CNF_ACT_SOFTPLUS),
    SPLIT_INST(4, 2, 2, CNF_ACT_SOFTPLUS),
    // the generic zero-padded per-wave instances (cnf_mfma_generic.hip) pad the state to 4 k-steps: D <= 16
    SPLIT_INST(4, 3, 4, CNF_ACT_TANH_PRESCALED), SPLIT_INST(4, 2, 4, CNF_ACT_TANH_PRESCALED),
    SPLIT_INST(4, 3, 4, CNF_ACT_SOFTPLUS), SPLIT_INST(4, 2, 4, CNF_ACT_SOFTPLUS),
};
static const SplitInst* split_find(int HT, int L, int ZR, int ACT) {
    for (const SplitInst& c : kSplit)
        if (c.HT == HT && c.L == L && c.ZR == ZR && (c.ACT == ACT || (c.ACT == CNF_ACT_TANH_PRESCALED && ACT == CNF_ACT_TANH))) return &c;
    return nullptr;
}
bool coop_split_supported(int HT, int L, int ZR, int ACT) { return split_find(HT, L, ZR, ACT) != nullptr; }
hipError_t coop_split_launch(int HT, int L, int ZR, int ACT, const KArgs& a, hipStream_t st) {
    const SplitInst* c = split_find(HT, L, ZR, ACT);
    if (!c) return hipErrorNotSupported;
    const long long ntiles = (a.B + 15) / 16;
    return c->fn[a.T.ns <= 4 ? 0 : 1](a, (int)ntiles, st);
}

// ---- the checkpointing form: the forward half of the cooperative gradient, for the shapes cnf_coop_grad.hip instantiates ----
struct CkInst {
    int HT, L, ZR;
    hipError_t (*fn[2])(const KArgs&, int, hipStream_t);
};
#define CK_INST(HT, L, ZR) \
    CkInst { HT, L, ZR, { &launch_coop<HT, L, ZR, CNF_ACT_TANH_PRESCALED, 4, 4, false, true>, \
                          &launch_coop<HT, L, ZR, CNF_ACT_TANH_PRESCALED, 6, 4, false, true> } }
static const CkInst kCk[] = {CK_INST(16, 3, 8), CK_INST(8, 3, 2), CK_INST(4, 3, 2)};
bool coop_ckpt_supported(int HT, int L, int ZR, int ACT) {
    if (ACT != CNF_ACT_TANH && ACT != CNF_ACT_TANH_PRESCALED) return false;
    for (const CkInst& c : kCk)
        if (c.HT == HT && c.L == L && c.ZR == ZR) return true;
    return false;
}
hipError_t coop_launch_ckpt(int HT, int L, int ZR, int ACT, const KArgs& a, int num_cus, hipStream_t st) {
    if (ACT != CNF_ACT_TANH && ACT != CNF_ACT_TANH_PRESCALED) return hipErrorNotSupported;
    for (const CkInst& c : kCk)
        if (c.HT == HT && c.L == L && c.ZR == ZR) {
            const long long nst = (a.B + 63) / 64;
            const int nblocks = (int)(nst < num_cus ? nst : num_cus);
            return c.fn[a.T.ns <= 4 ? 0 : 1](a, nblocks, st);
        }
    return hipErrorNotSupported;
}

bool coop_supported(int HT, int L, int ZR, int CR, int ACT, int engine, int KP, int* ZR_inst, int* HT_inst) {
    if (engine != ENG_VJP || KP != 1 || CR != 0) return false;
    const CoopInst* c = coop_find(HT, L, ZR, ACT);
    if (c && ZR_inst) *ZR_inst = c->ZR;
    if (c && HT_inst) *HT_inst = c->HT;
    return c != nullptr;
}

hipError_t coop_launch(int HT, int L, int ZR, int ACT, const KArgs& a, int num_cus, hipStream_t st) {
    const CoopInst* c = coop_find(HT, L, ZR, ACT);
    if (!c) return hipErrorNotSupported;
    const long long nst = (a.B + 63) / 64;
    const int nblocks = (int)(nst < num_cus ? nst : num_cus);
    return c->fn[a.T.ns <= 4 ? 0 : 1](a, nblocks, st);
}

}  // namespace cnf
