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
#include "cnf_mfma_dev.h"

namespace cnf {

// acc[m][q] += A(global image; M-tile mt0+m) * B(LDS image; sample tile nt0+q), over KG k-groups.
// Two fragment sets ping-pong (k-loop unrolled by 2): the loads of k-group kg+1 — A from L2, B from
// LDS — are issued before the 16 M NQ MFMAs of k-group kg, so both latencies hide behind them and
// no register copies are needed.
template <int M, int NQ>
__device__ __forceinline__ void coop_frag_load(const f32x4* __restrict__ A, int mt0, int KG, int kg,
                                               const f32x4* __restrict__ bimg, int nt0, int lane,
                                               f32x4 (&a)[M], f32x4 (&b)[NQ]) {
#pragma unroll
    for (int m = 0; m < M; ++m) a[m] = A[((mt0 + m) * KG + kg) * 64];
#pragma unroll
    for (int q = 0; q < NQ; ++q) b[q] = bimg[(kg * 4 + nt0 + q) * 64 + lane];
}

template <int M, int NQ>
__device__ __forceinline__ void coop_frag_mfma(const f32x4 (&a)[M], const f32x4 (&b)[NQ], f32x4 (&acc)[M][NQ]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[m][q] = mfma4(a[m][j], b[q][j], acc[m][q]);
}

template <int M, int NQ>
__device__ __forceinline__ void coop_gemm(const float* __restrict__ gimg, int mt0, int KG,
                                          const f32x4* __restrict__ bimg, int nt0, int lane,
                                          f32x4 (&acc)[M][NQ]) {
    const f32x4* __restrict__ A = reinterpret_cast<const f32x4*>(gimg) + lane;
    f32x4 a0[M], a1[M], b0[NQ], b1[NQ];
    coop_frag_load<M, NQ>(A, mt0, KG, 0, bimg, nt0, lane, a0, b0);
#pragma clang loop unroll(disable)
    for (int kg = 0; kg < KG; kg += 2) {
        const bool has1 = kg + 1 < KG, has2 = kg + 2 < KG;   // wave-uniform
        if (has1) coop_frag_load<M, NQ>(A, mt0, KG, kg + 1, bimg, nt0, lane, a1, b1);
        coop_frag_mfma<M, NQ>(a0, b0, acc);
        if (has1) {
            if (has2) coop_frag_load<M, NQ>(A, mt0, KG, kg + 2, bimg, nt0, lane, a0, b0);
            coop_frag_mfma<M, NQ>(a1, b1, acc);
        }
    }
}

template <int MT>
__device__ __forceinline__ void gload_cvec(const float* __restrict__ vec, int mt0, int g, f32x4 (&out)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) out[m] = *reinterpret_cast<const f32x4*>(vec + ((mt0 + m) * 4 + g) * 4);
}

template <int HT, int L, int ZR, int ACT>
__device__ __forceinline__ void coop_eval(const float* __restrict__ P, f32x4* __restrict__ xbuf,
                                          f32x4* __restrict__ zbuf, const f32x4* __restrict__ ebuf,
                                          int lane, int wave, float t, bool autonomous, bool reg_z,
                                          bool reg_j, const float (&zs)[ZR], const float (&eps)[ZR],
                                          float (&zd)[ZR], float& ld, float& ed, float& nd) {
    constexpr MfmaLayout LAY(HT, L, ZR, 0, true);
    constexpr int MTW = HT / 4, DT = (ZR + 3) / 4, XB = HT * 4 * 64;   // XB: f32x4 per exchange buffer
    const int g = lane >> 4;
    const int mt0 = wave * MTW;
    // publish this wave's stage state as the B image of sample tile `wave`
#pragma unroll
    for (int kg = 0; kg < DT; ++kg) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (4 * kg + j < ZR) ? zs[(4 * kg + j) < ZR ? 4 * kg + j : 0] : 0.f;
        zbuf[(kg * 4 + wave) * 64 + lane] = v;
    }
    __syncthreads();

    f32x4 d[L][MTW][4];
    f32x4 acc[MTW][4];
    // ---- layer 1 ----
    {
        f32x4 bias[MTW], wt[MTW];
        gload_cvec<MTW>(P + LAY.v_b1, mt0, g, bias);
        gload_cvec<MTW>(P + LAY.v_w1t, mt0, g, wt);
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[m][q] = autonomous ? bias[m] : bias[m] + wt[m] * t;
        coop_gemm<MTW, 4>(P + LAY.f1z, mt0, DT, zbuf, 0, lane, acc);
    }
    // cur = exchange buffer holding the current layer's activations (compile-time after unrolling)
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int cur = l & 1;
        if (l > 0) {
            f32x4 bias[MTW];
            gload_cvec<MTW>(P + LAY.v_bh + (l - 1) * MfmaLayout::vecC(HT), mt0, g, bias);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[m][q] = bias[m];
            coop_gemm<MTW, 4>(P + LAY.fh + (l - 1) * MfmaLayout::imgA(HT, HT), mt0, HT, xbuf + (cur ^ 1) * XB, 0,
                              lane, acc);
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float dd;
                    h[r] = act_fwd<ACT>(acc[m][q][r], dd);
                    d[l][m][q][r] = dd;
                }
                xbuf[cur * XB + ((mt0 + m) * 4 + q) * 64 + lane] = h;
            }
        __syncthreads();
    }
    constexpr int hbuf = (L - 1) & 1;   // buffer holding h_L
    // ---- last layer (identity) for this wave's own sample tile: zdot ----
    {
        f32x4 zacc[DT][1];
        f32x4 bias[DT];
        gload_cvec<DT>(P + LAY.v_bN, 0, g, bias);
#pragma unroll
        for (int m = 0; m < DT; ++m) zacc[m][0] = bias[m];
        coop_gemm<DT, 1>(P + LAY.fN, 0, HT, xbuf + hbuf * XB, wave, lane, zacc);
#pragma unroll
        for (int s = 0; s < ZR; ++s) zd[s] = zacc[s >> 2][0][s & 3];
    }
    ed = 0.f;
    if (reg_z) {
        float e2 = 0.f;
#pragma unroll
        for (int s = 0; s < ZR; ++s) e2 = fmaf(zd[s], zd[s], e2);
        ed = sqrtf(group_sum(e2));
    }
    // ---- pullback: delta_L = (W_N^T eps) .* act'_L, written to the buffer h_{L-1} occupied ----
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    coop_gemm<MTW, 4>(P + LAY.bN, mt0, DT, ebuf, 0, lane, acc);
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
        // acc holds W_{l+2}^T delta_{l+2} (or W_N^T eps): multiply by act'_{l+1}, publish, next product
        const int wbuf = ((L - 1 - l) & 1) ^ hbuf ^ 1;   // alternate, starting opposite to hbuf
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) xbuf[wbuf * XB + ((mt0 + m) * 4 + q) * 64 + lane] = acc[m][q] * d[l][m][q];
        __syncthreads();
        if (l > 0) {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            coop_gemm<MTW, 4>(P + LAY.bh + (l - 1) * MfmaLayout::imgA(HT, HT), mt0, HT, xbuf + wbuf * XB, 0, lane, acc);
        } else {
            // g = W_1[:,0:D]^T delta_1 for this wave's own sample tile
            f32x4 gacc[DT][1];
#pragma unroll
            for (int m = 0; m < DT; ++m) gacc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            coop_gemm<DT, 1>(P + LAY.b1, 0, HT, xbuf + wbuf * XB, wave, lane, gacc);
            float dot = 0.f, n2 = 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                const float gv = gacc[s >> 2][0][s & 3];
                dot = fmaf(gv, eps[s], dot);
                n2 = fmaf(gv, gv, n2);
            }
            ld = -group_sum(dot);
            nd = reg_j ? sqrtf(group_sum(n2)) : 0.f;
        }
    }
}

// NS: Runge-Kutta stage derivatives kept (4 for RK4, 6 for Tsit5)
template <int HT, int L, int ZR, int ACT, int NS>
__global__ void __launch_bounds__(256)
coop_vjp_solve_kernel(KArgs a) {
    constexpr int DT = (ZR + 3) / 4, XB = HT * 4 * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xbuf = reinterpret_cast<f32x4*>(smem);   // [2][HT][4 sample tiles][64 lanes]
    f32x4* zbuf = xbuf + 2 * XB;                     // [DT][4][64]
    f32x4* ebuf = zbuf + DT * 4 * 64;                // [DT][4][64]
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = a.D, S = D + 3;
    const bool reg_z = a.reg_z, reg_j = a.reg_j, autonomous = a.autonomous;
    const long long nst = (a.B + 63) / 64;

    for (long long st = blockIdx.x; st < nst; st += gridDim.x) {
        const long long smp = st * 64 + wave * 16 + n;
        const bool valid = smp < a.B;
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
        __syncthreads();   // previous super-tile's readers of ebuf are done
#pragma unroll
        for (int kg = 0; kg < DT; ++kg) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (4 * kg + j < ZR) ? eps[(4 * kg + j) < ZR ? 4 * kg + j : 0] : 0.f;
            ebuf[(kg * 4 + wave) * 64 + lane] = v;
        }

        float kz[NS][ZR], kl[NS], ke[NS], kn[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            kl[j] = ke[j] = kn[j] = 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) kz[j][s] = 0.f;
        }
        const float dt = a.dt;
        const bool single = a.nsteps == 0;
        const int ns = single ? 1 : (a.T.ns < NS ? a.T.ns : NS);
        const int nsteps = single ? 1 : a.nsteps;
#pragma clang loop unroll(disable)
        for (int step = 0; step < nsteps; ++step) {
            const float tn = a.t0 + (float)step * dt;
#pragma clang loop unroll(disable)
            for (int sg = 0; sg < ns; ++sg) {
                float zs[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < NS - 1; ++j) acc = fmaf(a.T.a[sg][j], kz[j][s], acc);
                    zs[s] = fmaf(dt, acc, z[s]);
                }
                float zd[ZR], ld, ed, nd;
                coop_eval<HT, L, ZR, ACT>(a.packed, xbuf, zbuf, ebuf, lane, wave, tn + a.T.c[sg] * dt, autonomous,
                                          reg_z, reg_j, zs, eps, zd, ld, ed, nd);
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const bool hit = (j == sg);
                    kl[j] = hit ? ld : kl[j];
                    ke[j] = hit ? ed : ke[j];
                    kn[j] = hit ? nd : kn[j];
#pragma unroll
                    for (int s = 0; s < ZR; ++s) kz[j][s] = hit ? zd[s] : kz[j][s];
                }
            }
            if (single) break;
            float sl = 0.f, se = 0.f, sn = 0.f;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const float bj = a.T.b[j];
                sl = fmaf(bj, kl[j], sl); se = fmaf(bj, ke[j], se); sn = fmaf(bj, kn[j], sn);
            }
            lacc = fmaf(dt, sl, lacc); eacc = fmaf(dt, se, eacc); nacc = fmaf(dt, sn, nacc);
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < NS; ++j) acc = fmaf(a.T.b[j], kz[j][s], acc);
                z[s] = fmaf(dt, acc, z[s]);
            }
        }
        if (single) {
            if (valid) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) a.u_out[smp * S + f] = kz[0][s]; }
                if (g == 0) { a.u_out[smp * S + D] = kl[0]; a.u_out[smp * S + D + 1] = ke[0]; a.u_out[smp * S + D + 2] = kn[0]; }
            }
            continue;
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
template <int HT, int L, int ZR, int ACT, int NS>
static hipError_t launch_coop(const KArgs& a, int nblocks, hipStream_t st) {
    constexpr int DT = (ZR + 3) / 4;
    constexpr int lds = (2 * HT * 4 * 64 + 2 * DT * 4 * 64) * 16;
    auto kern = coop_vjp_solve_kernel<HT, L, ZR, ACT, NS>;
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
    hipError_t (*fn4)(const KArgs&, int, hipStream_t);   // RK4 (4 stage derivatives kept)
    hipError_t (*fn6)(const KArgs&, int, hipStream_t);   // Tsit5
};
#define COOP_INST(HT, L, ZR, ACT) \
    CoopInst { HT, L, ZR, ACT, &launch_coop<HT, L, ZR, ACT, 4>, &launch_coop<HT, L, ZR, ACT, 6> }
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
    return (a.T.ns <= 4 ? c->fn4 : c->fn6)(a, nblocks, st);
}

}  // namespace cnf
