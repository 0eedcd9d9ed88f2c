// cnf_coop_dgrad.hip — the reverse sweep of the cooperative gradient in the dealt form (round 4): the flows whose forward solve
// runs on the dealt kernels of cnf_coop_d.hip (8 .. 15 hidden tiles, two hidden layers: the reference's default architecture at
// nvariables = 16 .. 29, src/core/icnf.jl:53-103) differentiate through this kernel instead of cnf_coop_grad.hip's.
//
// Same mathematics, same arguments (CGArgs), same operand arrays for the deferred weight-cotangent products and the same two-chain
// products as cnf_coop_grad.hip (read that header first; reference: Zygote through SciMLBase.solve, src/core/icnf.jl:90-99,
// src/exts/mlj_ext/core_icnf.jl:42-51).  What changes is who multiplies what.  cnf_coop_grad.hip gives wave w the hidden tiles
// [w HT / 4, (w + 1) HT / 4) of an instance whose tile and state counts are multiples of four and eight: at nvariables = 16
// (H = 136: 8.5 tiles, D = 33: 8.25 k-steps) it runs a (12 tiles, 16 k-steps) instance and issues 1.94 x the MFMAs of the
// arithmetic (profiles/r4/r4h_nv16_sweep_pmc.txt), with one wave multiplying nothing but padding.  Here
//   * a workgroup owns a 32-sample super-tile: a product has four column tiles (two chains x two sample tiles), and the real
//     hidden tiles HT = 4 A + b are DEALT: wave w takes tiles [w A, (w + 1) A) with all four columns, the 2 b left-over
//     (tile, sample) units - both chains of a unit stay in one wave, the elementwise phases couple them - go one each to the
//     waves in order;
//   * k-loops run the real k-steps (last k-group `rem` of 4), state rows come in whole M-tiles of the configuration (KZ), not
//     of the plan's layout;
//   * Zbar = W_1[:,0:D]^T sbar_1 is split along K by OWNERSHIP: every wave multiplies the sbar_1 tiles it has just produced,
//     from registers, and the owners of the two sample tiles add the partial tiles;
//   * act'_1 and dbar_1 wait in accumulation registers between the way up and the way down (cnf_coop_grad.hip multiplies
//     dbar_1 = W_1[:,0:D] gbar a second time instead);
//   * an exchange buffer leaves for HBM (the operand arrays of lg_wgrad) AFTER the barrier that publishes it, column w of every
//     tile pair by wave w, as full 128-byte lines - the store partition does not follow tile ownership.
// Six barriers per stage.  LDS: two exchange buffers [HT][4][64 lanes] (the [z | gbar] image aliases the second one, the partial
// tiles of Zbar the first), the [eps | kbar] image, the C vectors.
#define CNF_NO_PK_ASM 1
#define CNF_NO_PHASE_FENCE 1
#include "cnf_coop_d_dev.h"
#include "cnf_coop_grad.h"

#ifndef CG_TEMPORAL
#define DG_NT_AUX 2   // raw buffer store: nt (the operand arrays stream past the L2, see cnf_coop_grad.hip)
#else
#define DG_NT_AUX 0
#endif

// The kernel's barriers order LDS traffic only (the exchange images); no wave reads global memory another wave of the launch
// wrote (__syncthreads() would also wait for every outstanding global store).
#define DG_SYNC()                                                       \
    do {                                                                \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
        __builtin_amdgcn_s_barrier();                                   \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
    } while (0)

#ifdef DG_TRACE
#define DG_T(k) do { asm volatile("" ::: "memory"); tr[k] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#else
#define DG_T(k)
#endif

namespace cnf {

struct DGArgs {
    CGArgs c;
    DImg g;
};

namespace {

// act''(a) from h = act(a) and d = act'(a):  tanh: -2 h d;  softplus: d (1 - d)
template <int ACT>
__device__ __forceinline__ f32x4 dg_act_dd(const f32x4& h, const f32x4& d) {
    if constexpr (ACT == CNF_ACT_SOFTPLUS) return d * (1.f - d);
    else return h * d * -2.f;
}

// NSAMP: sample tiles of a super-tile (2: 32 samples, four column tiles per product; 1: 16 samples, two - the form for 16 .. 24
// hidden tiles, whose four-column exchange buffers would not fit the LDS)
template <int A, int NSAMP>
struct GAcc {
    f32x4 S[A][2 * NSAMP];   // tiles [w A, (w + 1) A) x (chain 0: the sample tiles | chain 1: the sample tiles)
    f32x4 R[2][2];           // this wave's left-over (tile, sample) units (see gunits): [slot][chain]
};
template <int A, int NSAMP>
struct GHalf {               // one chain's worth of this wave's units
    f32x4 S[A][NSAMP];
    f32x4 R[2];
};
template <int A>
struct GOff { unsigned S[A]; unsigned Rr[2]; };

// this wave's left-over units.  They are dealt from the LAST wave down (unit u = 3 - w and u + 4 of the 2 b (tile, sample) units:
// tile 4 A + (u >> 1) + 2 s, sample tile u & 1): waves 0 and 1 own the sample tiles - the dense phase that opens every stage is
// theirs while the other two wait - so the extra products go to waves 3 and 2 first.  Slot s is live when u + 4 s < 2 b; the tile
// is clamped for the loads.
struct GUnits {
    bool v0, v1;
    int t0, t1, q;
};
template <int NSAMP>
__device__ __forceinline__ GUnits gunits(int A, int b, int wave) {
    GUnits u;
    const int un = 3 - wave;
    const int tmax = 4 * A + b - 1;
    if constexpr (NSAMP == 2) {
        u.v0 = un < 2 * b; u.v1 = un + 4 < 2 * b;
        const int r0 = 4 * A + (un >> 1), r1 = r0 + 2;
        u.t0 = r0 < tmax ? r0 : tmax; u.t1 = r1 < tmax ? r1 : tmax;
        u.q = un & 1;
    } else {   // one sample tile: the b left-over tiles are the units, one each to waves 3, 2, 1
        u.v0 = un < b; u.v1 = false;
        const int r0 = 4 * A + un;
        u.t0 = r0 < tmax ? r0 : tmax; u.t1 = tmax;
        u.q = 0;
    }
    return u;
}
template <int A>
__device__ __forceinline__ GOff<A> g_offsets(const DRs& R, int KP, int mtS0, const GUnits& U) {
    GOff<A> t;
#pragma unroll
    for (int m = 0; m < A; ++m) { t.S[m] = R.lane16 + (unsigned)((mtS0 + m) * KP) * 1024u; asm volatile("" : "+v"(t.S[m])); }
    t.Rr[0] = R.lane16 + (unsigned)(U.t0 * KP) * 1024u; asm volatile("" : "+v"(t.Rr[0]));
    t.Rr[1] = R.lane16 + (unsigned)(U.t1 * KP) * 1024u; asm volatile("" : "+v"(t.Rr[1]));
    return t;
}
template <int A>
__device__ __forceinline__ void g_load_a(const DRs& R, const GOff<A>& T, unsigned img, int kg, f32x4 (&aS)[A], f32x4 (&aR)[2]) {
    const unsigned so = img + (unsigned)kg * 1024u;
#pragma unroll
    for (int m = 0; m < A; ++m) aS[m] = dloadv(R, T.S[m], so);
    aR[0] = dloadv(R, T.Rr[0], so);
    aR[1] = dloadv(R, T.Rr[1], so);
}
// B fragments of k-group kg: the four column tiles, and again the two columns of this wave's left-over sample tile (the column
// index is a scalar the compiler must not fold into per-column code paths)
template <int NSAMP>
__device__ __forceinline__ void g_load_b(const f32x4* __restrict__ bimg, int kg, int qr, int lane, f32x4 (&bq)[2 * NSAMP], f32x4 (&bo)[2]) {
    constexpr int NC = 2 * NSAMP;
#pragma unroll
    for (int c = 0; c < NC; ++c) bq[c] = bimg[(kg * NC + c) * 64 + lane];
    int qv = qr;
    asm volatile("" : "+s"(qv));
    bo[0] = bimg[(kg * NC + qv) * 64 + lane];
    bo[1] = bimg[(kg * NC + NSAMP + qv) * 64 + lane];
}
template <int A, int NSAMP, int JN>
__device__ __forceinline__ void g_mfma(const f32x4 (&aS)[A], const f32x4 (&aR)[2], const f32x4 (&bq)[2 * NSAMP], const f32x4 (&bo)[2], const GUnits& U,
                                       GAcc<A, NSAMP>& u) {
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < 2 * NSAMP; ++c) u.S[m][c] = mfma4(aS[m][j], bq[c][j], u.S[m][c]);
    if (U.v0) {
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) u.R[0][ch] = mfma4(aR[0][j], bo[ch][j], u.R[0][ch]);
    }
    if (U.v1) {
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) u.R[1][ch] = mfma4(aR[1][j], bo[ch][j], u.R[1][ch]);
    }
}
template <int A, int NSAMP>
__device__ __forceinline__ void g_mfma_rem(const f32x4 (&aS)[A], const f32x4 (&aR)[2], const f32x4 (&bq)[2 * NSAMP], const f32x4 (&bo)[2], const GUnits& U,
                                           int rem, GAcc<A, NSAMP>& u) {
    if (rem == 4) g_mfma<A, NSAMP, 4>(aS, aR, bq, bo, U, u);
    else if (rem == 3) g_mfma<A, NSAMP, 3>(aS, aR, bq, bo, U, u);
    else if (rem == 2) g_mfma<A, NSAMP, 2>(aS, aR, bq, bo, U, u);
    else g_mfma<A, NSAMP, 1>(aS, aR, bq, bo, U, u);
}
// u += A(image) * B(LDS image, four column tiles) over KG k-groups, the last one with `rem` k-steps; aS0 / aR0 arrive holding the
// fragments of k-group 0 (structure of dealt_gemm, cnf_coop_d.hip: two fragment sets ping-pong, one k-group of lead - a third set
// and two k-groups of lead were measured with s_memtime: no difference)
template <int A, int NSAMP>
__device__ __forceinline__ void g_gemm(const DRs& R, const GOff<A>& T, unsigned img, int KG, int rem, const GUnits& U,
                                       const f32x4* __restrict__ bimg, int lane, f32x4 (&aS0)[A], f32x4 (&aR0)[2], GAcc<A, NSAMP>& u) {
    f32x4 aS1[A], aR1[2], bq0[2 * NSAMP], bq1[2 * NSAMP], bo0[2], bo1[2];
    g_load_b<NSAMP>(bimg, 0, U.q, lane, bq0, bo0);
    const int KGf = KG - 1;
    int kg = 0;
#pragma clang loop unroll(disable)
    for (; kg + 2 <= KGf; kg += 2) {
        g_load_a<A>(R, T, img, kg + 1, aS1, aR1);
        g_load_b<NSAMP>(bimg, kg + 1, U.q, lane, bq1, bo1);
        g_mfma<A, NSAMP, 4>(aS0, aR0, bq0, bo0, U, u);
        g_load_a<A>(R, T, img, kg + 2, aS0, aR0);
        g_load_b<NSAMP>(bimg, kg + 2, U.q, lane, bq0, bo0);
        g_mfma<A, NSAMP, 4>(aS1, aR1, bq1, bo1, U, u);
    }
    if (kg < KGf) {
        g_load_a<A>(R, T, img, KG - 1, aS1, aR1);
        g_load_b<NSAMP>(bimg, KG - 1, U.q, lane, bq1, bo1);
        g_mfma<A, NSAMP, 4>(aS0, aR0, bq0, bo0, U, u);
        g_mfma_rem<A, NSAMP>(aS1, aR1, bq1, bo1, U, rem, u);
    } else {
        g_mfma_rem<A, NSAMP>(aS0, aR0, bq0, bo0, U, rem, u);
    }
}

// K-split D-row product by ownership, one chain: part[dm][q] = partial over this wave's shared k-groups for sample tile q,
// own[dm] = partial over its left-over units' k-groups (sample tile w & 1).  f0 arrives holding the fragments of k-group kgS0.
template <int A, int NSAMP, int DT>
__device__ __forceinline__ void g_drow(const DRs& R, const unsigned (&vd)[DT], unsigned img, int kgS0, int KG, int rem, const GUnits& U,
                                       const GHalf<A, NSAMP>& x, f32x4 (&f0)[DT], f32x4 (&part)[DT][NSAMP], f32x4 (&own)[DT]) {
    f32x4 f1[DT];
#pragma unroll
    for (int dm = 0; dm < DT; ++dm) {
        own[dm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NSAMP; ++q) part[dm][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int m = 0; m < A + 2; ++m) {
        f32x4(&cur)[DT] = (m & 1) ? f1 : f0;
        f32x4(&nxt)[DT] = (m & 1) ? f0 : f1;
        if (m + 1 < A + 2) {
            const int kgn = m + 1 < A ? kgS0 + m + 1 : (m + 1 == A ? U.t0 : U.t1);
#pragma unroll
            for (int dm = 0; dm < DT; ++dm) nxt[dm] = dloadv(R, vd[dm], img + (unsigned)kgn * 1024u);
        }
        if (m < A) {
            const bool last = kgS0 + m == KG - 1;
            if (!last || rem == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                        for (int q = 0; q < NSAMP; ++q) part[dm][q] = mfma4(cur[dm][j], x.S[m][q][j], part[dm][q]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < rem) {
#pragma unroll
                        for (int dm = 0; dm < DT; ++dm)
#pragma unroll
                            for (int q = 0; q < NSAMP; ++q) part[dm][q] = mfma4(cur[dm][j], x.S[m][q][j], part[dm][q]);
                    }
            }
        } else {
            const int s = m - A;
            if (s == 0 ? U.v0 : U.v1) {
                // (a left-over tile is the last k-group when b > 0: its k-steps beyond `rem` multiply zero columns of the image)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dm = 0; dm < DT; ++dm) own[dm] = mfma4(cur[dm][j], x.R[s][j], own[dm]);
            }
        }
    }
}

}  // namespace

// two exchange buffers + the [eps | kbar] image (+ the partial tiles when they do not alias the first exchange buffer) + C vectors
constexpr int coopd_grad_lds_bytes(int HT, int DT, int cvn, int NSAMP = 2, bool palias = true) {
    return (2 * ((HT + 1) / 2 * 2) * 2 * NSAMP * 64 + DT * 2 * NSAMP * 64 + (palias ? 0 : (NSAMP + 1) * 4 * DT * 64)) * 16 + (cvn + 3) / 4 * 16;
}

template <int A, int KZ, int ACT, int NS, int NSAMP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
coopd_grad_step_kernel(DGArgs da) {
    constexpr int NC = 2 * NSAMP, SUP = 16 * NSAMP;
    static_assert(ACT == CNF_ACT_TANH_PRESCALED || ACT == CNF_ACT_SOFTPLUS, "act'' is rebuilt from h and act': tanh and softplus");
    static_assert(KZ % 4 == 0, "state registers in whole M-tiles");
    const CGArgs& a = da.c;
    const DImg& G = da.g;
    constexpr int DT = KZ / 4;
    constexpr bool KEEP_H = ACT != CNF_ACT_SOFTPLUS;   // tanh's act'' needs h as well
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HT = 4 * A + G.b, HTE = (HT + 1) & ~1;
    f32x4* X0 = reinterpret_cast<f32x4*>(smem);        // [HTE][NC][64]: exchange buffer; the partial tiles of Zbar alias it (G.xalias)
    f32x4* X1 = X0 + HTE * NC * 64;                    // [HTE][NC][64]: exchange buffer; the [z | gbar] image aliases it
    f32x4* ekbuf = X1 + HTE * NC * 64;                 // [DT][NC][64]: [eps | kbar]
    f32x4* zebuf = X1;
    f32x4* pbuf = G.xalias ? X0 : ekbuf + DT * NC * 64;   // [NSAMP + 1 slots][4 waves][DT][64]
    float* cbuf = reinterpret_cast<float*>(ekbuf + DT * NC * 64 + (G.xalias ? 0 : (NSAMP + 1) * 4 * DT * 64));
    for (int i = threadIdx.x; i < G.cvn; i += 256) cbuf[i] = a.packed[G.v_b1 + i];
    const float* __restrict__ P = cbuf - G.v_b1;
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool owner = wave < NSAMP;
    const int D = a.D, H = a.H;
    const long long B = a.B;
    const long long nst = (B + SUP - 1) / SUP;
    const DRs R0{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, 0x7fffffff, 0x00020000), (unsigned)lane * 16u};
    const float inv_fs = ACT == CNF_ACT_TANH_PRESCALED ? 1.f / kTanhPrescale : 1.f;   // the forward images of tanh nets carry the pre-scale
    const int ns = a.T.ns < NS ? a.T.ns : NS;
    const float dt = a.dt, tn = a.tn;
    const long long nsB = (long long)ns * B;
    const int ckzr = G.ckzr;
    const GUnits U = gunits<NSAMP>(A, G.b, wave);
    const int mtS0 = wave * A;
    const GOff<A> TZ = g_offsets<A>(R0, G.KPZ, mtS0, U);
    const GOff<A> TH = g_offsets<A>(R0, G.HTP, mtS0, U);
    unsigned vd[DT];
#pragma unroll
    for (int dm = 0; dm < DT; ++dm) { vd[dm] = R0.lane16 + (unsigned)(dm * G.HTP) * 1024u; asm volatile("" : "+v"(vd[dm])); }
    const unsigned F1Z = (unsigned)G.f1z * 4u, FH = (unsigned)G.fh * 4u, BN = (unsigned)G.bN * 4u, BH = (unsigned)G.bh * 4u, B1 = (unsigned)G.b1 * 4u;
    const unsigned ldx = (unsigned)H * 4u, ldy = (unsigned)a.ldy * 4u;
    __amdgpu_buffer_rsrc_t rx[2], ry[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        rx[l] = __builtin_amdgcn_make_buffer_rsrc(a.xh[l], 0, (int)((long long)ldx * 2 * nsB), 0x00020000);
        ry[l] = __builtin_amdgcn_make_buffer_rsrc(a.yh[l], 0, (int)((long long)ldy * 2 * nsB), 0x00020000);
    }

    auto cvec_units = [&](const float* __restrict__ vec, f32x4 (&vS)[A], f32x4 (&vR)[2]) {
#pragma unroll
        for (int m = 0; m < A; ++m) vS[m] = *reinterpret_cast<const f32x4*>(vec + ((mtS0 + m) * 4 + g) * 4);
        vR[0] = *reinterpret_cast<const f32x4*>(vec + (U.t0 * 4 + g) * 4);
        vR[1] = *reinterpret_cast<const f32x4*>(vec + (U.t1 * 4 + g) * 4);
    };
    auto publish = [&](f32x4* __restrict__ xb, const GAcc<A, NSAMP>& v) {
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) xb[((mtS0 + m) * NC + c) * 64 + lane] = v.S[m][c];
        if (U.v0) { xb[(U.t0 * NC + U.q) * 64 + lane] = v.R[0][0]; xb[(U.t0 * NC + NSAMP + U.q) * 64 + lane] = v.R[0][1]; }
        if (U.v1) { xb[(U.t1 * NC + U.q) * 64 + lane] = v.R[1][0]; xb[(U.t1 * NC + NSAMP + U.q) * 64 + lane] = v.R[1][1]; }
    };
    // chain 0 <- the C vector of this wave's units, chain 1 <- 0
    auto acc_init = [&](GAcc<A, NSAMP>& u, const f32x4 (&vS)[A], const f32x4 (&vR)[2]) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int q = 0; q < NSAMP; ++q) { u.S[m][q] = vS[m]; u.S[m][NSAMP + q] = z; }
        u.R[0][0] = vR[0]; u.R[0][1] = z; u.R[1][0] = vR[1]; u.R[1][1] = z;
    };
    auto acc_zero = [&](GAcc<A, NSAMP>& u) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < A; ++m)
#pragma unroll
            for (int c = 0; c < NC; ++c) u.S[m][c] = z;
        u.R[0][0] = z; u.R[0][1] = z; u.R[1][0] = z; u.R[1][1] = z;
    };
    // this wave's share of an exchange buffer -> rows of a column-major operand array, as full 128-byte lines: eight lanes cover 32
    // consecutive rows (two row tiles) of one sample (bank assignment: see gstore of cnf_coop_grad.hip).  Four columns: column
    // `wave` of every tile pair; two columns: column wave & 1 of every other tile pair
    const int mm = (lane >> 2) & 1, gg = lane & 3, s8 = (lane >> 3) ^ (4 * mm);
    const int gcol = NSAMP == 2 ? wave : (wave & 1), gpar = wave >> 1;
    auto gstore = [&](const f32x4* __restrict__ xb4, const __amdgpu_buffer_rsrc_t& rs, const unsigned (&vo)[2], unsigned so_c0, unsigned so_c1) {
        typedef float __attribute__((may_alias)) float_a;
        const float_a* xb = reinterpret_cast<const float_a*>(xb4);
        const unsigned so0 = gcol >= NSAMP ? so_c1 : so_c0;
        // all pairs an instance can have (HT <= 4 A + 3), requested in one batch; pairs beyond the real tiles read the last tile
        // and are dropped by the store's out-of-range offset
#pragma unroll
        for (int p = 0; p < 2 * A + 2; ++p) {
            if (NSAMP == 1 && (p & 1) != gpar) continue;
            const int t = 2 * p + mm < HTE ? 2 * p + mm : HTE - 1;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int base = ((t * NC + gcol) * 64 + s8 + 8 * hf) * 4 + gg;
                f32x4 v;
                v[0] = xb[base]; v[1] = xb[base + 64]; v[2] = xb[base + 128]; v[3] = xb[base + 192];
                const unsigned so = so0 + (unsigned)(32 * p) * 4u;
                const unsigned vof = (16 * (2 * p + mm) + 4 * gg < H) ? vo[hf] : 0xffffffffu;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)vof, (int)so, DG_NT_AUX);
                CNF_STORE_DATA_HAZARD(v);
            }
        }
    };
    auto publish_dense = [&](f32x4* img, int ct, const float (&v)[KZ]) {
#pragma unroll
        for (int kg = 0; kg < DT; ++kg) img[(kg * NC + ct) * 64 + lane] = f32x4{v[4 * kg], v[4 * kg + 1], v[4 * kg + 2], v[4 * kg + 3]};
    };
    auto dense_store = [&](float* arr, int ld, long long col, const float (&v)[KZ]) {
#pragma unroll
        for (int s = 0; s < KZ; ++s) { const int f = 4 * s + g; if (f < D) arr[col * (long long)ld + f] = v[s]; }
    };

    for (long long st = blockIdx.x; st < nst; st += gridDim.x) {
        const long long smp0 = st * SUP;
        const long long smp = smp0 + (owner ? wave : 0) * 16 + n;
        const bool valid = owner && smp < B;
        const long long sc = smp < B ? smp : B - 1;
        const long long tile = st * NSAMP + (owner ? wave : 0), ntp = a.ntiles_pad;
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
        // byte offsets of this lane's sample in the operand arrays, for the stores of column `wave` (sample tile wave & 1)
        unsigned vox[2], voy[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const long long sq = smp0 + (gcol % NSAMP) * 16 + s8 + 8 * hf;
            const unsigned ro = 16u * (unsigned)gg + 64u * (unsigned)mm;
            vox[hf] = sq < B ? (unsigned)sq * ldx + ro : 0xffffffffu;
            voy[hf] = sq < B ? (unsigned)sq * ldy + ro : 0xffffffffu;
        }
        __syncthreads();                 // the previous super-tile's readers of the LDS images are done
        float* zbt = a.zb + (tile * 64 + lane) * (long long)(NS * KZ);
        f32x4 aS[A], aR[2];
        // stage-derivative checkpoints of stage `is` (rows j < ns - 1 for the stage state, row `is` of zdot and of g = eps^T J): one
        // batch of 16-byte loads, requested a stage ahead of their use
        f32x4 kr[NS - 1][DT], ki[DT], gi[DT];
        auto load_rows = [&](int is) {
            const long long rowb = (long long)a.step * ns * ntp + tile, rstride = ntp * 64 * (long long)ckzr;
            const float* kbase = a.ckpt_k + (rowb * 64 + lane) * ckzr;
            const float* gbase = (a.lam2 != 0.f ? a.ckpt_g : a.ckpt_k) + (rowb * 64 + lane) * ckzr;
#pragma unroll
            for (int j = 0; j < NS - 1; ++j) {
                const int jj = j < ns ? j : ns - 1;
#pragma unroll
                for (int q = 0; q < DT; ++q) kr[j][q] = *reinterpret_cast<const f32x4*>(kbase + jj * rstride + 4 * q);
            }
#pragma unroll
            for (int q = 0; q < DT; ++q) {
                ki[q] = *reinterpret_cast<const f32x4*>(kbase + is * rstride + 4 * q);
                gi[q] = *reinterpret_cast<const f32x4*>(gbase + is * rstride + 4 * q);
            }
        };
        // (A = 3: the rows would stay live across the loop's back edge beside 24 more accumulator registers, and this compiler's
        // AGPR-copy rewrite pass crashes instead of spilling: those instances request them in the dense phase itself)
        constexpr bool PREF = A == 2 && NSAMP == 2;
        if (PREF && owner) load_rows(ns - 1);

#pragma clang loop unroll(disable)
        for (int i = ns - 1; i >= 0; --i) {
#ifdef DG_TRACE
            unsigned long long tr[18];
#endif
            DG_T(0);
            // The image's buffer resource is rebuilt in every stage from the kernel argument, made scalar BY HAND: the kernel's scalar
            // registers overflow (kernel arguments, four operand-array resources, tableau coefficients), and whenever the allocator
            // parked the image pointer in a vector register every fragment load sat in a waterfall loop (4 v_readfirstlane +
            // compare + branch per load, each load its own basic block: an H x H product took 15.5 k cycles for 10.9 k of MFMAs).
            const unsigned long long pimg = (unsigned long long)a.packed;
            const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pimg), phi = __builtin_amdgcn_readfirstlane((unsigned)(pimg >> 32));
            const DRs R{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)phi << 32) | plo), 0, 0x7fffffff, 0x00020000), R0.lane16};
            float zs[KZ], kbar[KZ], gbar[KZ];
            const float bi = a.T.b[i];
            const float cl = valid ? dt * bi : 0.f;      // cotangent of ldot (dL/d dlogp = +1 per column); zero for padding columns
            const float tt = tn + a.T.c[i] * dt;
            const long long c1 = (long long)i * B, c2 = nsB + (long long)i * B;
            const unsigned sx1 = (unsigned)c1 * ldx, sx2 = (unsigned)c2 * ldx, sy1 = (unsigned)c1 * ldy, sy2 = (unsigned)c2 * ldy;
            if (owner) {
                // The other waves wait at the barrier below while the owners are here, and with one workgroup per CU nothing hides a
                // chain of dependent global round trips: everything this phase reads is requested in ONE batch of 16-byte loads
                // (straight-line code; rows the stage does not use are loaded and dropped).  (First build: per-row loops of dword
                // loads run by all four waves - 576 load instructions per stage through one address unit, a fifth of the kernel.)
                // The checkpoint rows (HBM) of this stage were requested one stage ago, behind the last barrier but one (load_rows
                // below): only the Zbar rows of the running step - written by this wave, L2 hits - are read here.
                f32x4 zr[NS - 1][DT];
                if constexpr (!PREF) load_rows(i);
#pragma unroll
                for (int j = 0; j < NS - 1; ++j)
#pragma unroll
                    for (int q = 0; q < DT; ++q) zr[j][q] = *reinterpret_cast<const f32x4*>(zbt + (j + 1) * KZ + 4 * q);
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
                // gbar = cotangent of g = eps^T J: -c_l eps (+ c_n g / |g|);  kbar += c_E zdot / |zdot|  (src/core/icnf.jl:184-251: Edot =
                // |zdot|, ndot = |eps^T J|; zdot_i and g_i of the stage are the forward solve's checkpoints)
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
                publish_dense(zebuf, wave, zs); publish_dense(zebuf, NSAMP + wave, gbar);
                publish_dense(ekbuf, wave, eps); publish_dense(ekbuf, NSAMP + wave, kbar);
            }
            GAcc<A, NSAMP> acc;   // (the elementwise phases work in place: results replace the accumulators they come from)
            GHalf<A, NSAMP> d1P, db1P, h1P;      // act'_1, dbar_1 (and h_1 for tanh) of this wave's units, parked until the way down
            // ================= up 1: [a_1 | dbar_1] = W_1[:,0:D] [z | gbar] (+ bias and time column on the first chain) =================
            {
                f32x4 bS[A], bR[2], wS[A], wR[2];
                cvec_units(P + G.v_b1, bS, bR);
                cvec_units(P + G.v_w1t, wS, wR);
                if (!a.autonomous) {
#pragma unroll
                    for (int m = 0; m < A; ++m) bS[m] = tile_fma(wS[m], tt, bS[m]);
                    bR[0] = tile_fma(wR[0], tt, bR[0]);
                    bR[1] = tile_fma(wR[1], tt, bR[1]);
                }
                acc_init(acc, bS, bR);
            }
            g_load_a<A>(R, TZ, F1Z, 0, aS, aR);
            DG_T(1);
            DG_SYNC();                                                                     // B0
            DG_T(2);
            g_gemm<A, NSAMP>(R, TZ, F1Z, G.KGZ, G.remZ, U, zebuf, lane, aS, aR, acc);
            DG_T(3);
            g_load_a<A>(R, TH, FH, 0, aS, aR);
            auto up_unit = [&](const f32x4& a0, const f32x4& a1, f32x4& o0, f32x4& o1, f32x4& hk, f32x4& dk, f32x4& dbk) {
                act_pair<ACT>(a0, hk, dk);
                dbk = a1 * inv_fs;          // dbar = W vbar of the level below (dbar_1 = W_1[:,0:D] gbar)
                o0 = hk;
                o1 = dbk * dk;              // vbar (cbar at the top)
            };
#pragma unroll
            for (int m = 0; m < A; ++m)
#pragma unroll
                for (int q = 0; q < NSAMP; ++q) {
                    f32x4 hk, dk, dbk;
                    up_unit(acc.S[m][q], acc.S[m][NSAMP + q], acc.S[m][q], acc.S[m][NSAMP + q], hk, dk, dbk);
                    d1P.S[m][q] = park4(dk); db1P.S[m][q] = park4(dbk);
                    if constexpr (KEEP_H) h1P.S[m][q] = park4(hk);
                }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 hk, dk, dbk;
                up_unit(acc.R[s][0], acc.R[s][1], acc.R[s][0], acc.R[s][1], hk, dk, dbk);
                d1P.R[s] = park4(dk); db1P.R[s] = park4(dbk);
                if constexpr (KEEP_H) h1P.R[s] = park4(hk);
            }
            publish(X0, acc);
            {
                f32x4 bS[A], bR[2];
                cvec_units(P + G.v_bh, bS, bR);
                acc_init(acc, bS, bR);
            }
            DG_T(4);
            DG_SYNC();                                                                     // B1
            DG_T(5);
            {
                // The dense operands of Wbar_1 and Wbar_N ([gbar; 0 | z; t; 1] and [eps | kbar]: rows of 4-byte stores, an exec mask per
                // row) leave HERE, behind the second barrier of the stage, read back from the LDS images ([z | gbar] lives in the second
                // exchange buffer until this stage's second publish) by the waves that wait longest at the NEXT barrier - the ones
                // without left-over units: the owners with two sample tiles (s_memtime: they waited 4.2 k cycles at this barrier
                // for the two other waves to finish these stores), waves 1 and 2 with one (wave 3 carries the left-over tile).
                const bool st_y = NSAMP == 2 ? wave < 2 : wave == 1, st_x = NSAMP == 2 ? wave < 2 : wave == 2;
                const int q = NSAMP == 2 ? wave : 0;
                const long long sm2 = smp0 + q * 16 + n;
                if ((st_y || st_x) && sm2 < B) {
                    float v0[KZ], v1[KZ];
                    auto fetch = [&](const f32x4* img, int ct, float (&v)[KZ]) {
#pragma unroll
                        for (int kg = 0; kg < DT; ++kg) {
                            const f32x4 t4 = img[(kg * NC + ct) * 64 + lane];
                            v[4 * kg] = t4[0]; v[4 * kg + 1] = t4[1]; v[4 * kg + 2] = t4[2]; v[4 * kg + 3] = t4[3];
                        }
                    };
                    if (st_y) {
                        fetch(zebuf, NSAMP + q, v0); fetch(zebuf, q, v1);
                        dense_store(a.y1, a.ld_y1, c1 + sm2, v0);
                        dense_store(a.y1, a.ld_y1, c2 + sm2, v1);
                        if (g == 0) {
                            float* col = a.y1 + (c2 + sm2) * (long long)a.ld_y1;
                            if (!a.autonomous) col[D] = tt;
                            col[a.ld_y1 - 1] = 1.f;
                        }
                    }
                    if (st_x) {
                        fetch(ekbuf, q, v0); fetch(ekbuf, NSAMP + q, v1);
                        dense_store(a.xN, D, c1 + sm2, v0);
                        dense_store(a.xN, D, c2 + sm2, v1);
                    }
                }
            }
            // ================= up 2: [a_2 | dbar_2] = W_2 [h_1 | vbar_1] =================
            g_gemm<A, NSAMP>(R, TH, FH, G.KGH, G.remH, U, X0, lane, aS, aR, acc);
            DG_T(6);
            g_load_a<A>(R, TZ, BN, 0, aS, aR);
            // (the operand stores go BEHIND the product and the next product's first fragment requests: the memory counter
            // retires in order, and a fragment wait behind twelve streaming stores waits for their acknowledgements)
            gstore(X0, ry[0], voy, sy2, sy1);                                              // Y_1 = [vbar_1; 0 | h_1; 1]
            GHalf<A, NSAMP> d2, db2, h2;
#pragma unroll
            for (int m = 0; m < A; ++m)
#pragma unroll
                for (int q = 0; q < NSAMP; ++q) up_unit(acc.S[m][q], acc.S[m][NSAMP + q], acc.S[m][q], acc.S[m][NSAMP + q], h2.S[m][q], d2.S[m][q], db2.S[m][q]);
#pragma unroll
            for (int s = 0; s < 2; ++s) up_unit(acc.R[s][0], acc.R[s][1], acc.R[s][0], acc.R[s][1], h2.R[s], d2.R[s], db2.R[s]);
            publish(X1, acc);
            acc_zero(acc);
            DG_T(7);
            DG_SYNC();                                                                     // B2
            DG_T(8);
            // ================= the top: [c | hbar_2] = W_N^T [eps | kbar] =================
            g_gemm<A, NSAMP>(R, TZ, BN, G.KGZ, G.remZ, U, ekbuf, lane, aS, aR, acc);
            DG_T(9);
            g_load_a<A>(R, TH, BH, 0, aS, aR);
            gstore(X1, ry[1], voy, sy2, sy1);                                              // Y_2 = [cbar; 0 | h_2; 1]
            // delta = u .* act', a2 = dbar .* u, sbar = hbar .* act' + a2 .* act''
            auto down_unit = [&](const f32x4& u, const f32x4& hb, const f32x4& h, const f32x4& d, const f32x4& db, f32x4& o0, f32x4& o1) {
                const f32x4 sbar = hb * d + (db * u) * dg_act_dd<ACT>(h, d);   // (o0 / o1 may be u / hb themselves)
                o0 = u * d;
                o1 = sbar;
            };
#pragma unroll
            for (int m = 0; m < A; ++m)
#pragma unroll
                for (int q = 0; q < NSAMP; ++q) down_unit(acc.S[m][q], acc.S[m][NSAMP + q], h2.S[m][q], d2.S[m][q], db2.S[m][q], acc.S[m][q], acc.S[m][NSAMP + q]);
#pragma unroll
            for (int s = 0; s < 2; ++s) down_unit(acc.R[s][0], acc.R[s][1], h2.R[s], d2.R[s], db2.R[s], acc.R[s][0], acc.R[s][1]);
            publish(X0, acc);
            acc_zero(acc);
            DG_T(10);
            DG_SYNC();                                                                     // B3
            DG_T(11);
            // ================= down 2: [u_1 | hbar_1] = W_2^T [delta_2 | sbar_2] =================
            g_gemm<A, NSAMP>(R, TH, BH, G.KGH, G.remH, U, X0, lane, aS, aR, acc);
            DG_T(12);
            f32x4 fd[DT];
#pragma unroll
            for (int dm = 0; dm < DT; ++dm) fd[dm] = dloadv(R, vd[dm], B1 + (unsigned)mtS0 * 1024u);
            gstore(X0, rx[1], vox, sx1, sx2);                                              // X_2 = [delta_2 | sbar_2]
            GHalf<A, NSAMP> sb;
#pragma unroll
            for (int m = 0; m < A; ++m)
#pragma unroll
                for (int q = 0; q < NSAMP; ++q) {
                    const f32x4 hk = KEEP_H ? unpark4(h1P.S[m][q]) : f32x4{0.f, 0.f, 0.f, 0.f};
                    down_unit(acc.S[m][q], acc.S[m][NSAMP + q], hk, unpark4(d1P.S[m][q]), unpark4(db1P.S[m][q]), acc.S[m][q], acc.S[m][NSAMP + q]);
                    sb.S[m][q] = acc.S[m][NSAMP + q];
                }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 hk = KEEP_H ? unpark4(h1P.R[s]) : f32x4{0.f, 0.f, 0.f, 0.f};
                down_unit(acc.R[s][0], acc.R[s][1], hk, unpark4(d1P.R[s]), unpark4(db1P.R[s]), acc.R[s][0], acc.R[s][1]);
                sb.R[s] = acc.R[s][1];
            }
            publish(X1, acc);
            // ================= Zbar_i = W_1[:,0:D]^T sbar_1: partial tiles over this wave's own k-groups, from registers =================
            f32x4 part[DT][NSAMP], own[DT];
            g_drow<A, NSAMP, DT>(R, vd, B1, mtS0, G.KGH, G.remH, U, sb, fd, part, own);
            DG_T(13);
            DG_SYNC();                                                                     // B4 (every reader of X0 is through)
            DG_T(14);
            gstore(X1, rx[0], vox, sx1, sx2);                                              // X_1 = [delta_1 | sbar_1]
            if (PREF && owner && i > 0) load_rows(i - 1);   // the next stage's checkpoint rows: they arrive under the stores, the barrier and the reduction
#pragma unroll
            for (int dm = 0; dm < DT; ++dm) {
#pragma unroll
                for (int q = 0; q < NSAMP; ++q) pbuf[((q * 4 + wave) * DT + dm) * 64 + lane] = part[dm][q];
                if (U.v0 || U.v1) pbuf[((NSAMP * 4 + wave) * DT + dm) * 64 + lane] = own[dm];
            }
            DG_T(15);
            DG_SYNC();                                                                     // B5
            DG_T(16);
            if (owner) {
                // waves 3 - q and 1 - q hold the left-over units of sample tile q (units q, q + 4 and q + 2, q + 6); with one sample tile
                // the b left-over tiles sit on waves 3, 2, 1
                const bool lo0 = wave < 2 * G.b, lo2 = wave + 2 < 2 * G.b;
#pragma unroll
                for (int dm = 0; dm < DT; ++dm) {
                    f32x4 v = pbuf[((wave * 4 + 0) * DT + dm) * 64 + lane];
#pragma unroll
                    for (int w = 1; w < 4; ++w) v += pbuf[((wave * 4 + w) * DT + dm) * 64 + lane];
                    if constexpr (NSAMP == 2) {
                        if (lo0) v += pbuf[((2 * 4 + 3 - wave) * DT + dm) * 64 + lane];
                        if (lo2) v += pbuf[((2 * 4 + 1 - wave) * DT + dm) * 64 + lane];
                    } else {
#pragma unroll
                        for (int u = 0; u < 3; ++u)
                            if (u < G.b) v += pbuf[((1 * 4 + 3 - u) * DT + dm) * 64 + lane];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) zbt[i * KZ + 4 * dm + j] = v[j];
                }
            }
#ifdef DG_TRACE
            DG_T(17);
            if (blockIdx.x == 3 && st == 3 && a.step == 5 && i == 2 && lane == 0) {
#define DG_D(k) (int)(tr[k] - tr[k - 1])
                printf("w%d: dense %d B0 %d up1 %d ew1 %d B1 %d up2 %d ew2 %d B2 %d top %d ew3 %d B3 %d dn2 %d ew4 %d B4 %d gst %d B5 %d red %d\n", wave, DG_D(1), DG_D(2),
                       DG_D(3), DG_D(4), DG_D(5), DG_D(6), DG_D(7), DG_D(8), DG_D(9), DG_D(10), DG_D(11), DG_D(12), DG_D(13), DG_D(14), DG_D(15), DG_D(16), DG_D(17));
            }
#endif
            // (the next stage's first LDS writes - the [z | gbar] image in X1, [eps | kbar] - follow the stores of X1 this wave has
            // just issued and every wave's reads of the [eps | kbar] image (before B3); the partial tiles are read before B0)
        }
        if (owner) {
#pragma unroll
            for (int s = 0; s < KZ; ++s) {
                float acc = lam[s];
                for (int j = 0; j < ns; ++j) acc += zbt[j * KZ + s];
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
template <int A, int KZ, int ACT, int NS, int NSAMP>
static hipError_t launch_dgrad(const DGArgs& a, int lds, int nblocks, hipStream_t st) {
    auto kern = coopd_grad_step_kernel<A, KZ, ACT, NS, NSAMP>;
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

struct DGradInst {
    int A, KZ, ACT, NSAMP;
    hipError_t (*fn[2])(const DGArgs&, int, int, hipStream_t);   // [0] RK4 (4 stages), [1] Tsit5 (6 stages)
};
#define DG_INST(A, KZ, ACT) DGradInst { A, KZ, ACT, 2, { &launch_dgrad<A, KZ, ACT, 4, 2>, &launch_dgrad<A, KZ, ACT, 6, 2> } }
// 16 .. 24 hidden tiles: one sample tile per super-tile (two column tiles per product)
#define DG1_INST(A, KZ, ACT) DGradInst { A, KZ, ACT, 1, { &launch_dgrad<A, KZ, ACT, 4, 1>, &launch_dgrad<A, KZ, ACT, 6, 1> } }
// the (A, KZ) pairs of cnf_coop_d.hip's forward instances
static const DGradInst kDGrad[] = {
    DG_INST(1, 8, CNF_ACT_SOFTPLUS),   // 5 .. 7 hidden tiles (the auxiliary cooperative plans of the default architecture at nvariables = 12, 13)
    DG_INST(2, 8, CNF_ACT_SOFTPLUS), DG_INST(3, 8, CNF_ACT_SOFTPLUS),
    DG_INST(2, 12, CNF_ACT_SOFTPLUS), DG_INST(3, 12, CNF_ACT_SOFTPLUS), DG_INST(3, 16, CNF_ACT_SOFTPLUS),
    DG_INST(2, 8, CNF_ACT_TANH_PRESCALED), DG_INST(2, 12, CNF_ACT_TANH_PRESCALED),   // tanh keeps h_1 as well: 8 .. 11 hidden tiles
    // the (A, KZ) pairs of cnf_coop_d2.hip's forward instances (the default architecture at nvariables = 30 .. 47)
    DG1_INST(4, 16, CNF_ACT_SOFTPLUS), DG1_INST(4, 20, CNF_ACT_SOFTPLUS), DG1_INST(5, 20, CNF_ACT_SOFTPLUS), DG1_INST(5, 24, CNF_ACT_SOFTPLUS), DG1_INST(6, 24, CNF_ACT_SOFTPLUS),
};
static const DGradInst* dg_find(int HT_real, int KZ, int ACT) {
    const int A = HT_real / 4;
    const DGradInst* best = nullptr;
    for (const DGradInst& c : kDGrad) {
        const bool act_ok = c.ACT == ACT || (c.ACT == CNF_ACT_TANH_PRESCALED && ACT == CNF_ACT_TANH);
        if (c.A == A && c.KZ >= KZ && act_ok && (!best || c.KZ < best->KZ)) best = &c;
    }
    return best;
}

// the [z | gbar] image aliases the second exchange buffer; the partial tiles alias the first where they fit it (*palias), else
// they get their own region if the LDS has room
static bool dg_fits(int HT_real, int DT, int A, int cvn, int NSAMP, bool* palias) {
    const int HTE = (HT_real + 1) / 2 * 2, b = HT_real - 4 * A, NC = 2 * NSAMP;
    if (DT > HTE) return false;
    const bool al = (NSAMP + (b > 0 ? 1 : 0)) * 4 * DT <= HTE * NC;
    if (palias) *palias = al;
    return coopd_grad_lds_bytes(HT_real, DT, cvn, NSAMP, al) <= 160 * 1024;
}

// H hidden units, D state rows, L hidden layers; (HT, ZR, CR) = the plan's layout
bool coopd_grad_supported(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int CR_lay) {
    if (L != 2 || CR_lay != 0) return false;
    bool force = false;
    if (tuning().coopd_grad == 0) return false;
    force = tuning().coopd_grad == 2;
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    if (HT_real < 5 || HT_real > HT_lay || KZ > ZR_lay) return false;
    const DGradInst* c = dg_find(HT_real, KZ, ACT);
    if (!c || (c->KZ + 3) / 4 > (ZR_lay + 3) / 4) return false;
    // (where the plan's layout IS the configuration - hidden tiles a multiple of four, state k-steps as laid out - cnf_coop_grad.hip
    // multiplies no padding either; the dealt sweep is still 1 - 3 % faster there: nvariables = 15 40.0 -> 38.8 ms, 30 121.2 -> 117.1,
    // 47 277.1 -> 273.8 at B = 32 768)
    (void)force;
    const int cvn = (1 + L) * 16 * HT_lay + 16 * ((ZR_lay + 3) / 4);
    return dg_fits(HT_real, c->KZ / 4, c->A, cvn, c->NSAMP, nullptr);
}

hipError_t coopd_grad_step_launch(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, const CGArgs& a, int num_cus, hipStream_t st) {
    const int HT_real = (H + 15) / 16, KZ = (D + 3) / 4;
    const DGradInst* c = L == 2 ? dg_find(HT_real, KZ, ACT) : nullptr;
    if (!c) return hipErrorNotSupported;
    DGArgs da{};
    da.c = a;
    dimg_fill(da.g, H, D, L, HT_lay, ZR_lay, 0, c->A, 0);
    const int DT = c->KZ / 4;
    bool palias = true;
    if (!dg_fits(HT_real, DT, c->A, da.g.cvn, c->NSAMP, &palias)) return hipErrorNotSupported;
    da.g.xalias = palias ? 1 : 0;
    const int lds = coopd_grad_lds_bytes(HT_real, DT, da.g.cvn, c->NSAMP, palias);
    const long long nst = (a.B + 16 * c->NSAMP - 1) / (16 * c->NSAMP);
    const int nblocks = (int)(nst < num_cus ? nst : num_cus);
    return c->fn[a.T.ns <= 4 ? 0 : 1](da, lds, nblocks, st);
}

}  // namespace cnf
