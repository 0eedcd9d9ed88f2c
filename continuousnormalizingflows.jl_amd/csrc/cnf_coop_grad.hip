// cnf_coop_grad.hip — cooperative reverse sweep for wide hidden layers (gfx950): the parameter gradient of the FFJORD loss
// through the fixed-step solve (reference: Zygote through SciMLBase.solve with QuadratureAdjoint +
// ZygoteVJP, src/core/icnf.jl:90-99, driven by src/exts/mlj_ext/core_icnf.jl:42-51) for the shapes whose forward solve runs on
// the cooperative kernel of cnf_coop.hip (BASELINE cfg4: D = 32, 3 x 256).
//
// Same mathematics as cnf_grad.hip (discretise-then-optimise; per stage: recompute the chain, first-order pullback, its
// bottom-up reverse, the top-down pass, the RK adjoint recursion) organised like cnf_coop.hip: a 256-thread workgroup owns a
// 64-sample super-tile, wave w the output features [w H/4, (w+1) H/4) of every product for all four sample tiles, operands
// travel between the waves as B images in two LDS exchange buffers, weight fragments stream from the L2-resident packed image.
// One launch per RK step (all its stages, in reverse); nothing of a stage's activation set ever makes a round trip through
// HBM as a GEMM operand.  What does not fit the register file between its producer and its consumer (h_l, u_l, dbar_l .* u_l:
// 64 registers per wave each) waits in a per-workgroup scratch in tile-native layout (one coalesced 1 KB store / load per tile).
//
// WEIGHT COTANGENTS ARE DEFERRED: with 256 x 256 matrices the cotangent accumulators (2 x 256 KB per workgroup) fit neither
// registers nor LDS, and a product with the sample index on K wants all samples of a column chunk anyway.  The kernel writes
// the operands of  Wbar_{l+1} += delta_{l+1} vbar_l^T + sbar_{l+1} [h_l; 1]^T  (and of Wbar_1, Wbar_N) for every stage of the
// step into column-major arrays laid out so that ONE lg_wgrad launch per weight matrix per step (K = 2 x stages x B columns)
// accumulates both terms and the bias column (cnf_lgemm.hip; slabs per column chunk, summed in a fixed order at the end).
// A tile leaves for HBM through the exchange buffer it was published in: the lane that stores rows 4r .. 4r + 3 of a sample
// reads its four values with conflict-free ds_read_b32 and issues one 16-byte store.
//
// Scope of this first version: Hutchinson VJP, one probe, no conditions, tanh, no |zdot| / |eps^T J| regularisers (FFJORD;
// l3 |z_aug| is in), uniform steps, 2 or 3 hidden layers of one width: everything else stays on the layer-wise path.
#define CNF_NO_PK_ASM 1
#define CNF_NO_PHASE_FENCE 1
#include "cnf_coop_dev.h"
#include "cnf_coop_grad.h"

// The operand arrays (read next by the weight-cotangent kernels) and the per-workgroup scratch move 5 GB per launch; written and
// read with the non-temporal hint they stream past the L2 instead of evicting the 1.15 MB operand image every workgroup
// re-reads for every product (-DCG_TEMPORAL restores plain accesses for an A/B).
#ifndef CG_TEMPORAL
#define CG_NT_AUX 2                                        // raw buffer store: nt
#define CG_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#define CG_NT_LOAD(p) __builtin_nontemporal_load((p))
#else
#define CG_NT_AUX 0
#define CG_NT_STORE(v, p) (*(p) = (v))
#define CG_NT_LOAD(p) (*(p))
#endif

namespace cnf {

namespace {

template <int MTW, int NT>
struct TS {   // one wave's share of an [H x 64-sample] quantity: MTW x NT accumulator tiles
    f32x4 t[MTW][NT];
};

// d = act'(a) from h = tanh(a)  (act''(a) = -2 h d)
__device__ __forceinline__ f32x4 tanh_d(const f32x4& h) { return 1.f - h * h; }
// A product's k-loop in two parts, so that loads whose data is needed only AFTER the product (the scratch operands of the
// elementwise phase that follows) can be requested behind the product's LAST fragment requests: they then return under its
// final two k-groups instead of sitting in front of its fragment loads in the in-order vmcnt queue.
//   head: k-groups 0 .. KG-3; on return a0 / b0 hold the (requested) fragments of k-group KG-2, a1 / b1 those of KG-1
//   tail: the MFMAs of k-groups KG-2 and KG-1.        (KG even: 2 for the D-sized products, HT for the H x H ones)
template <int M, int NQ, int NT>
__device__ __forceinline__ void coop_gemm_head(const AImg& A, int mt0, int KG, const f32x4* __restrict__ bimg, int lane,
                                               f32x4 (&a0)[M], f32x4 (&a1)[M], f32x4 (&b0)[NQ], f32x4 (&b1)[NQ], f32x4 (&acc)[M][NQ]) {
    coop_load_b<NQ, NT>(bimg, 0, 0, lane, b0);
#pragma clang loop unroll(disable)
    for (int kg = 0; kg + 2 < KG; kg += 2) {
        coop_load_a<M>(A, mt0, KG, kg + 1, a1); coop_load_b<NQ, NT>(bimg, 0, kg + 1, lane, b1);
        coop_frag_mfma<M, NQ>(a0, b0, acc);
        coop_load_a<M>(A, mt0, KG, kg + 2, a0); coop_load_b<NQ, NT>(bimg, 0, kg + 2, lane, b0);
        coop_frag_mfma<M, NQ>(a1, b1, acc);
    }
    if (KG >= 2) { coop_load_a<M>(A, mt0, KG, KG - 1, a1); coop_load_b<NQ, NT>(bimg, 0, KG - 1, lane, b1); }
}
template <int M, int NQ>
__device__ __forceinline__ void coop_gemm_tail(int KG, const f32x4 (&a0)[M], const f32x4 (&a1)[M], const f32x4 (&b0)[NQ],
                                               const f32x4 (&b1)[NQ], f32x4 (&acc)[M][NQ]) {
    coop_frag_mfma<M, NQ>(a0, b0, acc);
    if (KG >= 2) coop_frag_mfma<M, NQ>(a1, b1, acc);      // (KG == 1: a D-sized product with D <= 16)
}
// 16-byte store to a 4-byte-aligned address (odd leading dimensions): one global_store_dwordx4, not four scattered dwords
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

}  // namespace

// NT: sample tiles per super-tile.  4: one workgroup per CU, one wave per SIMD with up to 512 registers (every tile set of a
// wave is 64 registers).  2: 32-sample super-tiles, two workgroups per CU, two waves per SIMD with 256 registers each (tile
// sets of 32): half the reuse of every weight fragment, but a second wave on the SIMD to run while the first waits for its
// scratch / operand traffic - which is where this kernel, unlike the forward solve, spends its stalls.
template <int HT, int L, int ZR, int ACT, int NS, int NT = 4>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT == 4 ? 1 : 2, NT == 4 ? 1 : 2)))
coop_grad_step_kernel(CGArgs a) {
    static_assert(ACT == CNF_ACT_TANH_PRESCALED, "tanh nets only (act' and act'' are rebuilt from h)");
    static_assert(L == 2 || L == 3, "two or three hidden layers");
    static_assert(NT == 4 || NT == 2, "64- or 32-sample super-tiles");
    constexpr int SUP = 16 * NT;
    constexpr MfmaLayout LAY(HT, L, ZR, 0, true);
    constexpr int MTW = HT / 4, DT = (ZR + 3) / 4, XB = HT * NT * 64, DB = DT * NT * 64;
    constexpr int IMG = MfmaLayout::imgA(HT, HT);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xbuf = reinterpret_cast<f32x4*>(smem);   // [2][HT][NT][64]: exchange buffers
    f32x4* zbuf = xbuf + 2 * XB;                     // [DT][NT][64]: stage state
    f32x4* ebuf = zbuf + DB;                         // eps
    f32x4* kbuf = ebuf + DB;                         // kbar
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mt0 = wave * MTW;
    const int D = a.D, H = a.H;
    const long long B = a.B;
    const long long nst = (B + SUP - 1) / SUP;
    const bool owner = wave < NT;                     // this wave integrates sample tile `wave` of the super-tile
    const float* P = a.packed;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
#define AIMG(X) AImg{rP, (unsigned)(X) * 4u, lane16, nullptr}
    const float inv_fs = 1.f / kTanhPrescale;        // the forward images carry the tanh pre-scale
    const int ns = a.T.ns < NS ? a.T.ns : NS;
    const float dt = a.dt, tn = a.tn;
    const long long nsB = (long long)ns * B;
    using T4 = f32x4[MTW][NT];

    // own-tile helpers -------------------------------------------------------------------------------------------------
    auto publish = [&](int buf, const T4& v) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) xbuf[buf * XB + ((mt0 + m) * NT + q) * 64 + lane] = v[m][q];
    };
    // this wave's tiles of exchange buffer `buf` -> rows [16 mt0, 16 (mt0 + MTW)) of a column-major operand array.  Lane
    // (r = lane >> 4, n) stores rows 16 mt + 4 r .. + 3 of sample n: four conflict-free ds_read_b32 and one 16-byte buffer store
    // (per-lane byte offset voff[q] precomputed per super-tile - 0xffffffff, i.e. out of range, for padding columns and rows -
    // and the (column block, row tile) offset wave-uniform: no address arithmetic per tile).
    auto gstore = [&](int buf, const __amdgpu_buffer_rsrc_t& rs, const unsigned (&voff)[NT], unsigned soff0, unsigned ldb) {
#ifdef CG_EXP_NO_GSTORE
        return;
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own ds_writes have landed; the reads below alias them
        // (float reads of memory written as f32x4: a may_alias type, and a compiler barrier behind the stores - type-based alias
        // analysis would otherwise let a later publish() into the same tiles move ahead of the reads)
        typedef float __attribute__((may_alias)) float_a;
        const float_a* xb = reinterpret_cast<const float_a*>(xbuf + buf * XB);
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int base = (((mt0 + m) * NT + q) * 64 + n) * 4 + g;   // + 64 g': lane group g' of the image
                f32x4 v;
                v[0] = xb[base]; v[1] = xb[base + 64]; v[2] = xb[base + 128]; v[3] = xb[base + 192];
                const unsigned so = soff0 + (unsigned)(16 * (mt0 + m)) * 4u;
                const unsigned vo = (16 * (mt0 + m) + 4 * g < H) ? voff[q] : 0xffffffffu;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)vo, (int)so, CG_NT_AUX);
            }
        asm volatile("" ::: "memory");
        (void)ldb;
    };
    float* scr = a.scratch + (long long)blockIdx.x * a.scratch_stride;
    auto sstore = [&](int slot, const T4& v) {
#ifdef CG_EXP_NO_SCRATCH
        return;
#endif
        f32x4* s4 = reinterpret_cast<f32x4*>(scr) + ((slot * 4 + wave) * MTW * NT) * 64 + lane;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) CG_NT_STORE(v[m][q], &s4[(m * NT + q) * 64]);
    };
    auto sload = [&](int slot, T4& v) {
#ifdef CG_EXP_NO_SCRATCH
        for (int m = 0; m < MTW; ++m) for (int q = 0; q < NT; ++q) v[m][q] = f32x4{0.5f, 0.25f, 0.125f, 0.0625f};
        return;
#endif
        const f32x4* s4 = reinterpret_cast<const f32x4*>(scr) + ((slot * 4 + wave) * MTW * NT) * 64 + lane;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) v[m][q] = CG_NT_LOAD(&s4[(m * NT + q) * 64]);
    };
    auto zero = [&](T4& v) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) v[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // A product in two halves: its first weight fragments are requested BEFORE the elementwise / store / barrier phase that
    // precedes it (an L2 round trip with nothing to hide behind otherwise), the k-loop runs after the barrier
    auto pre_a = [&](int img, int KG, f32x4 (&afr)[MTW]) { coop_load_a<MTW>(AIMG(img), mt0, KG, 0, afr); };
    auto run = [&](int img, int KG, const f32x4* bimg, f32x4 (&afr)[MTW], T4& acc) {
        coop_gemm<MTW, NT, NT>(AIMG(img), mt0, KG, bimg, 0, lane, afr, acc);
    };
    f32x4 fa1[MTW], fb0[NT], fb1[NT];     // the split k-loop's second fragment set
    auto head = [&](int img, int KG, const f32x4* bimg, f32x4 (&afr)[MTW], T4& acc) {
        coop_gemm_head<MTW, NT, NT>(AIMG(img), mt0, KG, bimg, lane, afr, fa1, fb0, fb1, acc);
    };
    auto tail = [&](int KG, f32x4 (&afr)[MTW], T4& acc) { coop_gemm_tail<MTW, NT>(KG, afr, fa1, fb0, fb1, acc); };
    // dense D-row registers of this wave's sample tile -> B image
    auto publish_dense = [&](f32x4* img, const float (&v)[ZR]) {
#pragma unroll
        for (int kg = 0; kg < DT; ++kg) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (4 * kg + j < ZR) ? v[(4 * kg + j) < ZR ? 4 * kg + j : 0] : 0.f;
            img[(kg * NT + wave) * 64 + lane] = o;
        }
    };
    // dense D-row registers -> rows [0, D) of column `col` of a column-major array
    auto dense_store = [&](float* arr, int ld, long long col, const float (&v)[ZR]) {
#pragma unroll
        for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) arr[col * (long long)ld + f] = v[s]; }
    };

    // scratch slots (tile-native): H_l (l = 1 .. L), U_l (l = 2 .. L-1), A2_l (l = 1 .. L-1), Q = W_1[:,0:D] eps, C = W_N^T eps
    constexpr int SLOT_H = 0, SLOT_U = L, SLOT_A2 = SLOT_U + (L - 2), SLOT_Q = SLOT_A2 + (L - 1), SLOT_C = SLOT_Q + 1;
    static_assert(SLOT_C + 1 == 3 * L - 1, "coop_grad_scratch_slots");
    // operand arrays as buffer resources (byte sizes: 2 ns B columns)
    const unsigned ldx = (unsigned)H * 4u, ldy = (unsigned)(H + 1) * 4u;
    __amdgpu_buffer_rsrc_t rx[L], ry[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        rx[l] = __builtin_amdgcn_make_buffer_rsrc(a.xh[l], 0, (int)((long long)ldx * 2 * nsB), 0x00020000);
        ry[l] = __builtin_amdgcn_make_buffer_rsrc(a.yh[l], 0, (int)((long long)ldy * 2 * nsB), 0x00020000);
    }

    for (long long st = blockIdx.x; st < nst; st += gridDim.x) {
        const long long smp0 = st * SUP;
        const long long smp = smp0 + (owner ? wave : 0) * 16 + n;    // the owner waves' own sample tile
        const bool valid = owner && smp < B;
        const long long sc = smp < B ? smp : B - 1;
        const long long tile = st * NT + (owner ? wave : 0), ntp = a.ntiles_pad;
        float eps[ZR], zn[ZR], lam[ZR];
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            eps[s] = f < D ? a.eps[sc * D + f] : 0.f;
            zn[s] = a.ckpt[(((long long)a.step * ntp + tile) * 64 + lane) * ZR + s];
        }
        if (a.step == a.nsteps - 1) {
            // lambda_N = dL/dz_N = z_N (+ l3 z_aug / |z_aug|, src/core/base_icnf.jl:106-122); zero for padding columns
#pragma unroll
            for (int s = 0; s < ZR; ++s) lam[s] = valid ? a.ckpt[(((long long)a.nsteps * ntp + tile) * 64 + lane) * ZR + s] : 0.f;
            if (a.lam3 != 0.f) {
                float sa = 0.f;
#pragma unroll
                for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) sa = fmaf(lam[s], lam[s], sa); }
                sa = group_sum(sa);
                const float inv = sa > 0.f ? a.lam3 * rsqrtf(sa) : 0.f;
#pragma unroll
                for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) lam[s] = fmaf(inv, lam[s], lam[s]); }
            }
        } else {
#pragma unroll
            for (int s = 0; s < ZR; ++s) lam[s] = a.lam[(tile * 64 + lane) * ZR + s];
        }
        // per-lane byte offsets of this lane's columns (sample q * 16 + n of the super-tile, rows 4 g ..) in the X / Y arrays;
        // validity of those columns as a weight
        unsigned vox[NT], voy[NT];
        float vq[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const long long sq = smp0 + q * 16 + n;
            vq[q] = sq < B ? 1.f : 0.f;
            vox[q] = sq < B ? (unsigned)sq * ldx + 16u * (unsigned)g : 0xffffffffu;
            voy[q] = sq < B ? (unsigned)sq * ldy + 16u * (unsigned)g : 0xffffffffu;
        }
        __syncthreads();                 // the previous super-tile's readers of the LDS images are done
        if (owner) publish_dense(ebuf, eps);
        __syncthreads();
        {   // solve-invariant products of the super-tile: q = W_1[:,0:D] eps (un-scaled), c = W_N^T eps
            f32x4 afr[MTW];
            T4 t;
            zero(t);
            pre_a(LAY.f1z, DT, afr);
            run(LAY.f1z, DT, ebuf, afr, t);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q) t[m][q] *= inv_fs;
            sstore(SLOT_Q, t);
            zero(t);
            pre_a(LAY.bN, DT, afr);
            run(LAY.bN, DT, ebuf, afr, t);
            sstore(SLOT_C, t);
        }
        float* zbt = a.zb + (tile * 64 + lane) * (long long)(NS * ZR);   // Zbar_j of this step's stages, this lane

#pragma clang loop unroll(disable)
        for (int i = ns - 1; i >= 0; --i) {
            // ---- this wave's sample tile: stage state, cotangent of the stage derivative ----
            float zs[ZR], kbar[ZR];
            const float bi = a.T.b[i];
            {
                float acc[ZR], kb[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) { acc[s] = 0.f; kb[s] = bi * lam[s]; }
                for (int j = 0; j < i; ++j) {
                    const float aij = a.T.a[i][j];
                    const float* kj = a.ckpt_k + ((((long long)a.step * ns + j) * ntp + tile) * 64 + lane) * ZR;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) acc[s] = fmaf(aij, kj[s], acc[s]);
                }
                for (int j = i + 1; j < ns; ++j) {
                    const float aji = a.T.a[j][i];
#pragma unroll
                    for (int s = 0; s < ZR; ++s) kb[s] = fmaf(aji, zbt[j * ZR + s], kb[s]);
                }
#pragma unroll
                for (int s = 0; s < ZR; ++s) { zs[s] = fmaf(dt, acc[s], zn[s]); kbar[s] = valid ? dt * kb[s] : 0.f; }
            }
            const float cl = dt * bi;                                 // cotangent of ldot (dL/d dlogp = +1 per column)
            const float tt = tn + a.T.c[i] * dt;
            const long long c1 = (long long)i * B, c2 = nsB + (long long)i * B;   // first / second half of the operand arrays
            const unsigned sx1 = (unsigned)c1 * ldx, sx2 = (unsigned)c2 * ldx, sy1 = (unsigned)c1 * ldy, sy2 = (unsigned)c2 * ldy;
            if (owner) { publish_dense(zbuf, zs); publish_dense(kbuf, kbar); }
            if (valid) {
                // Wbar_1 operands: [gbar; 0; 0] with gbar = -c_l eps, and [z; t; 1];  Wbar_N operands: eps, kbar
                float gb[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) gb[s] = -cl * eps[s];
                dense_store(a.y1, a.ld_y1, c1 + smp, gb);
                dense_store(a.y1, a.ld_y1, c2 + smp, zs);
                if (g == 0) {
                    float* col = a.y1 + (c2 + smp) * (long long)a.ld_y1;
                    if (!a.autonomous) col[D] = tt;
                    col[a.ld_y1 - 1] = 1.f;
                }
                dense_store(a.xN, D, c1 + smp, eps);
                dense_store(a.xN, D, c2 + smp, kbar);
            }
            f32x4 afr[MTW];
            T4 acc;
            // ================= (1) recompute the chain (forward images carry the pre-scale) =================
            {
                f32x4 bias[MTW], wt[MTW];
                pre_a(LAY.f1z, DT, afr);
                gload_cvec<MTW>(P + LAY.v_b1, mt0, g, bias);
                gload_cvec<MTW>(P + LAY.v_w1t, mt0, g, wt);
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const f32x4 b0 = a.autonomous ? bias[m] : tile_fma(wt[m], tt, bias[m]);
#pragma unroll
                    for (int q = 0; q < NT; ++q) acc[m][q] = b0;
                }
                __syncthreads();
                run(LAY.f1z, DT, zbuf, afr, acc);
            }
            int cur = 0;
            T4 cvL;
#pragma unroll
            for (int l = 0; l < L; ++l) {      // layer l + 1
                T4 h;
                if (l + 1 < L) pre_a(LAY.fh + l * IMG, HT, afr);
                else pre_a(LAY.bh + (L - 2) * IMG, HT, afr);                 // first pullback product
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q) { f32x4 dd; act_tile<ACT>(acc[m][q], h[m][q], dd); }
                publish(cur, h);
                sstore(SLOT_H + l, h);
                gstore(cur, ry[l], voy, sy2, ldy);                            // [h_{l+1}; 1] half of Y_{l+1}
                if (l + 1 < L) {
                    f32x4 bnx[MTW];
                    gload_cvec<MTW>(P + LAY.v_bh + l * MfmaLayout::vecC(HT), mt0, g, bnx);
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) acc[m][q] = bnx[m];
                    __syncthreads();
                    if (l + 2 == L) {   // the last forward product: c (for delta_L) is requested under its final k-groups
                        head(LAY.fh + l * IMG, HT, xbuf + cur * XB, afr, acc);
                        sload(SLOT_C, cvL);
                        tail(HT, afr, acc);
                    } else {
                        run(LAY.fh + l * IMG, HT, xbuf + cur * XB, afr, acc);
                    }
                    cur ^= 1;
                } else {
                    // ===== (2) pullback starts here: delta_L = c .* act'_L while h_L is in registers =====
                    T4 dl;
                    T4& cv = cvL;
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) dl[m][q] = cv[m][q] * tanh_d(h[m][q]);
                    // delta_L goes into the buffer the layer-L product has just read: slower waves may still be inside that
                    // product (reading every wave's tiles), so this publish needs its own barrier in front
                    __syncthreads();
                    cur ^= 1;
                    publish(cur, dl);
                    gstore(cur, rx[L - 1], vox, sx1, ldx);                    // delta_L half of X_L
                    __syncthreads();
                }
            }
#pragma unroll
            for (int l = L - 1; l >= 1; --l) {   // u_l = W_{l+1}^T delta_{l+1}; delta_l = u_l .* act'_l
                T4 u, hl, qv;
                zero(u);
                head(LAY.bh + (l - 1) * IMG, HT, xbuf + cur * XB, afr, u);
                sload(SLOT_H + l - 1, hl);
                if (l == 1) sload(SLOT_Q, qv);
                tail(HT, afr, u);
                if (l > 1) pre_a(LAY.bh + (l - 2) * IMG, HT, afr);
                else pre_a(LAY.fh + 0 * IMG, HT, afr);                       // first bottom-up product
                cur ^= 1;
                if (l > 1) {
                    T4 dl;
                    sstore(SLOT_U + l - 2, u);
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) dl[m][q] = u[m][q] * tanh_d(hl[m][q]);
                    publish(cur, dl);
                    gstore(cur, rx[l - 1], vox, sx1, ldx);                    // delta_l half of X_l
                    __syncthreads();
                } else {
                    // delta_1, and at once the bottom of the reverse pass: dbar_1 = W_1[:,0:D] gbar = -c_l q,
                    // a2_1 = dbar_1 .* u_1, vbar_1 = dbar_1 .* act'_1
                    T4 t;
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) t[m][q] = u[m][q] * tanh_d(hl[m][q]);
                    publish(cur, t);
                    gstore(cur, rx[0], vox, sx1, ldx);                        // delta_1 half of X_1
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) {
                            const f32x4 d1 = qv[m][q] * (-cl * vq[q]);
                            t[m][q] = d1 * u[m][q];
                            qv[m][q] = d1 * tanh_d(hl[m][q]);
                        }
                    sstore(SLOT_A2 + 0, t);
                    publish(cur, qv);     // own tiles of delta_1 have been stored by this wave: overwrite them with vbar_1
                    gstore(cur, ry[0], voy, sy1, ldy);                        // [vbar_1; 0] half of Y_1
                    __syncthreads();
                }
            }
            // ================= (3) bottom-up: dbar_{l+1} = W_{l+1} vbar_l, vbar_l = dbar_l .* act'_l =================
            T4 a2L;    // a2_L = dbar_L .* c
#pragma unroll
            for (int l = 1; l < L; ++l) {
                T4 db, hl, x2;
                zero(db);
                head(LAY.fh + (l - 1) * IMG, HT, xbuf + cur * XB, afr, db);
                sload(SLOT_H + l, hl);                                        // h_{l+1}
                if (l + 1 < L) sload(SLOT_U + l - 1, x2);                     // u_{l+1}
                else sload(SLOT_C, x2);                                       // u_L = c
                tail(HT, afr, db);
                if (l + 1 < L) pre_a(LAY.fh + l * IMG, HT, afr);
                else pre_a(LAY.bN, DT, afr);                                  // hbar_L = W_N^T kbar
                cur ^= 1;
                if (l + 1 < L) {
                    T4 vb;
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) {
                            const f32x4 dd = db[m][q] * inv_fs;
                            x2[m][q] = dd * x2[m][q];
                            vb[m][q] = dd * tanh_d(hl[m][q]);
                        }
                    sstore(SLOT_A2 + l, x2);
                    publish(cur, vb);
                    gstore(cur, ry[l], voy, sy1, ldy);                        // [vbar_{l+1}; 0] half of Y_{l+1}
                    __syncthreads();
                } else {
                    T4 cb;   // cbar = dbar_L .* act'_L: Wbar_N += eps cbar^T
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) {
                            const f32x4 dd = db[m][q] * inv_fs;
                            a2L[m][q] = dd * x2[m][q];
                            cb[m][q] = dd * tanh_d(hl[m][q]);
                        }
                    publish(cur, cb);
                    gstore(cur, ry[L - 1], voy, sy1, ldy);                    // [cbar; 0] half of Y_L
                    __syncthreads();   // every wave has left the last bottom-up product: its operand buffer is free
                }
            }
            // ================= (4) top-down: sbar_l = hbar_l .* act'_l + a2_l .* act''_l, hbar_{l-1} = W_l^T sbar_l =================
            {
                T4 hb, hl, sb;
                zero(hb);
                head(LAY.bN, DT, kbuf, afr, hb);
                sload(SLOT_H + L - 1, hl);
                tail(DT, afr, hb);
                pre_a(LAY.bh + (L - 2) * IMG, HT, afr);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const f32x4 d = tanh_d(hl[m][q]);
                        sb[m][q] = hb[m][q] * d + a2L[m][q] * (hl[m][q] * d * -2.f);
                    }
                cur ^= 1;
                publish(cur, sb);
                gstore(cur, rx[L - 1], vox, sx2, ldx);                        // sbar_L half of X_L
                __syncthreads();
            }
            f32x4 afd[DT];
#pragma unroll
            for (int l = L - 1; l >= 1; --l) {   // hbar_l = W_{l+1}^T sbar_{l+1}
                T4 hb, hl, a2, sb;
                zero(hb);
                head(LAY.bh + (l - 1) * IMG, HT, xbuf + cur * XB, afr, hb);
                sload(SLOT_H + l - 1, hl);
                sload(SLOT_A2 + l - 1, a2);
                tail(HT, afr, hb);
                if (l > 1) pre_a(LAY.bh + (l - 2) * IMG, HT, afr);
                else if (owner) coop_load_a<DT>(AIMG(LAY.b1), 0, HT, 0, afd);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const f32x4 d = tanh_d(hl[m][q]);
                        sb[m][q] = hb[m][q] * d + a2[m][q] * (hl[m][q] * d * -2.f);
                    }
                cur ^= 1;
                publish(cur, sb);
                gstore(cur, rx[l - 1], vox, sx2, ldx);                        // sbar_l half of X_l
                __syncthreads();
            }
            // Zbar_i = W_1[:,0:D]^T sbar_1 for this wave's own sample tile
            if (owner) {
                f32x4 zacc[DT][1];
#pragma unroll
                for (int m = 0; m < DT; ++m) zacc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                coop_gemm<DT, 1, NT>(AIMG(LAY.b1), 0, HT, xbuf + cur * XB, wave, lane, afd, zacc);
#pragma unroll
                for (int s = 0; s < ZR; ++s) zbt[i * ZR + s] = zacc[s >> 2][0][s & 3];
            }
            __syncthreads();   // the next stage republishes zbuf / kbuf and reuses the exchange buffers
        }
        // lambda_n = lambda_{n+1} + sum_i Zbar_i
        if (owner) {
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                float acc = lam[s];
                for (int j = 0; j < ns; ++j) acc += zbt[j * ZR + s];
                lam[s] = acc;
                a.lam[(tile * 64 + lane) * ZR + s] = acc;
            }
        }
        if (a.step == 0 && a.grad_x && valid) {   // costate at t0 = dL/dz_0; its first nvars rows are dL/dx
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                const int f = 4 * s + g;
                if (f < a.nvars) a.grad_x[smp * a.nvars + f] = lam[s];
            }
        }
    }
#undef AIMG
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
template <int HT, int L, int ZR, int ACT, int NS, int NT>
static hipError_t launch_grad_step(const CGArgs& a, int num_cus, hipStream_t st) {
    constexpr int DT = (ZR + 3) / 4;
    constexpr int lds = (2 * HT * NT * 64 + 3 * DT * NT * 64) * 16;
    static_assert(lds * (NT == 4 ? 1 : 2) <= 160 * 1024, "exchange buffers exceed LDS");
    const long long nst = (a.B + 16 * NT - 1) / (16 * NT);
    const long long cap = (long long)num_cus * (NT == 4 ? 1 : 2);
    const int nblocks = (int)(nst < cap ? nst : cap);
    auto kern = coop_grad_step_kernel<HT, L, ZR, ACT, NS, NT>;
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

int coop_grad_nt();
struct CoopGradInst {
    int HT, L, ZR, ACT;
    hipError_t (*fn[4])(const CGArgs&, int, hipStream_t);   // [0] RK4 (4 stages), [1] Tsit5 (6 stages); [2], [3]: the same with NT = 2
};
#define CG_INST(HT, L, ZR) \
    CoopGradInst { HT, L, ZR, CNF_ACT_TANH_PRESCALED, { &launch_grad_step<HT, L, ZR, CNF_ACT_TANH_PRESCALED, 4, 4>, \
                                                        &launch_grad_step<HT, L, ZR, CNF_ACT_TANH_PRESCALED, 6, 4>, \
                                                        &launch_grad_step<HT, L, ZR, CNF_ACT_TANH_PRESCALED, 4, 2>, \
                                                        &launch_grad_step<HT, L, ZR, CNF_ACT_TANH_PRESCALED, 6, 2> } }
// the same (HT, L, ZR) as the forward instances of cnf_coop.hip they pair with (the plan's packed image is shared)
static const CoopGradInst kCoopGrad[] = {
    CG_INST(16, 3, 8),   // cfg4: D = 32, 3 x 256
    CG_INST(8, 3, 2),    // D <= 8, 3 x 128
    CG_INST(4, 3, 2),    // D <= 8, 3 x 64: cross-check of the register-accumulator kernel (cnf_grad.hip)
};

static const CoopGradInst* cg_find(int HT, int L, int ZR, int ACT) {
    for (const CoopGradInst& c : kCoopGrad)
        if (c.HT == HT && c.L == L && c.ZR == ZR && (ACT == CNF_ACT_TANH || ACT == CNF_ACT_TANH_PRESCALED)) return &c;
    return nullptr;
}

bool coop_grad_supported(int HT, int L, int ZR, int ACT) { return cg_find(HT, L, ZR, ACT) != nullptr; }
// sample tiles per super-tile of the reverse-sweep kernel: 2 (two workgroups per CU, two waves per SIMD: cfg4 loss + gradient
// 143 ms) unless CNF_CG_NT=4 asks for the one-wave-per-SIMD form (153 ms)
int coop_grad_nt() {
    static const int nt = [] { const char* e = getenv("CNF_CG_NT"); const int v = (e && *e) ? atoi(e) : 2; return v == 4 ? 4 : 2; }();
    return nt;
}
int coop_grad_scratch_slots(int L) { return 3 * L - 1; }   // H_1..H_L, U_2..U_{L-1}, A2_1..A2_{L-1}, Q, C

hipError_t coop_grad_step_launch(int HT, int L, int ZR, int ACT, const CGArgs& a, int num_cus, hipStream_t st) {
    const CoopGradInst* c = cg_find(HT, L, ZR, ACT);
    if (!c) return hipErrorNotSupported;
    return c->fn[(a.T.ns <= 4 ? 0 : 1) + (coop_grad_nt() == 2 ? 2 : 0)](a, num_cus, st);
}

}  // namespace cnf
