// cnf_coop_grad.hip — cooperative reverse sweep for wide hidden layers (gfx950): the parameter gradient of the FFJORD loss
// through the fixed-step solve (reference: Zygote through SciMLBase.solve with QuadratureAdjoint +
// ZygoteVJP, src/core/icnf.jl:90-99, driven by src/exts/mlj_ext/core_icnf.jl:42-51) for the shapes whose forward solve runs on
// the cooperative kernel of cnf_coop.hip (BASELINE cfg4: D = 32, 3 x 256).
//
// Same mathematics as cnf_grad.hip (discretise-then-optimise; per stage: recompute the chain, first-order pullback, its
// bottom-up reverse, the top-down pass, the RK adjoint recursion) organised like cnf_coop.hip: a 256-thread workgroup owns a
// 16-sample super-tile, wave w the output features [w H/4, (w+1) H/4) of every product, operands travel between the waves as
// B images in two LDS exchange buffers, weight fragments stream from the L2-resident packed image.  One launch per RK step (all
// its stages, in reverse); nothing of a stage's activation set ever makes a round trip through HBM as a GEMM operand.
//
// TWO CHAINS PER PRODUCT.  The stage's four passes pair up: the bottom-up tangent pass (dbar_{l+1} = W_{l+1} vbar_l) needs only
// act'_l of the forward recompute and multiplies with the SAME forward images; the top-down pass (hbar_l = W_{l+1}^T sbar_{l+1})
// multiplies with the same transposed images as the pullback (u_l = W_{l+1}^T delta_{l+1}) and needs, at its level, exactly what
// the pullback has just produced there (a2_l = dbar_l .* u_l).  So every product carries two chains side by side as two column
// tiles - [h_l | vbar_l] on the way up, [delta_{l+1} | sbar_{l+1}] on the way down - and a stage is 2 L products instead of
// 4 L - 2, with half the barriers.  What waits in the per-workgroup scratch (tile-native layout, one coalesced 1 KB store /
// load per tile) between its producer and its consumer is h_l (l < L) and dbar_l (1 < l < L): 3 stores and 3 loads of a tile
// set per stage at L = 3 (the four-pass organisation of this kernel's first version: 6 and 13); c = W_N^T eps and q = W_1 eps
// ride along in the D-sized products ([eps | kbar] and [z | eps] as two-chain operands) instead of being kept.  The kernel is
// bound by its HBM traffic (first version: 5.6 GB per launch at cfg4 against 0.98 ms of MFMA work, 1.75 ms per launch), and
// this form moves ~40 % less: cfg4 loss + gradient 127 -> 117 ms.
//
// WEIGHT COTANGENTS ARE DEFERRED: with 256 x 256 matrices the cotangent accumulators (2 x 256 KB per workgroup) fit neither
// registers nor LDS, and a product with the sample index on K wants all samples of a column chunk anyway.  The kernel writes
// the operands of  Wbar_{l+1} += delta_{l+1} vbar_l^T + sbar_{l+1} [h_l; 1]^T  (and of Wbar_1, Wbar_N) for every stage of the
// step into column-major arrays laid out so that ONE lg_wgrad launch per weight matrix per step (K = 2 x stages x B columns)
// accumulates both terms and the bias column (cnf_lgemm.hip; slabs per column chunk, summed in a fixed order at the end).
// A tile leaves for HBM through the exchange buffer it was published in, as FULL 128-BYTE LINES: eight lanes cover 32
// consecutive rows (two row tiles) of one sample - four conflict-free ds_read_b32 and one 16-byte store per lane (with 64-byte
// runs, one row tile per sample, the same bytes cost 4 % more of the whole gradient), and the Y arrays' leading dimension is
// padded to a multiple of 16 floats so that no store straddles a 64-byte block.
//
// REGULARISERS (the reference's default lambdas are non-zero, src/core/icnf.jl:73-75): the cotangents of Edot = |zdot| and
// ndot = |eps^T J| need zdot_i and g_i = eps^T J of the stage - both are checkpoints of the forward solve (ckpt_k, ckpt_g) -
// and enter as dense D-row operations of the owner wave: kbar += c_E zdot / |zdot|, gbar = -c_l eps + c_n g / |g|; the tangent
// chain then starts from dbar_1 = W_1[:,0:D] gbar, the D-sized product the two-chain form runs anyway.
//
// Scope: Hutchinson VJP, one probe, tanh or softplus, uniform steps, 2 or 3 hidden layers of one width, up to 16 conditions
// (the condition rows of layer 1 are one more small product per stage; their cotangent rides in the W_1 operand array):
// everything else stays on the layer-wise path.
#define CNF_NO_PK_ASM 1
#define CNF_NO_PHASE_FENCE 1
#include "cnf_coop_dev.h"
#include "cnf_coop_grad.h"

// One translation unit per activation (compile time): this file builds the tanh instances and the shared host side;
// cnf_coop_grad_softplus.hip includes it with CG_ACT_SOFTPLUS defined and contributes only its instance table.
#ifdef CG_ACT_SOFTPLUS
#define CG_ACT CNF_ACT_SOFTPLUS
#define CG_TABLE_FN coop_grad_table_softplus
#else
#define CG_ACT CNF_ACT_TANH_PRESCALED
#define CG_TABLE_FN coop_grad_table_tanh
#endif

// The operand arrays (read next by the weight-cotangent kernels) and the per-workgroup scratch are streams; written and
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

// The kernel's barriers order LDS traffic only (the exchange images); no wave reads global memory another wave of the launch
// wrote.  __syncthreads() also waits for every outstanding global store (vmcnt(0): the operand and scratch streams) at each of
// the ~7 barriers of a stage; a barrier that waits for the LDS queue alone was measured at cfg4: no difference (112.7 vs 112.4 ms;
// vmcnt retires in order, so the next product's first fragment wait drains the stores anyway) - the plain form stays.
#define CG_SYNC() __syncthreads()

namespace cnf {

namespace {

// d = act'(a) from h = act(a):  tanh: 1 - h^2 (act'' = -2 h d);  softplus: h = log(1 + e^a), so e^-h = 1 / (1 + e^a) and
// act' = sigmoid(a) = 1 - e^-h (act'' = d (1 - d))
template <int ACT>
__device__ __forceinline__ f32x4 act_d(const f32x4& h) {
    if constexpr (ACT == CNF_ACT_SOFTPLUS) {
        f32x4 d;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = 1.f - __expf(-h[j]);
        return d;
    } else {
        return 1.f - h * h;
    }
}
// act''(a) from h and d = act'(a)
template <int ACT>
__device__ __forceinline__ f32x4 act_dd(const f32x4& h, const f32x4& d) {
    if constexpr (ACT == CNF_ACT_SOFTPLUS) return d * (1.f - d);
    else return h * d * -2.f;
}
// A product's k-loop in two parts, so that loads whose data is needed only AFTER the product (the scratch operands of the
// elementwise phase that follows) can be requested behind the product's LAST fragment requests: they then return under its
// final two k-groups instead of sitting in front of its fragment loads in the in-order vmcnt queue.
//   head: k-groups 0 .. KG-3; on return a0 / b0 hold the (requested) fragments of k-group KG-2, a1 / b1 those of KG-1
//   tail: the MFMAs of k-groups KG-2 and KG-1.        (KG even: 2 for the D-sized products, HT for the H x H ones)
//   KP: the image's row pitch in k-groups (the instance's HT for the H x H images), KG <= KP the (even) number multiplied - a
//   hidden width that fills fewer tiles than the instance has skips the zero ones
template <int M, int NQ, int NT>
__device__ __forceinline__ void coop_gemm_head(const AImg& A, int mt0, int KP, int KG, const f32x4* __restrict__ bimg, int lane,
                                               f32x4 (&a0)[M], f32x4 (&a1)[M], f32x4 (&b0)[NQ], f32x4 (&b1)[NQ], f32x4 (&acc)[M][NQ]) {
    coop_load_b<NQ, NT>(bimg, 0, 0, lane, b0);
#pragma clang loop unroll(disable)
    for (int kg = 0; kg + 2 < KG; kg += 2) {
        coop_load_a<M>(A, mt0, KP, kg + 1, a1); coop_load_b<NQ, NT>(bimg, 0, kg + 1, lane, b1);
        coop_frag_mfma<M, NQ>(a0, b0, acc);
        coop_load_a<M>(A, mt0, KP, kg + 2, a0); coop_load_b<NQ, NT>(bimg, 0, kg + 2, lane, b0);
        coop_frag_mfma<M, NQ>(a1, b1, acc);
    }
    if (KG >= 2) { coop_load_a<M>(A, mt0, KP, KG - 1, a1); coop_load_b<NQ, NT>(bimg, 0, KG - 1, lane, b1); }
}
template <int M, int NQ>
__device__ __forceinline__ void coop_gemm_tail(int KG, const f32x4 (&a0)[M], const f32x4 (&a1)[M], const f32x4 (&b0)[NQ],
                                               const f32x4 (&b1)[NQ], f32x4 (&acc)[M][NQ]) {
    coop_frag_mfma<M, NQ>(a0, b0, acc);
    if (KG >= 2) coop_frag_mfma<M, NQ>(a1, b1, acc);      // (KG == 1: a D-sized product with D <= 16)
}

}  // namespace

// NT: sample tiles per super-tile (per chain); a product has CT = 2 NT column tiles
template <int HT, int L, int ZR, int CR, int ACT, int NS, int NT>
// (two workgroups per CU - two waves per SIMD, 256 registers each - where two sets of exchange buffers fit the LDS; one with the
// whole register file otherwise: 16 hidden tiles x 16 state k-steps)
__global__ void __launch_bounds__(256)
    __attribute__((amdgpu_waves_per_eu(2 * coop_grad_lds_bytes(HT, ZR, NT, CR) <= 160 * 1024 ? 2 : 1, 2 * coop_grad_lds_bytes(HT, ZR, NT, CR) <= 160 * 1024 ? 2 : 1)))
coop_grad_step_kernel(CGArgs a) {
    static_assert(ACT == CNF_ACT_TANH_PRESCALED || ACT == CNF_ACT_SOFTPLUS, "act' and act'' are rebuilt from h: tanh and softplus");
    static_assert(L == 2 || L == 3, "two or three hidden layers");
    constexpr int SUP = 16 * NT, CT = 2 * NT;
    constexpr MfmaLayout LAY(HT, L, ZR, CR, true);
    constexpr int MTW = HT / 4, DT = (ZR + 3) / 4, KGC = (CR + 3) / 4, XB = HT * CT * 64, DB2 = DT * CT * 64;
    constexpr int IMG = MfmaLayout::imgA(HT, HT);
    constexpr bool GS = MTW % 2 == 0;                 // operand stores as full 128-byte lines (two row tiles x 8 samples)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xbuf = reinterpret_cast<f32x4*>(smem);   // [2][HT][CT][64]: exchange buffers (column tile = chain * NT + sample tile)
    f32x4* zebuf = xbuf + 2 * XB;                    // [DT][CT][64]: [z_stage | gbar]
    f32x4* ekbuf = zebuf + DB2;                      // [DT][CT][64]: [eps | kbar]
    f32x4* gbuf = ekbuf + DB2;                       // [DT][NT][64]: gbar (for the second dbar_1 product of the stage)
    f32x4* ybuf = gbuf + DT * NT * 64;               // [KGC][NT][64]: conditions (constant over the solve)
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mt0 = wave * MTW;
    const int D = a.D, H = a.H;
    const long long B = a.B;
    const long long nst = (B + SUP - 1) / SUP;
    const bool owner = wave < NT;
    const float* P = a.packed;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
#define AIMG(X) AImg{rP, (unsigned)(X) * 4u, lane16, nullptr}
    const float inv_fs = ACT == CNF_ACT_TANH_PRESCALED ? 1.f / kTanhPrescale : 1.f;   // the forward images of tanh nets carry the pre-scale
    const int ns = a.T.ns < NS ? a.T.ns : NS;
    const float dt = a.dt, tn = a.tn;
    const long long nsB = (long long)ns * B;
    using T1 = f32x4[MTW][NT];
    using T2 = f32x4[MTW][CT];

    auto publish2 = [&](int buf, const T1& c0, const T1& c1) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                xbuf[buf * XB + ((mt0 + m) * CT + q) * 64 + lane] = c0[m][q];
                xbuf[buf * XB + ((mt0 + m) * CT + NT + q) * 64 + lane] = c1[m][q];
            }
    };
    // this wave's tiles of chain `ch` in exchange buffer `buf` -> rows [16 mt0, 16 (mt0 + MTW)) of a column-major operand array
    auto gstore = [&](int buf, int ch, const __amdgpu_buffer_rsrc_t& rs, const unsigned (&voff)[NT][2], unsigned soff0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        typedef float __attribute__((may_alias)) float_a;
        const float_a* xb = reinterpret_cast<const float_a*>(xbuf + buf * XB);
        if constexpr (GS) {
            // ds_read_b32 banks are (word % 32) per 32-lane half: a half's lanes hold 4 samples x 2 row tiles x 4 row quads, and two
            // row tiles are a multiple of 32 words apart - the lanes of the second row tile therefore take the samples 4 further
            // on (s8 ^ 4: banks + 16), which made 2-way conflicts of every one of these reads (22 % of the kernel's LDS cycles,
            // profiles/r3/r3T_cfg4_grad_pmc_summary.txt); the store offsets vox / voy follow the same assignment
            const int mm = (lane >> 2) & 1, gg = lane & 3, s8 = (lane >> 3) ^ (4 * mm);
#pragma unroll
            for (int m = 0; m < MTW; m += 2)
#pragma unroll
                for (int q = 0; q < NT; ++q)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int base = (((mt0 + m + mm) * CT + ch * NT + q) * 64 + s8 + 8 * hf) * 4 + gg;
                        f32x4 v;
                        v[0] = xb[base]; v[1] = xb[base + 64]; v[2] = xb[base + 128]; v[3] = xb[base + 192];
                        const unsigned so = soff0 + (unsigned)(16 * (mt0 + m)) * 4u;
                        const unsigned vo = (16 * (mt0 + m + mm) + 4 * gg < H) ? voff[q][hf] : 0xffffffffu;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)vo, (int)so, CG_NT_AUX);
                        CNF_STORE_DATA_HAZARD(v);
                    }
        } else {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    const int base = (((mt0 + m) * CT + ch * NT + q) * 64 + n) * 4 + g;
                    f32x4 v;
                    v[0] = xb[base]; v[1] = xb[base + 64]; v[2] = xb[base + 128]; v[3] = xb[base + 192];
                    const unsigned so = soff0 + (unsigned)(16 * (mt0 + m)) * 4u;
                    const unsigned vo = (16 * (mt0 + m) + 4 * g < H) ? voff[q][0] : 0xffffffffu;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)vo, (int)so, CG_NT_AUX);
                    CNF_STORE_DATA_HAZARD(v);
                }
        }
        asm volatile("" ::: "memory");
    };
    float* scr = a.scratch + (long long)blockIdx.x * a.scratch_stride;
    auto sstore = [&](int slot, const T1& v) {
        f32x4* s4 = reinterpret_cast<f32x4*>(scr) + ((slot * 4 + wave) * MTW * NT) * 64 + lane;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) CG_NT_STORE(v[m][q], &s4[(m * NT + q) * 64]);
    };
    auto sload = [&](int slot, T1& v) {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(scr) + ((slot * 4 + wave) * MTW * NT) * 64 + lane;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) v[m][q] = CG_NT_LOAD(&s4[(m * NT + q) * 64]);
    };
    auto zero2 = [&](T2& v) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < CT; ++q) v[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto pre_a = [&](int img, int KG, f32x4 (&afr)[MTW]) { coop_load_a<MTW>(AIMG(img), mt0, KG, 0, afr); };
    // hidden k-groups that are not zero padding, rounded up to even (the k-loops run two k-groups per iteration)
    const int KHE = (H + 15) / 16 < HT ? (((H + 15) / 16 + 1) & ~1) : HT;
    auto run2 = [&](int img, int KG, const f32x4* bimg, f32x4 (&afr)[MTW], T2& acc) {   // KG == HT: an H x H image
        if (KG == HT) coop_gemm_rt<MTW, CT, CT>(AIMG(img), mt0, HT, KHE, bimg, 0, lane, afr, acc);
        else coop_gemm<MTW, CT, CT>(AIMG(img), mt0, KG, bimg, 0, lane, afr, acc);
    };
    f32x4 fa1[MTW], fb0[CT], fb1[CT];
    auto head2 = [&](int img, int KG, const f32x4* bimg, f32x4 (&afr)[MTW], T2& acc) {
        coop_gemm_head<MTW, CT, CT>(AIMG(img), mt0, KG, KG == HT ? KHE : KG, bimg, lane, afr, fa1, fb0, fb1, acc);
    };
    auto tail2 = [&](int KG, f32x4 (&afr)[MTW], T2& acc) { coop_gemm_tail<MTW, CT>(KG == HT ? KHE : KG, afr, fa1, fb0, fb1, acc); };
    // dense D-row registers of this wave's sample tile -> column tile `ct` of a B image with `nct` column tiles per k-group
    auto publish_dense = [&](f32x4* img, int nct, int ct, const float (&v)[ZR]) {
#pragma unroll
        for (int kg = 0; kg < DT; ++kg) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (4 * kg + j < ZR) ? v[(4 * kg + j) < ZR ? 4 * kg + j : 0] : 0.f;
            img[(kg * nct + ct) * 64 + lane] = o;
        }
    };
    auto dense_store = [&](float* arr, int ld, long long col, const float (&v)[ZR]) {
#pragma unroll
        for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f < D) arr[col * (long long)ld + f] = v[s]; }
    };

    // scratch slots (tile-native, one chain's tile set each): h_l (l = 1 .. L-1), dbar_l (l = 2 .. L-1)
    constexpr int SLOT_H = 0, SLOT_DB = L - 1;
    const unsigned ldx = (unsigned)H * 4u, ldy = (unsigned)a.ldy * 4u;
    __amdgpu_buffer_rsrc_t rx[L], ry[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        rx[l] = __builtin_amdgcn_make_buffer_rsrc(a.xh[l], 0, (int)((long long)ldx * 2 * nsB), 0x00020000);
        ry[l] = __builtin_amdgcn_make_buffer_rsrc(a.yh[l], 0, (int)((long long)ldy * 2 * nsB), 0x00020000);
    }

    for (long long st = blockIdx.x; st < nst; st += gridDim.x) {
        const long long smp0 = st * SUP;
        const long long smp = smp0 + (owner ? wave : 0) * 16 + n;
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
        unsigned vox[NT][2], voy[NT][2];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const long long sq = GS ? smp0 + q * 16 + ((lane >> 3) ^ (4 * ((lane >> 2) & 1))) + 8 * hf : smp0 + q * 16 + n;
                const unsigned ro = GS ? 16u * (unsigned)(lane & 3) + 64u * (unsigned)((lane >> 2) & 1) : 16u * (unsigned)g;
                vox[q][hf] = sq < B ? (unsigned)sq * ldx + ro : 0xffffffffu;
                voy[q][hf] = sq < B ? (unsigned)sq * ldy + ro : 0xffffffffu;
            }
        }
        CG_SYNC();                 // the previous super-tile's readers of the LDS images are done
        float ycond[CR > 0 ? CR : 1];
        ycond[0] = 0.f;
        if constexpr (CR > 0) {
#pragma unroll
            for (int s = 0; s < CR; ++s) { const int f = 4 * s + g; ycond[s] = f < a.C ? a.ys[sc * a.C + f] : 0.f; }
            if (owner) {
#pragma unroll
                for (int kg = 0; kg < KGC; ++kg) {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (4 * kg + j < CR) ? ycond[(4 * kg + j) < CR ? 4 * kg + j : 0] : 0.f;
                    ybuf[(kg * NT + wave) * 64 + lane] = o;
                }
            }
        }
        float* zbt = a.zb + (tile * 64 + lane) * (long long)(NS * ZR);

#pragma clang loop unroll(disable)
        for (int i = ns - 1; i >= 0; --i) {
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
            const float cl = valid ? dt * bi : 0.f;      // cotangent of ldot (dL/d dlogp = +1 per column); zero for padding columns
            // gbar = cotangent of g = eps^T J: -c_l eps (+ c_n g / |g|);  kbar += c_E zdot / |zdot|  (src/core/icnf.jl:184-251: Edot =
            // |zdot|, ndot = |eps^T J|; zdot_i and g_i of the stage are the forward solve's checkpoints)
            float gbar[ZR];
#pragma unroll
            for (int s = 0; s < ZR; ++s) gbar[s] = -cl * eps[s];
            if (a.lam1 != 0.f) {
                const float* ki = a.ckpt_k + ((((long long)a.step * ns + i) * ntp + tile) * 64 + lane) * ZR;
                float e2 = 0.f;
#pragma unroll
                for (int s = 0; s < ZR; ++s) e2 = fmaf(ki[s], ki[s], e2);
                e2 = group_sum(e2);
                const float inv = e2 > 0.f ? cl * a.lam1 * rsqrtf(e2) : 0.f;
#pragma unroll
                for (int s = 0; s < ZR; ++s) kbar[s] = fmaf(inv, ki[s], kbar[s]);
            }
            if (a.lam2 != 0.f) {
                const float* gi = a.ckpt_g + ((((long long)a.step * ns + i) * ntp + tile) * 64 + lane) * ZR;
                float n2 = 0.f;
#pragma unroll
                for (int s = 0; s < ZR; ++s) n2 = fmaf(gi[s], gi[s], n2);
                n2 = group_sum(n2);
                const float inv = n2 > 0.f ? cl * a.lam2 * rsqrtf(n2) : 0.f;
#pragma unroll
                for (int s = 0; s < ZR; ++s) gbar[s] = fmaf(inv, gi[s], gbar[s]);
            }
            const float tt = tn + a.T.c[i] * dt;
            const long long c1 = (long long)i * B, c2 = nsB + (long long)i * B;
            const unsigned sx1 = (unsigned)c1 * ldx, sx2 = (unsigned)c2 * ldx, sy1 = (unsigned)c1 * ldy, sy2 = (unsigned)c2 * ldy;
            if (owner) {
                publish_dense(zebuf, CT, wave, zs); publish_dense(zebuf, CT, NT + wave, gbar);
                publish_dense(ekbuf, CT, wave, eps); publish_dense(ekbuf, CT, NT + wave, kbar);
                publish_dense(gbuf, NT, wave, gbar);
            }
            if (valid) {
                dense_store(a.y1, a.ld_y1, c1 + smp, gbar);
                dense_store(a.y1, a.ld_y1, c2 + smp, zs);
                if (g == 0) {
                    float* col = a.y1 + (c2 + smp) * (long long)a.ld_y1;
                    if (!a.autonomous) col[D] = tt;
                    col[a.ld_y1 - 1] = 1.f;
                }
                if constexpr (CR > 0) {   // [z; t; ys; 1]: the condition rows (src/core/base_icnf.jl:49-60, cond_layer.jl:7-31)
                    float* col = a.y1 + (c2 + smp) * (long long)a.ld_y1 + D + (a.autonomous ? 0 : 1);
#pragma unroll
                    for (int s = 0; s < CR; ++s) { const int f = 4 * s + g; if (f < a.C) col[f] = ycond[s]; }
                }
                dense_store(a.xN, D, c1 + smp, eps);
                dense_store(a.xN, D, c2 + smp, kbar);
            }
            f32x4 afr[MTW];
            T2 acc;
            // ================= up: [a_{l+1} | dbar_{l+1}] = W_{l+1} [h_l | vbar_l]  (forward images carry the pre-scale) =================
            {
                f32x4 bias[MTW], wt[MTW];
                pre_a(LAY.f1z, DT, afr);
                gload_cvec<MTW>(P + LAY.v_b1, mt0, g, bias);
                gload_cvec<MTW>(P + LAY.v_w1t, mt0, g, wt);
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const f32x4 b0 = a.autonomous ? bias[m] : tile_fma(wt[m], tt, bias[m]);
#pragma unroll
                    for (int q = 0; q < NT; ++q) { acc[m][q] = b0; acc[m][NT + q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                }
                CG_SYNC();
                run2(LAY.f1z, DT, zebuf, afr, acc);                          // [a_1 | dbar_1], dbar_1 = W_1[:,0:D] gbar
                if constexpr (CR > 0) {   // a_1 += W_1[:, conditions] ys (first chain only)
                    T1 ya;
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) ya[m][q] = acc[m][q];
                    f32x4 afy[MTW];
                    coop_load_a<MTW>(AIMG(LAY.f1y), mt0, KGC, 0, afy);
                    coop_gemm<MTW, NT, NT>(AIMG(LAY.f1y), mt0, KGC, ybuf, 0, lane, afy, ya);
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) acc[m][q] = ya[m][q];
                }
            }
            int cur = 0;
#pragma unroll
            for (int l = 0; l < L; ++l) {      // layer l + 1
                T1 h, vb, db;
                if (l + 1 < L) pre_a(LAY.fh + l * IMG, HT, afr);
                else pre_a(LAY.bN, DT, afr);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        f32x4 dd;
                        act_tile<ACT>(acc[m][q], h[m][q], dd);
                        db[m][q] = acc[m][NT + q] * inv_fs;                   // dbar_{l+1} = W_{l+1} vbar_l (dbar_1 = W_1[:,0:D] gbar)
                        vb[m][q] = db[m][q] * act_d<ACT>(h[m][q]);             // vbar_{l+1} (cbar at the top)
                    }
                publish2(cur, h, vb);
                if (l + 1 < L) {
                    sstore(SLOT_H + l, h);
                    if (l > 0) sstore(SLOT_DB + l - 1, db);
                }
                // The operand stores of a published pair are issued INSIDE the product that reads it, behind its last fragment
                // requests (round 5): vmcnt retires in order, so in front of the product (round 3) its first fragment wait drained
                // 8 KB of HBM stores per wave, six times a stage (108.4 against 108.2 ms at cfg4: the A/B build was deleted in round 6).
                if (l + 1 < L) {
                    f32x4 bnx[MTW];
                    gload_cvec<MTW>(P + LAY.v_bh + l * MfmaLayout::vecC(HT), mt0, g, bnx);
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) { acc[m][q] = bnx[m]; acc[m][NT + q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    CG_SYNC();
                    head2(LAY.fh + l * IMG, HT, xbuf + cur * XB, afr, acc);
                    gstore(cur, 0, ry[l], voy, sy2);                          // [h_{l+1}; 1] half of Y_{l+1}
                    gstore(cur, 1, ry[l], voy, sy1);                          // [vbar_{l+1}; 0] half
                    tail2(HT, afr, acc);
                    cur ^= 1;
                } else {
                    gstore(cur, 0, ry[l], voy, sy2);                          // (the top: the pair is overwritten below)
                    gstore(cur, 1, ry[l], voy, sy1);
                    // ===== the top: [c | hbar_L] = W_N^T [eps | kbar]; delta_L = c .* act'_L, a2_L = dbar_L .* c,
                    //       sbar_L = hbar_L .* act'_L + a2_L .* act''_L - all while h_L and dbar_L are in registers =====
                    T2 t;
                    zero2(t);
                    run2(LAY.bN, DT, ekbuf, afr, t);
                    pre_a(LAY.bh + (L - 2) * IMG, HT, afr);
                    T1 dl, sb;
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) {
                            const f32x4 d = act_d<ACT>(h[m][q]), c = t[m][q];
                            dl[m][q] = c * d;
                            sb[m][q] = t[m][NT + q] * d + (db[m][q] * c) * act_dd<ACT>(h[m][q], d);
                        }
                    // own tiles of the Y_L pair have been stored by this wave (gstore waits for its LDS reads): overwrite them
                    publish2(cur, dl, sb);
                    CG_SYNC();
                }
            }
            // ================= down: [u_l | hbar_l] = W_{l+1}^T [delta_{l+1} | sbar_{l+1}] =================
            f32x4 afd[DT];
            const int kq0z = wave * KHE / 4;      // this wave's first k-group of the Zbar product (split along K, below)
#pragma unroll
            for (int l = L - 1; l >= 1; --l) {
                T2 t;
                T1 hl, dbl, dl, sb;
                zero2(t);
                if (l == 1) {
                    // dbar_1 = W_1[:,0:D] gbar again: a D-sized product, cheaper than a scratch round trip
                    f32x4 afq[MTW];
                    coop_load_a<MTW>(AIMG(LAY.f1z), mt0, DT, 0, afq);
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q) dbl[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    coop_gemm<MTW, NT, NT>(AIMG(LAY.f1z), mt0, DT, gbuf, 0, lane, afq, dbl);
                }
                head2(LAY.bh + (l - 1) * IMG, HT, xbuf + cur * XB, afr, t);
                gstore(cur, 0, rx[l], vox, sx1);                              // delta_{l+1} half of X_{l+1}: the pair this product reads
                gstore(cur, 1, rx[l], vox, sx2);                              // sbar_{l+1} half
                sload(SLOT_H + l - 1, hl);
                if (l > 1) sload(SLOT_DB + l - 2, dbl);
                tail2(HT, afr, t);
                if (l > 1) pre_a(LAY.bh + (l - 2) * IMG, HT, afr);
                else coop_load_a<DT>(AIMG(LAY.b1), 0, HT, kq0z < HT ? kq0z : HT - 1, afd);   // first fragments of this wave's share of the Zbar product
                cur ^= 1;
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const f32x4 d = act_d<ACT>(hl[m][q]), u = t[m][q];
                        const f32x4 dbv = l == 1 ? dbl[m][q] * inv_fs : dbl[m][q];
                        dl[m][q] = u * d;
                        sb[m][q] = t[m][NT + q] * d + (dbv * u) * act_dd<ACT>(hl[m][q], d);
                    }
                publish2(cur, dl, sb);
                CG_SYNC();
            }
            // Zbar_i = W_1[:,0:D]^T sbar_1: D rows x the sample tiles, K = H.  SPLIT ALONG K over the four waves (wave w: k-groups
            // [w KHE / 4, (w + 1) KHE / 4)) - run by the owner waves alone it is a quarter of a stage of the reference's default
            // architecture (D ~ H / 4, two layers) during which the other waves wait.  The partial tiles meet in the exchange
            // buffer no product reads any more; the owner adds them in wave order.
            {
                f32x4 zacc[DT][NT];
#pragma unroll
                for (int m = 0; m < DT; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q) zacc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int kq1 = (wave + 1) * KHE / 4;
                f32x4 bz[NT];
                coop_load_b<NT, CT>(xbuf + cur * XB, NT, kq0z < HT ? kq0z : HT - 1, lane, bz);
#pragma clang loop unroll(disable)
                for (int kg = kq0z; kg < kq1; ++kg) {
                    const int kn = kg + 1 < kq1 ? kg + 1 : kq1 - 1;   // clamped prefetch (no load under a branch)
                    f32x4 an[DT], bn[NT];
                    coop_load_a<DT>(AIMG(LAY.b1), 0, HT, kn, an);
                    coop_load_b<NT, CT>(xbuf + cur * XB, NT, kn, lane, bn);
                    coop_frag_mfma<DT, NT>(afd, bz, zacc);
#pragma unroll
                    for (int m = 0; m < DT; ++m) afd[m] = an[m];
#pragma unroll
                    for (int q = 0; q < NT; ++q) bz[q] = bn[q];
                }
                gstore(cur, 0, rx[0], vox, sx1);                              // X_1 = [delta_1 | sbar_1], behind the product that read it
                gstore(cur, 1, rx[0], vox, sx2);
                f32x4* pz = xbuf + (cur ^ 1) * XB;        // [wave][DT][NT][64] partial tiles
#pragma unroll
                for (int m = 0; m < DT; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q) pz[((wave * DT + m) * NT + q) * 64 + lane] = zacc[m][q];
            }
            CG_SYNC();
            if (owner) {
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    const f32x4* pz = xbuf + (cur ^ 1) * XB + (((s >> 2)) * NT + wave) * 64 + lane;
                    zbt[i * ZR + s] = ((pz[0 * DT * NT * 64][s & 3] + pz[1 * DT * NT * 64][s & 3]) + pz[2 * DT * NT * 64][s & 3]) + pz[3 * DT * NT * 64][s & 3];
                }
            }
            // (no barrier here: the next stage writes these buffers only behind its own first barrier, which the owner reaches
            // after this sum; the super-tile loop starts with one)
        }
        if (owner) {
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                float acc = lam[s];
                for (int j = 0; j < ns; ++j) acc += zbt[j * ZR + s];
                lam[s] = acc;
                a.lam[(tile * 64 + lane) * ZR + s] = acc;
            }
        }
        if (a.step == 0 && a.grad_x && valid) {
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
template <int HT, int L, int ZR, int CR, int ACT, int NS, int NT>
static hipError_t launch_grad_step(const CGArgs& a, int num_cus, hipStream_t st) {
    constexpr int lds = coop_grad_lds_bytes(HT, ZR, NT, CR);
    static_assert(lds <= 160 * 1024, "exchange buffers exceed LDS");
    const int nblocks = coop_grad_nblocks(a.B, num_cus, HT, ZR, CR, NT);
    auto kern = coop_grad_step_kernel<HT, L, ZR, CR, ACT, NS, NT>;
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

#define CG_INSTC(HT, L, ZR, CR) \
    CoopGradInst { HT, L, ZR, CR, CG_ACT, 1, { &launch_grad_step<HT, L, ZR, CR, CG_ACT, 4, 1>, &launch_grad_step<HT, L, ZR, CR, CG_ACT, 6, 1> } }
// (NT = 2 - 32-sample super-tiles, one workgroup per CU, each weight fragment feeding four column tiles - was measured on the
// default architecture at nvariables = 16 / 20: 66.6 / 75.2 ms against 66.0 / 74.5 ms with NT = 1; not instantiated)
#define CG_INST(HT, L, ZR) CG_INSTC(HT, L, ZR, 0)
// the (HT, L, ZR) of the forward plans they pair with (the plan's packed image is shared): cnf_coop.hip's instances and the
// unconditioned shapes of cnf_coop_x.hip (hidden tiles 8 / 12 / 16, 8 or 16 state k-steps)
#define CG_XSHAPES(HT) CG_INST(HT, 3, 8), CG_INST(HT, 2, 8), CG_INST(HT, 3, 16), CG_INST(HT, 2, 16), \
                       CG_INSTC(HT, 3, 8, 4), CG_INSTC(HT, 2, 8, 4), CG_INSTC(HT, 3, 16, 4), CG_INSTC(HT, 2, 16, 4)   /* <= 16 conditions */
const CoopGradInst* CG_TABLE_FN(int* n) {
    static const CoopGradInst table[] = {
#ifndef CG_ACT_SOFTPLUS
        CG_INST(8, 3, 2),    // D <= 8, 3 x 128
        CG_INST(4, 3, 2),    // D <= 8, 3 x 64: cross-check of the register-accumulator kernel (cnf_grad.hip)
#endif
        CG_XSHAPES(8), CG_XSHAPES(12), CG_XSHAPES(16),   // (16, 3, 8) is cfg4: D = 32, 3 x 256
        CG_INST(20, 3, 24), CG_INST(20, 2, 24), CG_INST(24, 3, 24), CG_INST(24, 2, 24),   // H <= 320 / 384, D <= 96
    };
    *n = (int)(sizeof(table) / sizeof(table[0]));
    return table;
}

#ifndef CG_ACT_SOFTPLUS
static const CoopGradInst* cg_find(int HT, int L, int ZR, int CR, int ACT) {
    int n = 0;
    const bool sp = ACT == CNF_ACT_SOFTPLUS;
    if (!sp && ACT != CNF_ACT_TANH && ACT != CNF_ACT_TANH_PRESCALED) return nullptr;
    const CoopGradInst* t = sp ? coop_grad_table_softplus(&n) : coop_grad_table_tanh(&n);
    for (int i = 0; i < n; ++i)
        if (t[i].HT == HT && t[i].L == L && t[i].ZR == ZR && t[i].CR == CR) return &t[i];
    return nullptr;
}
int coop_grad_nt(int HT, int L, int ZR, int CR, int ACT) {
    const CoopGradInst* c = cg_find(HT, L, ZR, CR, ACT);
    return c ? c->NT : 1;
}

bool coop_grad_supported(int HT, int L, int ZR, int CR, int ACT) { return cg_find(HT, L, ZR, CR, ACT) != nullptr; }
int coop_grad_scratch_slots(int L) { return 2 * L - 3; }   // h_1 .. h_{L-1}, dbar_2 .. dbar_{L-1}
// workgroups of a launch (16-sample super-tiles; two workgroups per CU where two sets of exchange buffers fit): the host sizes
// the per-workgroup scratch with it
int coop_grad_nblocks(long long B, int num_cus, int HT, int ZR, int CR, int NT) {
    const int one = tuning().cg_one_per_cu == 1 ? 1 : 0;
    const long long nst = (B + 16 * NT - 1) / (16 * NT), cap = (long long)num_cus * ((!one && 2 * coop_grad_lds_bytes(HT, ZR, NT, CR) <= 160 * 1024) ? 2 : 1);
    return (int)(nst < cap ? nst : cap);
}

hipError_t coop_grad_step_launch(int HT, int L, int ZR, int CR, int ACT, const CGArgs& a, int num_cus, hipStream_t st) {
    const CoopGradInst* c = cg_find(HT, L, ZR, CR, ACT);
    if (!c) return hipErrorNotSupported;
    return c->fn[a.T.ns <= 4 ? 0 : 1](a, num_cus, st);
}
#endif

}  // namespace cnf
