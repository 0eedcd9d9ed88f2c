// cnf_mfma.hip — fused whole-solve kernels for gfx950: the Dense chain, its pullback
// (Hutchinson eps^T J) and the fixed-step Runge-Kutta loop in ONE launch per solve.
//
// Replaces, for MatrixMode + a fixed-step solver, the reference's
//   inference_prob -> solve(ODEProblem) { augmented_f -> icnf_jacobian -> Lux Chain + Zygote } -> inference_sol
// (src/core/base_icnf.jl:247-296,134-172; src/core/icnf.jl:517-559; src/core/utils.jl:150-159).
//
// Design:
//   * one wave owns a tile of 16 samples for the whole solve: state z, dlogp, E, n and the RK
//     stage derivatives stay in registers; HBM is touched once at the start (x, eps, ys) and
//     once at the end (logp, regs, optional final state).
//   * every weight product runs on v_mfma_f32_16x16x4_f32 (exact f32, = fmaf chain) with the
//     sample tile on N; an accumulator tile is the next product's B operand with no data
//     movement (row permutation explained in cnf_mfma_layout.h).
//   * weights (forward and transposed images) are staged once per workgroup into LDS in
//     MFMA-operand order and read with conflict-free ds_read_b128 (one read per 4 MFMAs).
//   * eps^T J eps, |zdot|, |eps^T J| are reduced over the 4 lane groups that share a sample with
//     two cross-lane exchanges.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CNF_WITH_DEVICE_CONTROLLER 1
#include "cnf_mfma_kernel.h"
#include "cnf_coop_grad.h"

namespace cnf {

// ---------------------------------------------------------------------------------------
// plan: which instantiation serves a configuration, and how to pack its weights
// ---------------------------------------------------------------------------------------
struct MfmaPlan {
    int HT, L, ZR, CR, ACT, ENGINE, KP;
    bool with_bwd;
    MfmaLayout lay;
    LaunchFn launch;
    LaunchAdaptFn launch_adapt;   // adaptive Tsit5 with the step controller on the device, or null
    int adapt_per_cu = -1;        // workgroups of that kernel per compute unit (occupancy query, cached)
    LaunchAdaptFn launch_vcabm;   // the default solver VCABM on the device (256-thread workgroups), or null
    int vcabm_per_cu = -1;
    cnf_config cfg;
    int nthreads;
    int num_cus;
    int kind;           // 0: per-wave LDS-resident kernel, 1: cooperative wide-layer kernel, 2: its extended form (cnf_coop_x.hip)
    int arith;          // CNF_ARITH_* of the hidden products
    float fwd_scale;    // factor folded into the forward hidden-layer images/biases (pre-scaled tanh)
    int prio_mode;      // see KArgs
    int use_queue;
    int* queue_dev;     // one int per plan, zeroed on the stream before every launch
    float* rk_dev = nullptr;   // ring of the Runge-Kutta sums of cnf_coop_d2.hip's 20 .. 24-tile instances (KArgs::rk), allocated on first use
    int pre = 0;        // the instance's hoisting level (2: the trace term as a dot with q = W_1 eps where |eps^T J| is not asked for)
    int q_extra = 0;    // extended cooperative plans of two-hidden-layer exact-trace flows: float offset of the Q image appended
                        // behind the layout (0 = none); per-wave plans carry theirs inside the layout (lay.qtr)

    char name[128];
    MfmaPlan() : lay(1, 2, 1, 0, true) {}
};

// Instantiations.  The first entry that matches (shape, engine, K) and the requested thread count
// (CNF_MFMA_NT, default: first match) serves the configuration; everything else runs on the
// generic SIMT path.
static const Inst kInsts[] = {
    // --- Hutchinson VJP (LuxVecJacMatrixMode + TrainMode) ---
    MFMA_INST_AD(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 1, 2, 512),      // cfg2 / cfg2': D=8, 3x64, K=1, FFJORD
    MFMA_INST_AD(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 1, 1, 512),      // same shape with reg_j (RNODE, K=1)
    MFMA_INST(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 1, 0, 512),      // no hoisting (A/B reference)
    MFMA_INST(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 1, 2, 1024),
    MFMA_INST(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 1, 2, 256),
    MFMA_INST(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 4, 1, 512),      // cfg3: RNODE K=4, c_k = W_N^T eps_k hoisted, probes unrolled
    MFMA_INST(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 4, 0, 512),      // the same with the rolled probe loop, no hoisting (A/B: CNF_MFMA_PRE=0)
    MFMA_INST_AD(2, 2, 1, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 1, 2, 512),      // cfg1: D=2, 2x32
    MFMA_INST_AD(2, 2, 1, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 1, 1, 512),
    MFMA_INST_AD(1, 2, 1, 0, CNF_ACT_SOFTPLUS, ENG_VJP, 1, 1, 512),  // reference default net, nvariables=1
    // --- split-bf16 hidden products (cnf_config.arith = CNF_ARITH_BF16X6), headline shape ---
    MFMA_INST_BF16X6(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 1, 2, 512),
    MFMA_INST_BF16X6(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 1, 1, 512),
    MFMA_INST_BF16X6(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 4, 1, 512),   // cfg3 with c_k hoisted, probes unrolled
    MFMA_INST_BF16X6(4, 3, 2, 0, CNF_ACT_TANH_PRESCALED, ENG_VJP, 4, 0, 512),
    MFMA_INST_BF16X6(4, 3, 2, 0, CNF_ACT_TANH, ENG_TAN, 1, 0, 512),
    // --- tangent engine: Hutchinson JVP (LuxJacVecMatrixMode) and exact trace (TestMode) ---
    MFMA_INST_AD(8, 3, 2, 2, CNF_ACT_TANH, ENG_TAN, 1, 0, 256),      // cfg5: D=8, C=8, 3x128 (200 VGPR, 1 wave/SIMD:
    MFMA_INST(8, 3, 2, 2, CNF_ACT_TANH, ENG_TAN, 1, 0, 512),      //  13.1 ms; the 512-thread build spills: 23.7 ms)
    MFMA_INST_AD(4, 3, 2, 0, CNF_ACT_TANH, ENG_TAN, 1, 0, 512),      // D=8, 3x64 exact / JVP
    MFMA_INST_AD(2, 2, 1, 0, CNF_ACT_TANH, ENG_TAN, 1, 0, 512),      // D<=4, 2x32 exact / JVP
    MFMA_INST_AD(1, 2, 1, 0, CNF_ACT_SOFTPLUS, ENG_TAN, 1, 0, 512),  // reference default net, TestMode
    MFMA_INST(1, 2, 1, 0, CNF_ACT_TANH, ENG_TAN, 1, 0, 512),
};

// an instance compiled for pre-scaled tanh serves a tanh configuration
static bool act_matches(int inst_act, int cfg_act) {
    return inst_act == cfg_act || (inst_act == CNF_ACT_TANH_PRESCALED && cfg_act == CNF_ACT_TANH);
}

static MfmaPlan* mfma_plan_create_impl(const cnf_config& c, bool coop_only) {
    const int N = c.n_layers, L = N - 1;
    if (L < 1) return nullptr;
    // hidden layers may have different widths: every one is zero-padded to the widest (act(0) of a padded
    // feature only ever meets zero weights), so only the activation has to be common
    int H = 0;
    for (int l = 1; l <= L; ++l) {
        if (c.acts[l - 1] != c.acts[0]) return nullptr;
        if (c.widths[l] > H) H = c.widths[l];
    }
    if (c.acts[N - 1] != CNF_ACT_IDENTITY) return nullptr;
    const int D = c.nvars + c.naug;
    const int HT = (H + 15) / 16, ZR = (D + 3) / 4, CR = (c.ncond + 3) / 4;
    const int engine = c.mode == CNF_MODE_HUTCH_VJP ? ENG_VJP : ENG_TAN;
    const int KP = c.mode == CNF_MODE_EXACT ? 1 : c.nprobes;
    const int want_nt = tuning().mfma_nt;
    const int want_pre = tuning().mfma_pre;
    const bool force_coop = coop_only || tuning().mfma_coop != 0;
    auto make_coop = [&]() -> MfmaPlan* {
        int zr_inst = ZR, ht_inst = HT;
        if (!coop_supported(HT, L, ZR, CR, c.acts[0], engine, KP, &zr_inst, &ht_inst)) return nullptr;
        MfmaPlan* p = new MfmaPlan();
        const int HT = ht_inst;   // the instance's hidden tiles (zero-padded)
        p->HT = HT; p->L = L; p->ZR = zr_inst; p->CR = CR; p->ACT = c.acts[0]; p->ENGINE = engine; p->KP = KP;
        p->with_bwd = true;
        p->lay = MfmaLayout(HT, L, zr_inst, CR, true);
        p->launch = nullptr;
        p->launch_adapt = nullptr;
        p->launch_vcabm = nullptr;
        p->cfg = c;
        p->nthreads = 256;
        p->num_cus = 0;
        p->prio_mode = 0; p->use_queue = 0; p->queue_dev = nullptr;
        p->kind = 1;
        p->arith = 0;
        p->fwd_scale = c.acts[0] == CNF_ACT_TANH ? kTanhPrescale : 1.f;   // cnf_coop.hip runs pre-scaled tanh
        snprintf(p->name, sizeof(p->name), "coop_vjp<HT=%d,L=%d,ZR=%d,act=%d>", HT, L, zr_inst, c.acts[0]);
        return p;
    };
    if (force_coop) return c.arith == CNF_ARITH_F32 ? make_coop() : nullptr;
    constexpr size_t kMaxLds = 160 * 1024;
    auto make = [&](const Inst& in) -> MfmaPlan* {
        MfmaPlan* p = new MfmaPlan();
        const int HT = in.HT;   // the instance's hidden tiles (>= the configuration's: zero-padded)
        p->HT = HT; p->L = L; p->ZR = in.ZR; p->CR = in.CR; p->ACT = in.ACT; p->ENGINE = engine; p->KP = KP;
        p->with_bwd = engine == ENG_VJP;
        p->arith = in.arith;
        p->fwd_scale = in.ACT == CNF_ACT_TANH_PRESCALED ? kTanhPrescale : 1.f;
        p->lay = MfmaLayout(HT, L, in.ZR, in.CR, p->with_bwd, in.arith);
        p->launch = in.fn;
        p->launch_adapt = in.fn_adapt;
        p->launch_vcabm = in.fn_vcabm;
        p->cfg = c;
        p->nthreads = in.nthreads;
        p->pre = in.PRE;
        p->num_cus = 0;
        p->prio_mode = tuning().mfma_prio;
        p->use_queue = tuning().mfma_queue;
        p->queue_dev = nullptr;
        p->kind = 0;
        snprintf(p->name, sizeof(p->name), "mfma_%s<HT=%d,L=%d,ZR=%d,CR=%d,act=%d,K=%d,pre=%d,nt=%d,%s>",
                 engine == ENG_VJP ? "vjp" : "tan", HT, L, in.ZR, in.CR, in.ACT, KP, in.PRE, in.nthreads,
                 in.arith ? "bf16x6" : "f32");
        // whole fixed-step solves of this plan run on the hand-scheduled form of the same kernel (mfma_solve, cnf_mfma2.hip)
        if (engine == ENG_VJP && KP == 1 && in.CR == 0 && !in.arith) {
            const bool s2 = tuning().solve2 != 0 && solve2_supported(HT, L, in.ZR, in.ACT);
            const bool s2p = tuning().solve2_pair != 0 && solve2p_supported(HT, L, in.ZR, in.ACT, 0, 1);
            const size_t n = strlen(p->name);
            if (s2) snprintf(p->name + n, sizeof(p->name) - n, " | solves: mfma_solve2<nt=%d>%s", tuning().solve2 == 1 ? 256 : 512, s2p ? ", mfma_solve2p up to 2 tiles per CU" : "");
            else if (s2p) snprintf(p->name + n, sizeof(p->name) - n, " | solves of up to 2 tiles per CU: mfma_solve2p");
        }
        return p;
    };
    // 1. specialised instances: exact state / condition k-steps
    for (const Inst& in : kInsts) {
        if (in.PRE == 2 && c.reg_j) continue;   // the dot-product shortcut needs no |eps^T J|
        if (in.HT == HT && in.L == L && in.ZR == ZR && in.CR == CR && act_matches(in.ACT, c.acts[0]) &&
            in.ENGINE == engine && in.KP == KP && in.arith == c.arith &&
            (want_nt == 0 || want_nt == in.nthreads) && (want_pre < 0 || want_pre == in.PRE))
            return make(in);
    }
    if (c.arith != CNF_ARITH_F32) return nullptr;   // split-bf16: specialised instances only
    // 2. generic zero-padded instances whose images fit LDS
    if (want_nt == 0 && want_pre < 0) {
        // smallest instance that holds the configuration: hidden tiles, then state k-steps (zero padding
        // costs MFMAs, so the tightest fit wins); K = 1 tables first, then the probe-capacity table
        auto pick = [&](const Inst* gen, int ng, bool probes) -> const Inst* {
            const Inst* best = nullptr;
            for (int i = 0; i < ng; ++i) {
                const Inst& in = gen[i];
                if (in.HT >= HT && in.L == L && in.ZR >= ZR && in.CR >= CR && (CR > 0 || in.CR == 0) &&
                    act_matches(in.ACT, c.acts[0]) && in.ENGINE == engine && (probes ? in.KP >= KP : in.KP == KP) &&
                    (size_t)MfmaLayout(in.HT, L, in.ZR, in.CR, engine == ENG_VJP).lds_total * sizeof(float) <= kMaxLds &&
                    (!best || in.HT < best->HT || (in.HT == best->HT && in.ZR < best->ZR)))
                    best = &in;
            }
            return best;
        };
        int ng = 0, ng8 = 0, ngp = 0;
        const Inst* gen = c.acts[0] == CNF_ACT_SOFTPLUS ? mfma_generic_softplus_insts(&ng) : mfma_generic_insts(&ng);
        const Inst* gen8 = mfma_generic_zr8_insts(&ng8);
        const Inst* a4 = pick(gen, ng, false);
        const Inst* a8 = pick(gen8, ng8, false);
        const Inst* pickd = (a4 && (!a8 || a4->HT <= a8->HT)) ? a4 : a8;
        if (pickd) {
            // exact trace of a two-hidden-layer net whose Q image does not fit LDS beside this instance's images (8 hidden
            // tiles with 8 state k-steps): D tangent passes here would lose to the layer-wise path's single Q GEMM
            if (c.mode == CNF_MODE_EXACT && L == 2 && MfmaLayout(pickd->HT, L, pickd->ZR, pickd->CR, false).qtr < 0 &&
                c.kernel_path == CNF_PATH_AUTO && layered_supports(c))
                return nullptr;
            return make(*pickd);
        }
        if (KP > 1) {
            const Inst* genp = mfma_generic_probe_insts(&ngp);
            if (const Inst* ap = pick(genp, ngp, true)) return make(*ap);
        }
    }
    // 3. cooperative wide-layer kernel (Hutchinson VJP, one probe, no conditions) ...
    if (MfmaPlan* p = make_coop()) return p;
    // 4. ... and its extended form: conditions, several probes, the exact trace as D unit probes (three hidden layers: with two
    //    the layer-wise path's single Q product is cheaper than D pullbacks), Hutchinson JVP (probes pushed through the forward images).
    if (c.arith != CNF_ARITH_F32 || tuning().mfma_coopx == 0) return nullptr;
    const bool exact = c.mode == CNF_MODE_EXACT;
    if (!(c.mode == CNF_MODE_HUTCH_VJP || c.mode == CNF_MODE_HUTCH_JVP || (exact && (L == 3 || L == 2)))) return nullptr;
    if (!exact && (c.nprobes < 1 || c.nprobes > 64)) return nullptr;
    int hti = HT, zri = ZR, cri = CR;
    if (!coopx_supported(HT, L, ZR, CR, c.acts[0], &hti, &zri, &cri)) return nullptr;
    MfmaPlan* p = new MfmaPlan();
    p->HT = hti; p->L = L; p->ZR = zri; p->CR = cri; p->ACT = c.acts[0]; p->ENGINE = ENG_VJP; p->KP = exact ? 1 : c.nprobes;
    p->with_bwd = true;
    p->lay = MfmaLayout(hti, L, zri, cri, true);
    p->launch = nullptr; p->launch_adapt = nullptr; p->launch_vcabm = nullptr;
    p->cfg = c;
    p->nthreads = 256; p->num_cus = 0; p->prio_mode = 0; p->use_queue = 0; p->queue_dev = nullptr;
    p->kind = 2; p->arith = 0;
    // two hidden layers, exact trace: tr J = act'_2^T Q act'_1 - ONE H x H product per evaluation (the Q image sits behind the layout)
    p->q_extra = (exact && L == 2) ? p->lay.total : 0;
    p->fwd_scale = c.acts[0] == CNF_ACT_TANH ? kTanhPrescale : 1.f;
    snprintf(p->name, sizeof(p->name), "coopx<HT=%d,L=%d,ZR=%d,CR=%d,act=%d,%s>", hti, L, zri, cri, c.acts[0],
             exact ? (L == 2 ? "exact (Q product)" : "exact (unit probes)") : c.mode == CNF_MODE_HUTCH_JVP ? "jvp" : "vjp");
    return p;
}

static hipError_t plan_ensure_cus(MfmaPlan* mp);
// The ring of the Runge-Kutta sums some dealt instances keep in device memory (KArgs::rk) is allocated with the plan where a
// device is current, so that the first solve - which may run under stream capture, where hipMalloc is not allowed - finds it;
// mfma_solve still allocates it on first use otherwise.
MfmaPlan* mfma_plan_create(const cnf_config& c, bool coop_only) {
    MfmaPlan* p = mfma_plan_create_impl(c, coop_only);
    if (p && (p->kind == 1 || p->kind == 2) && p->KP == 1 && plan_ensure_cus(p) == hipSuccess) {
        int hmax = 0;
        for (int l = 1; l < c.n_layers; ++l) hmax = c.widths[l] > hmax ? c.widths[l] : hmax;
        const size_t a = coopd_rk_floats(hmax, c.nvars + c.naug, p->L, p->ACT, 0, p->num_cus), b = coopd_rk_floats(hmax, c.nvars + c.naug, p->L, p->ACT, 1, p->num_cus);
        const size_t rkf = a > b ? a : b;
        if (rkf && hipMalloc((void**)&p->rk_dev, rkf * sizeof(float)) != hipSuccess) { p->rk_dev = nullptr; (void)hipGetLastError(); }
    }
    return p;
}

void mfma_plan_destroy(MfmaPlan* p) {
    if (p && p->queue_dev) (void)hipFree(p->queue_dev);
    if (p && p->rk_dev) (void)hipFree(p->rk_dev);
    delete p;
}
size_t mfma_packed_bytes(const MfmaPlan* p) {
    return ((size_t)p->lay.total + (p->q_extra ? (size_t)MfmaLayout::imgA(p->lay.HT, p->lay.HT) : 0)) * sizeof(float);
}
// float offset of the Q image of a two-hidden-layer exact-trace plan, or -1
static long long plan_q_offset(const MfmaPlan* p) { return p->q_extra ? p->q_extra : (p->kind == 0 ? p->lay.qtr : -1); }
const char* mfma_plan_name(const MfmaPlan* p) { return p->name; }

// A image: out[(mt*KG + kg)*256 + lane*4 + j] = A(rowmap(mt, lane&15), 16 kg + 4 j + (lane>>4))
template <typename F>
static void pack_imgA(float* out, int MT, int KG, int M, int K, F A) {
    for (int mt = 0; mt < MT; ++mt)
        for (int kg = 0; kg < KG; ++kg)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int row = mfma_rowmap(mt, lane & 15);
                    const int k = 16 * kg + 4 * j + (lane >> 4);
                    out[((mt * KG + kg) * 64 + lane) * 4 + j] = (row < M && k < K) ? A(row, k) : 0.f;
                }
}

// split-bf16 hidden image (see gemm_hidden_bf16x6): three bf16 parts of every weight, RNE at each level
static inline unsigned short bf16_rne(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_val(unsigned short b) {
    const unsigned u = (unsigned)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
template <typename F>
static void pack_imgH16(float* out, int HT, int M, int K, F A) {
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    const int NC = HT / 2;
    for (int mt = 0; mt < HT; ++mt)
        for (int c = 0; c < NC; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int row = mfma_rowmap(mt, lane & 15);
                    const int k = 16 * (2 * c + (j >> 2)) + 4 * (j & 3) + (lane >> 4);
                    const float w = (row < M && k < K) ? A(row, k) : 0.f;
                    unsigned short part[3];
                    part[0] = bf16_rne(w);
                    const float r1 = w - bf16_val(part[0]);
                    part[1] = bf16_rne(r1);
                    const float r2 = r1 - bf16_val(part[1]);
                    part[2] = bf16_rne(r2);
                    for (int sp = 0; sp < 3; ++sp)
                        o[((((size_t)sp * HT + mt) * NC + c) * 64 + lane) * 8 + j] = part[sp];
                }
}

// C vector: out[(mt*4 + g)*4 + r] = v(16 mt + 4 r + g)
template <typename F>
static void pack_vecC(float* out, int MT, int M, F v) {
    for (int mt = 0; mt < MT; ++mt)
        for (int g = 0; g < 4; ++g)
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * mt + 4 * r + g;
                out[(mt * 4 + g) * 4 + r] = f < M ? v(f) : 0.f;
            }
}

void mfma_pack(const MfmaPlan* p, const float* lux, const size_t* w_off, const size_t* b_off, float* packed) {
    const cnf_config& c = p->cfg;
    const MfmaLayout& Y = p->lay;
    const int N = c.n_layers, L = N - 1, D = c.nvars + c.naug, C = c.ncond;
    auto Hl = [&](int l) { return c.widths[l]; };   // width of hidden layer l (1-based); padded to 16 HT
    const int n_in = c.widths[0];
    const int tcol = D;                           // time column of W1 (if !autonomous)
    const int ycol = D + (c.autonomous ? 0 : 1);  // first cond column
    (void)n_in;
    // Lux Dense weight (out x in) column-major: W(o,i) = lux[w_off + o + out*i]
    auto W = [&](int l, int o, int i) { return lux[w_off[l] + (size_t)o + (size_t)c.widths[l + 1] * i]; };
    auto Bv = [&](int l, int o) { return lux[b_off[l] + o]; };
    const float fs = p->fwd_scale;   // hidden-layer pre-activations are produced pre-scaled (forward images only)
    pack_imgA(packed + Y.f1z, Y.HT, Y.KGZ, Hl(1), D, [&](int r, int k) { return fs * W(0, r, k); });
    if (Y.CR > 0) pack_imgA(packed + Y.f1y, Y.HT, Y.KGC, Hl(1), C, [&](int r, int k) { return fs * W(0, r, ycol + k); });
    for (int l = 1; l < L; ++l) {
        if (Y.arith) pack_imgH16(packed + Y.fh + (l - 1) * Y.imgHid(), Y.HT, Hl(l + 1), Hl(l), [&](int r, int k) { return fs * W(l, r, k); });
        else pack_imgA(packed + Y.fh + (l - 1) * Y.imgHid(), Y.HT, Y.HT, Hl(l + 1), Hl(l), [&](int r, int k) { return fs * W(l, r, k); });
    }
    pack_imgA(packed + Y.fN, Y.DT, Y.HT, D, Hl(L), [&](int r, int k) { return W(L, r, k); });
    if (p->with_bwd) {
        pack_imgA(packed + Y.bN, Y.HT, Y.KGZ, Hl(L), D, [&](int r, int k) { return W(L, k, r); });   // W_N^T
        for (int l = 1; l < L; ++l) {                                                                 // W_l^T
            if (Y.arith) pack_imgH16(packed + Y.bh + (l - 1) * Y.imgHid(), Y.HT, Hl(l), Hl(l + 1), [&](int r, int k) { return W(l, k, r); });
            else pack_imgA(packed + Y.bh + (l - 1) * Y.imgHid(), Y.HT, Y.HT, Hl(l), Hl(l + 1), [&](int r, int k) { return W(l, k, r); });
        }
        pack_imgA(packed + Y.b1, Y.DT, Y.HT, D, Hl(1), [&](int r, int k) { return W(0, k, r); });     // W_1[:,0:D]^T
    }
    pack_vecC(packed + Y.v_b1, Y.HT, Hl(1), [&](int f) { return fs * Bv(0, f); });
    pack_vecC(packed + Y.v_w1t, Y.HT, Hl(1), [&](int f) { return c.autonomous ? 0.f : fs * W(0, f, tcol); });
    for (int l = 1; l < L; ++l)
        pack_vecC(packed + Y.v_bh + (l - 1) * MfmaLayout::vecC(Y.HT), Y.HT, Hl(l + 1), [&](int f) { return fs * Bv(l, f); });
    pack_vecC(packed + Y.v_bN, Y.DT, D, [&](int f) { return Bv(L, f); });
    if (plan_q_offset(p) >= 0) {
        // Q[a][b] = W_2[a][b] * (W_1[:,0:D] W_3)[b][a]: with two hidden layers tr J = sum_ab act'_2[a] Q[a][b] act'_1[b]
        const int H1 = Hl(1), H2 = Hl(2);
        std::vector<double> P((size_t)H1 * H2, 0.0);
        for (int b = 0; b < H1; ++b)
            for (int a = 0; a < H2; ++a) {
                double acc = 0.0;
                for (int i = 0; i < D; ++i) acc += (double)W(0, b, i) * (double)W(2, i, a);
                P[(size_t)b * H2 + a] = acc;
            }
        pack_imgA(packed + plan_q_offset(p), Y.HT, Y.HT, H2, H1, [&](int r, int k) { return (float)((double)W(1, r, k) * P[(size_t)k * H2 + r]); });
    }
    if (Y.v_w1c >= 0)   // columns of W_1[:, 0:D] in accumulator layout (first-layer tangent of a unit seed = a column load)
        for (int i = 0; i < D; ++i)
            pack_vecC(packed + Y.v_w1c + i * MfmaLayout::vecC(Y.HT), Y.HT, Hl(1), [&](int f) { return fs * W(0, f, i); });
    if (Y.v_wNr >= 0)   // rows of W_N in accumulator layout (exact trace reads J_ii off a dot product instead of a last-layer product)
        for (int i = 0; i < D; ++i)
            pack_vecC(packed + Y.v_wNr + i * MfmaLayout::vecC(Y.HT), Y.HT, Hl(L), [&](int f) { return W(L, i, f); });
}

// operand image for the gradient kernel: f32, no tanh pre-scale, forward + transposed
// Device-side pack of the Q image (see mfma_pack): every element is W_2[a][b] * sum_i W_1[b][i] W_3[i][a] in the A-image
// slot of (row a, column b); same arithmetic as the host packer (double accumulation, one rounding).
__global__ void pack_q_kernel(const float* __restrict__ lux, float* __restrict__ img, int HT, int H1, int H2, int D,
                              long long w0, long long w1, long long w2) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= HT * HT * 256) return;
    const int j = e & 3, lane = (e >> 2) & 63, tile = e >> 8, kg = tile % HT, mt = tile / HT;
    const int a = mfma_rowmap(mt, lane & 15), b = 16 * kg + 4 * j + (lane >> 4);
    float v = 0.f;
    if (a < H2 && b < H1) {
        double acc = 0.0;
        for (int i = 0; i < D; ++i) acc += (double)lux[w0 + b + (long long)H1 * i] * (double)lux[w2 + i + (long long)D * a];
        v = (float)((double)lux[w1 + a + (long long)H2 * b] * acc);
    }
    img[e] = v;
}

bool mfma_plan_q_region(const MfmaPlan* p, size_t* off, size_t* len) {
    if (!p || plan_q_offset(p) < 0) return false;
    *off = (size_t)plan_q_offset(p);
    *len = (size_t)MfmaLayout::imgA(p->lay.HT, p->lay.HT);
    return true;
}

hipError_t mfma_pack_q_device(const MfmaPlan* p, const float* lux_dev, const size_t* w_off, float* packed_dev, hipStream_t st) {
    const cnf_config& c = p->cfg;
    const int D = c.nvars + c.naug, H1 = c.widths[1], H2 = c.widths[2], HT = p->lay.HT;
    const int n = HT * HT * 256;
    hipLaunchKernelGGL(pack_q_kernel, dim3((n + 255) / 256), dim3(256), 0, st, lux_dev, packed_dev + plan_q_offset(p), HT, H1, H2, D,
                       (long long)w_off[0], (long long)w_off[1], (long long)w_off[2]);
    return hipGetLastError();
}

int mfma_plan_zr(const MfmaPlan* p) { return p->ZR; }
bool mfma_plan_coop_shape(const MfmaPlan* p, int* HT, int* L, int* ZR, int* ACT) {
    if (!p || p->kind != 1) return false;
    *HT = p->HT; *L = p->L; *ZR = p->ZR; *ACT = p->ACT;
    return true;
}
// a plan whose forward solve can checkpoint for the cooperative gradient: the cooperative kernel, or its extended form in the
// one-probe VJP configuration (layout family MfmaLayout(HT, L, ZR, CR, true); *CR = its condition k-steps)
bool mfma_plan_coop_grad_shape(const MfmaPlan* p, int* HT, int* L, int* ZR, int* ACT, int* CR) {
    if (CR) *CR = 0;
    if (mfma_plan_coop_shape(p, HT, L, ZR, ACT)) return true;
    if (!p || p->kind != 2 || p->KP != 1 || p->cfg.mode != CNF_MODE_HUTCH_VJP) return false;
    if (p->CR != 0 && !CR) return false;
    *HT = p->HT; *L = p->L; *ZR = p->ZR; *ACT = p->ACT;
    if (CR) *CR = p->CR;
    return true;
}
// 16-sample tiles of the checkpoint arrays such a solve writes (the kernel's super-tile count x its tiles per super-tile)
static hipError_t plan_ensure_cus(MfmaPlan* mp) {
    if (mp->num_cus != 0) return hipSuccess;
    int dev = 0;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    mp->num_cus = prop.multiProcessorCount;
    return hipSuccess;
}

// an extended-kernel plan whose one-probe VJP solves of B columns run on the dealt cooperative kernel (cnf_coop_d.hip): its
// 64-sample form from 16 columns per compute unit on (4096) - below that its super-tiles leave CUs empty that the extended
// kernel's 32-sample ones fill (measured at nvariables = 24: 9.7 against 9.0 ms at B <= 4096, 9.7 against 11.7 ms at 8192) -
// its 32-sample form (cnf_coop_d2.hip) at every batch size.  CNF_COOPD=0: never, =2: both forms at any batch size.
static bool plan_uses_coopd(const MfmaPlan* p, long long B) {
    if (!p || (p->kind != 2 && p->kind != 1) || p->KP != 1) return false;   // extended plans, and cooperative ones (one probe, VJP, no conditions)
    const bool exact = p->cfg.mode == CNF_MODE_EXACT;
    if (!(p->cfg.mode == CNF_MODE_HUTCH_VJP || (exact && p->L == 2 && p->q_extra > 0))) return false;
    const int env = tuning().coopd;
    if (env == 0) return false;
    int hmax = 0;
    for (int l = 1; l < p->cfg.n_layers; ++l) hmax = p->cfg.widths[l] > hmax ? p->cfg.widths[l] : hmax;
    // the batch threshold belongs to the 64-sample form; the 32-sample form (16 .. 24 hidden tiles) has the extended kernel's
    // super-tile size and beats it at every batch size (nvariables = 32: 7.7 against 11.6 ms at B = 1024 .. 4096; 40: 11.1 against 15.9)
    if (env != 2 && B <= 16LL * (p->num_cus > 0 ? p->num_cus : 256) &&
        coopd_supertile(hmax, p->cfg.nvars + p->cfg.naug, p->L, p->ACT, exact ? 1 : 0) == 64) return false;
    // a cooperative plan whose hidden width fills whole quads of tiles stays on its own, exact, tuned kernel (3 x 192, D = 20:
    // 20.5 against 22.4 ms); one that it pads - 13 tiles run as 16 - is dealt (3 x 200: 31.0 -> 24.5 ms)
    if (p->kind == 1 && ((hmax + 15) / 16) % 4 == 0) return false;
    return coopd_supported(hmax, p->cfg.nvars + p->cfg.naug, p->L, p->ACT, p->HT, p->ZR, exact ? 1 : 0, p->cfg.ncond);
}

long long mfma_plan_ckpt_tiles(const MfmaPlan* p, long long B, bool on_grid) {
    (void)plan_ensure_cus(const_cast<MfmaPlan*>(p));   // (the batch threshold of the dealt kernel counts compute units)
    if (plan_uses_coopd(p, B)) {
        int hmax = 0;
        for (int l = 1; l < p->cfg.n_layers; ++l) hmax = p->cfg.widths[l] > hmax ? p->cfg.widths[l] : hmax;
        return coopd_supertile(hmax, p->cfg.nvars + p->cfg.naug, p->L, p->ACT, 0) == 64 ? (B + 63) / 64 * 4 : (B + 31) / 32 * 2;
    }
    const bool x = p->kind == 2 || on_grid || !coop_ckpt_supported(p->HT, p->L, p->ZR, p->ACT);   // which kernel checkpoints: see mfma_solve
    return x ? (B + 31) / 32 * 2 : (B + 63) / 64 * 4;
}
// hidden tiles per sample tile of the STAGE STORE (cnf_tiles.h) the plan's checkpointing forward solve writes when SolveArgs::kfull
// is set - h_l and delta_l of every stage, for the second-order reverse sweep - or 0 when the kernel that would serve the solve
// does not write one (then the older sweeps recompute both chains)
int mfma_plan_stage_store_tiles(const MfmaPlan* p, long long B, bool on_grid) {
    if (!p || on_grid) return 0;
    (void)plan_ensure_cus(const_cast<MfmaPlan*>(p));
    if (plan_uses_coopd(p, B)) {   // the dealt kernel in its 64-sample form (cnf_coop_d.hip) stores the real tiles
        int hmax = 0;
        for (int l = 1; l < p->cfg.n_layers; ++l) hmax = p->cfg.widths[l] > hmax ? p->cfg.widths[l] : hmax;
        if (p->cfg.ncond != 0 || coopd_supertile(hmax, p->cfg.nvars + p->cfg.naug, p->L, p->ACT, 0) != 64) return 0;
        return (hmax + 15) / 16;
    }
    if (p->kind == 1 && coop_ckpt_supported(p->HT, p->L, p->ZR, p->ACT)) return p->HT;
    return 0;
}
bool mfma_plan_ckpt_rows_as_tiles(const MfmaPlan* p, long long B, bool on_grid) {
    if (!p || on_grid || p->ZR % 4 != 0) return false;
    (void)plan_ensure_cus(const_cast<MfmaPlan*>(p));
    if (!plan_uses_coopd(p, B)) return false;
    int hmax = 0;
    for (int l = 1; l < p->cfg.n_layers; ++l) hmax = p->cfg.widths[l] > hmax ? p->cfg.widths[l] : hmax;
    return coopd_supertile(hmax, p->cfg.nvars + p->cfg.naug, p->L, p->ACT, 0) == 64;
}
bool mfma_plan_is_per_wave(const MfmaPlan* p) { return p->kind == 0; }
// can this plan's forward solve checkpoint for the cooperative gradient (on the caller's grid when `on_grid`)?  Extended-kernel
// plans always do (run-time switch); cooperative plans through their CK instance on uniform steps, otherwise through the
// extended kernel's instance of exactly their layout (mfma_solve)
bool mfma_plan_can_checkpoint(const MfmaPlan* p, bool on_grid) {
    if (!p) return false;
    if (p->kind == 2) return true;
    if (p->kind != 1) return false;
    if (!on_grid && coop_ckpt_supported(p->HT, p->L, p->ZR, p->ACT)) return true;
    return coopx_exact_supported(p->HT, p->L, p->ZR, 0, p->ACT);
}

// plain f32 image (forward + transposed, no tanh pre-scale) for a given layout: the gradient kernels' operand image
void mfma_pack_layout(const cnf_config& c, int HT, int L, int ZR, int CR, const float* lux, const size_t* w_off,
                      const size_t* b_off, float* packed) {
    MfmaPlan p;
    p.HT = HT; p.L = L; p.ZR = ZR; p.CR = CR; p.with_bwd = true; p.arith = 0; p.fwd_scale = 1.f;
    p.lay = MfmaLayout(HT, L, ZR, CR, true, 0);
    p.cfg = c;
    mfma_pack(&p, lux, w_off, b_off, packed);
}

void grad_pack(const cnf_config& c, const float* lux, const size_t* w_off, const size_t* b_off, float* packed) {
    int HT, L, ZR, CR;
    grad_shape(c, &HT, &L, &ZR, &CR);
    mfma_pack_layout(c, HT, L, ZR, CR, lux, w_off, b_off, packed);
}


// Small batches (at most one 16-sample tile per compute unit) of a per-wave plan: the tile-split form - the tile's hidden width
// over the four SIMDs of a CU (cnf_coop.hip, NT = 1) - instead of one wave per tile with three SIMDs of its CU idle.  Same packed
// image (the layouts coincide: forward + transposed images, pre-scaled tanh), same arithmetic per product; the sums over a
// sample's lane groups are taken in the same order.  Whole fixed-step solves only: single dynamics calls (boundary A, the
// attempts of the adaptive host loops) stay on the per-wave kernel, whose arithmetic the one-launch adaptive kernels share bit
// for bit.  CNF_TILE_SPLIT=0 keeps the per-wave kernel everywhere, =2 forces the split form at any batch size.
static bool plan_takes_tile_split(MfmaPlan* p, long long B) {
    if (p->kind != 0 || plan_ensure_cus(p) != hipSuccess) return false;
    const int split_env = tuning().tile_split;   // (cnf_set_tuning switches it inside one process: tests and A/B runs)
    const long long ntiles = (B + 15) / 16;
    const int D = p->cfg.nvars + p->cfg.naug;
    return split_env > 0 && p->ENGINE == ENG_VJP && p->KP == 1 && p->CR == 0 && p->arith == 0 && p->with_bwd && !p->use_queue &&
           (split_env == 2 || ntiles <= p->num_cus) && p->ZR * 4 >= D && coop_split_supported(p->HT, p->L, p->ZR, p->ACT);
}

// CNF_FAMILY_* of the kernel that serves a whole fixed-step solve of B columns (whole_solve) or a single dynamics call
int mfma_plan_family_for(MfmaPlan* p, long long B, bool whole_solve) {
    if (p->kind == 1) return (B > 0 && plan_ensure_cus(p) == hipSuccess && plan_uses_coopd(p, B)) ? CNF_FAMILY_COOPD : CNF_FAMILY_COOP;
    if (p->kind == 2) return (B > 0 && plan_ensure_cus(p) == hipSuccess && plan_uses_coopd(p, B)) ? CNF_FAMILY_COOPD : CNF_FAMILY_COOPX;
    return (whole_solve && B > 0 && plan_takes_tile_split(p, B)) ? CNF_FAMILY_TILE_SPLIT : CNF_FAMILY_PER_WAVE;
}

hipError_t mfma_solve(MfmaPlan* p, const float* packed_dev, const SolveArgs& s, hipStream_t st) {
    if (s.B == 0) return hipSuccess;
    MfmaPlan* mp = p;   // caches the CU count and owns the optional queue word
    {
        hipError_t e = plan_ensure_cus(mp);
        if (e != hipSuccess) return e;
    }
    KArgs a{};
    a.packed = packed_dev;
    a.x = s.x; a.u0 = s.u0; a.eps = s.eps; a.ys = s.ys;
    a.u_out = s.u_out; a.logp = s.logp; a.regs = s.regs; a.ckpt = s.ckpt; a.ckpt_k = s.ckpt_k; a.kfull = s.kfull; a.ckpt_g = s.ckpt_g; a.tgrid = s.tgrid_dev;
    a.ck_tiles = s.ck_tiles;
    if (s.ck_tiles && !mfma_plan_ckpt_rows_as_tiles(p, s.B, s.tgrid_dev != nullptr)) return hipErrorNotSupported;   // (only that kernel writes the layout)
    a.B = s.B; a.nsteps = s.nsteps; a.t0 = s.t0;
    a.dt = s.nsteps > 0 ? (s.dt_exact != 0.f ? s.dt_exact : (s.t1 - s.t0) / (float)s.nsteps) : 0.f;
    a.nvars = s.nvars; a.D = p->cfg.nvars + p->cfg.naug; a.C = p->cfg.ncond;
    a.reg_z = p->cfg.reg_z; a.reg_j = p->cfg.reg_j; a.reg_aug = s.reg_aug; a.autonomous = p->cfg.autonomous;
    a.T = make_tableau(s.alg);
    for (int st = 0; st < 6; ++st)
        for (int i = 0; i < 5; ++i) a.acol[st][i] = (st + 1 + i < a.T.ns && st < a.T.ns) ? a.T.a[st + 1 + i][st] : 0.f;
    a.exact = p->cfg.mode == CNF_MODE_EXACT;
    a.K = p->KP;
    a.prio_mode = p->prio_mode;
    a.queue = nullptr;
    if (mp->use_queue) {
        if (!mp->queue_dev) {
            hipError_t e = hipMalloc((void**)&mp->queue_dev, sizeof(int));
            if (e != hipSuccess) return e;
        }
        hipError_t e = zero_async(mp->queue_dev, sizeof(int), st);
        if (e != hipSuccess) return e;
        a.queue = mp->queue_dev;
    }
    if (p->kind != 0) { int hmax = 0; for (int l = 1; l < p->cfg.n_layers; ++l) hmax = p->cfg.widths[l] > hmax ? p->cfg.widths[l] : hmax; a.KH = (hmax + 15) / 16; }
    if (p->kind == 2) {
        if (p->cfg.mode == CNF_MODE_HUTCH_JVP) a.exact = 2;   // this kernel family's code for the JVP form (cnf_coop_x.hip)
        a.q_off = p->q_extra;
        if (plan_uses_coopd(p, s.B)) {
            int hmax = 0;
            for (int l = 1; l < p->cfg.n_layers; ++l) hmax = p->cfg.widths[l] > hmax ? p->cfg.widths[l] : hmax;
            const size_t rkf = coopd_rk_floats(hmax, a.D, p->L, p->ACT, a.exact == 1, mp->num_cus);
            if (rkf && !mp->rk_dev) {
                hipError_t e = hipMalloc((void**)&mp->rk_dev, rkf * sizeof(float));
                if (e != hipSuccess) return e;
            }
            a.rk = mp->rk_dev;
            return coopd_launch(hmax, a.D, p->L, p->ACT, p->HT, p->ZR, p->CR, a, mp->num_cus, st);
        }
        return coopx_launch(p->HT, p->L, p->ZR, p->CR, p->ACT, a, mp->num_cus, st);
    }
    if (p->kind == 1 && plan_uses_coopd(p, s.B)) {
        int hmax = 0;
        for (int l = 1; l < p->cfg.n_layers; ++l) hmax = p->cfg.widths[l] > hmax ? p->cfg.widths[l] : hmax;
        const size_t rkf = coopd_rk_floats(hmax, a.D, p->L, p->ACT, 0, mp->num_cus);
        if (rkf && !mp->rk_dev) {
            hipError_t e = hipMalloc((void**)&mp->rk_dev, rkf * sizeof(float));
            if (e != hipSuccess) return e;
        }
        a.rk = mp->rk_dev;
        return coopd_launch(hmax, a.D, p->L, p->ACT, p->HT, p->ZR, 0, a, mp->num_cus, st);
    }
    if (p->kind == 1) {
        // with checkpoint buffers: the checkpointing form of the cooperative solve (the forward half of cnf_coop_grad.hip)
        // (shapes without a checkpointing instance of this kernel checkpoint through the extended kernel, which runs on the same
        // packed image - MfmaLayout(HT, L, ZR, 0, true) - and checkpoints at run time)
        if (s.ckpt) return (coop_ckpt_supported(p->HT, p->L, p->ZR, p->ACT) && !s.tgrid_dev) ? coop_launch_ckpt(p->HT, p->L, p->ZR, p->ACT, a, mp->num_cus, st)
                                                                            : coopx_launch_exact(p->HT, p->L, p->ZR, 0, p->ACT, a, mp->num_cus, st);
        return coop_launch(p->HT, p->L, p->ZR, p->ACT, a, mp->num_cus, st);
    }
    const long long ntiles = (s.B + 15) / 16;
    if (s.nsteps > 0 && !s.ckpt && !s.ckpt_k && !s.kfull && plan_takes_tile_split(mp, s.B))
        return coop_split_launch(p->HT, p->L, p->ZR, p->ACT, a, st);
    // one-probe VJP solves without conditions: the hand-scheduled forms of the same kernel (cnf_mfma2.hip), bit-identical results
    if (p->ENGINE == ENG_VJP && p->KP == 1 && p->CR == 0 && p->arith == CNF_ARITH_F32 && s.nsteps > 0 && !s.kfull && !mp->use_queue) {
        // small batches of two-tile nets: two waves per tile (forward chain / pullback)
        if (tuning().solve2_pair != 0 && solve2p_supported(p->HT, p->L, p->ZR, p->ACT, ntiles, mp->num_cus)) {
            a.use_q = p->pre == 2;   // the plan's own arithmetic for the trace term
            return solve2p_launch(p->HT, p->L, p->ZR, p->ACT, a, st);
        }
        if (tuning().solve2 != 0 && solve2_supported(p->HT, p->L, p->ZR, p->ACT))
            return solve2_launch(p->HT, p->L, p->ZR, p->ACT, tuning().solve2 == 1 ? 256 : 512, a, mp->num_cus, st);
    }
    const int wpb = p->nthreads / 64;
    long long want = ntiles;   // tile t runs on workgroup t % nblocks (cnf_mfma_kernel.h): small batches spread over the CUs
    (void)wpb;
    const int lds = p->lay.lds_total * (int)sizeof(float);
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    const long long cap = (long long)mp->num_cus * per_cu;
    const int nblocks = (int)(want < cap ? want : cap);
    return p->launch(a, lds, nblocks, st);
}


// Largest batch the device-controlled adaptive kernel takes (one tile of 16 samples per resident wave), 0 if the plan has
// no such kernel.
int64_t mfma_adaptive_capacity(MfmaPlan* p) {
    if (!p || p->kind != 0 || !p->launch_adapt) return 0;
    if (tuning().device_controller == 0) return 0;
    if (p->num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        p->num_cus = prop.multiProcessorCount;
    }
    if (p->adapt_per_cu < 0) {
        int occ = 0;
        const hipError_t e = p->launch_adapt(KArgs{}, AArgs{}, p->lay.lds_total * (int)sizeof(float), 0, nullptr, &occ);
        if (e != hipSuccess) { (void)hipGetLastError(); occ = 0; }
        // one workgroup per CU: what the kernel is compiled for (waves_per_eu) and what the tests cover; a second resident
        // workgroup (possible when the register count happens to allow it) only halves the active waves per workgroup
        const int cap_env = tuning().dc_per_cu;
        p->adapt_per_cu = occ < cap_env ? occ : cap_env;
    }
    return (int64_t)p->num_cus * p->adapt_per_cu * (p->nthreads / 64) * 16;
}

// scratch of one device-controlled solve: [2][3][ntiles] doubles, then counter (4 ints) + 8 stats, dts_cap floats, dts_cap orders
size_t mfma_adaptive_scratch_bytes(int64_t B, int dts_cap) {
    const size_t ntiles = (size_t)((B + 15) / 16);
    return 64 + 12 * ntiles * sizeof(double) + (size_t)dts_cap * (sizeof(float) + sizeof(int));   // status words (fixed, at the head), slots, steps + orders
}

static void fill_kargs_adaptive(const MfmaPlan* p, const float* packed_dev, const SolveArgs& s, KArgs& a) {
    a.packed = packed_dev;
    a.u0 = s.u0; a.eps = s.eps; a.ys = s.ys; a.u_out = s.u_out;
    a.ckpt = s.ckpt; a.ckpt_k = s.ckpt_k;   // adaptive Tsit5 only: the accepted steps' checkpoints (AArgs::ckpt_cap of them)
    a.B = s.B; a.nsteps = 1; a.t0 = s.t0; a.dt = 0.f;
    a.nvars = s.nvars; a.D = p->cfg.nvars + p->cfg.naug; a.C = p->cfg.ncond;
    a.reg_z = p->cfg.reg_z; a.reg_j = p->cfg.reg_j; a.reg_aug = s.reg_aug; a.autonomous = p->cfg.autonomous;
    a.exact = p->cfg.mode == CNF_MODE_EXACT;
    a.K = p->KP;
}

// `epoch`: the caller's opaque launch state for this scratch buffer (0 after it was allocated): low half = the launch counter the slots
// and the abort flag are tagged with (AArgs::epoch), so the buffer is zeroed only for the first launch on it and when the 16-bit
// counter wraps; high half = a key of the layout (tiles, step capacity) of the previous launch - the slots' extent moves with the
// batch size, so a launch with another layout could otherwise find stale steps / orders words where it expects tagged slots, one of
// which might carry the current epoch.  The status words (counter, stats, abort flag) sit at a FIXED offset at the head.
static hipError_t fill_aargs_scratch(AArgs& q, void* scratch, long long ntiles, int dts_cap, int** stats_dev, float** dts_dev,
                                     int** orders_dev, unsigned* epoch, hipStream_t st) {
    char* base = (char*)scratch;
    int* ints = (int*)base;                       // 16 status words
    q.counter = (unsigned*)ints;
    q.stats = ints + 4;
    q.slots = (double*)(base + 64);
    // grid_sum3's slots: [2 (round parity)][workgroups <= tiles][6] tagged words
    const size_t slot_bytes = 12 * (size_t)ntiles * sizeof(double);
    q.dts = (float*)(base + 64 + slot_bytes);
    q.orders = (int*)(q.dts + dts_cap);
    *stats_dev = q.stats;
    *dts_dev = q.dts;
    if (orders_dev) *orders_dev = q.orders;
    hipError_t e = hipSuccess;
    const unsigned key = ((((unsigned)ntiles * 2654435761u) ^ ((unsigned)dts_cap * 40503u)) >> 13) & 0xffffu;
    unsigned ep = *epoch & 0xffffu;
    if (ep == 0 || ep >= 0xfffeu || (*epoch >> 16) != key) {
        e = zero_async(base, 64 + slot_bytes, st);
        ep = 0;
    }
    q.epoch = ++ep;
    *epoch = (key << 16) | ep;
    return e;
}

hipError_t mfma_solve_adaptive(MfmaPlan* p, const float* packed_dev, const SolveArgs& s, float abstol, float reltol, float dt_init,
                               int maxiters, void* scratch, unsigned* epoch, int dts_cap, int** stats_dev, float** dts_dev, int* host_rec, int ckpt_cap,
                               hipStream_t st) {
    const long long ntiles = (s.B + 15) / 16;
    KArgs a{};
    fill_kargs_adaptive(p, packed_dev, s, a);
    AArgs q{};
    q.host_rec = host_rec;
    q.ckpt_cap = (s.ckpt && s.ckpt_k) ? ckpt_cap : 0;
    if (!q.ckpt_cap) a.ckpt = a.ckpt_k = nullptr;
    q.abstol = abstol; q.reltol = reltol; q.t1 = s.t1; q.dt_init = dt_init; q.maxiters = maxiters; q.dts_cap = dts_cap;
    const Tableau T = make_tableau(CNF_ALG_TSIT5);
    // b - bhat of the embedded 4th-order solution (Tsitouras 2011); the same constants as cnf_step_embedded
    static const float btilde[7] = {-0.00178001105222577714f, -0.0008164344596567469f, 0.007880878010261995f,
                                    -0.1447110071732629f, 0.5823571654525552f, -0.45808210592918697f,
                                    0.015151515151515152f};
    for (int i = 0; i < 7; ++i) { q.bt[i] = btilde[i]; q.c[i] = i < 6 ? T.c[i] : 1.f; }
    for (int j = 0; j < 6; ++j) {
        q.b[j] = T.b[j];
        for (int i = 0; i < 6; ++i) {
            const int row = j + 1 + i;   // the stage that receives stage j's derivative
            q.acol[j][i] = row < 6 ? T.a[row][j] : (row == 6 ? T.b[j] : 0.f);
        }
    }
    hipError_t e = fill_aargs_scratch(q, scratch, ntiles, dts_cap, stats_dev, dts_dev, nullptr, epoch, st);
    if (e != hipSuccess) return e;
    const int wpb = p->nthreads / 64;
    const int lds = p->lay.lds_total * (int)sizeof(float);
    const long long cap = (long long)p->num_cus * (p->adapt_per_cu > 0 ? p->adapt_per_cu : 0);
    // tile t runs on workgroup t % nblocks, wave t / nblocks: spread over the CUs first
    long long nblocks = ntiles < cap ? ntiles : cap;
    if (nblocks * wpb < ntiles) return hipErrorInvalidValue;
    return p->launch_adapt(a, q, lds, (int)nblocks, st, nullptr);
}

// Largest batch the device-resident VCABM kernel takes (256-thread workgroups, one tile per wave), 0 if the plan has none.
int64_t mfma_vcabm_capacity(MfmaPlan* p) {
    if (!p || p->kind != 0 || !p->launch_vcabm) return 0;
    if (tuning().device_controller == 0) return 0;
    if (p->num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        p->num_cus = prop.multiProcessorCount;
    }
    if (p->vcabm_per_cu < 0) {
        int occ = 0;
        const hipError_t e = p->launch_vcabm(KArgs{}, AArgs{}, p->lay.lds_total * (int)sizeof(float), 0, nullptr, &occ);
        if (e != hipSuccess) { (void)hipGetLastError(); occ = 0; }
        const int cap_env = tuning().dc_per_cu;
        p->vcabm_per_cu = occ < cap_env ? occ : cap_env;
    }
    return (int64_t)p->num_cus * p->vcabm_per_cu * 4 * 16;
}

hipError_t mfma_solve_vcabm(MfmaPlan* p, const float* packed_dev, const SolveArgs& s, float abstol, float reltol, float dt_init,
                            int maxiters, void* scratch, unsigned* epoch, int dts_cap, int** stats_dev, float** dts_dev, int** orders_dev, int* host_rec, hipStream_t st) {
    const long long ntiles = (s.B + 15) / 16;
    KArgs a{};
    fill_kargs_adaptive(p, packed_dev, s, a);
    a.ckpt = a.ckpt_k = nullptr;
    AArgs q{};
    q.host_rec = host_rec;
    q.abstol = abstol; q.reltol = reltol; q.t1 = s.t1; q.dt_init = dt_init; q.maxiters = maxiters; q.dts_cap = dts_cap;
    hipError_t e = fill_aargs_scratch(q, scratch, ntiles, dts_cap, stats_dev, dts_dev, orders_dev, epoch, st);
    if (e != hipSuccess) return e;
    const int lds = p->lay.lds_total * (int)sizeof(float);
    // one workgroup per CU as long as the batch fits that way (fewer workgroups to sum over: 0.233 against 0.267 ms per solve at
    // 16 384 samples); the second one the small-net instances allow only beyond (<= 32 768 samples: 0.27 ms against the host
    // loop's 0.51, profiles/dc_per_cu_ab.py)
    const long long cap1 = (long long)p->num_cus * (p->vcabm_per_cu > 0 ? 1 : 0);
    const long long cap = ntiles <= cap1 * 4 ? cap1 : (long long)p->num_cus * (p->vcabm_per_cu > 0 ? p->vcabm_per_cu : 0);
    long long nblocks = ntiles < cap ? ntiles : cap;
    if (nblocks * 4 < ntiles) return hipErrorInvalidValue;
    return p->launch_vcabm(a, q, lds, (int)nblocks, st, nullptr);
}

}  // namespace cnf
