/* cnf_oracle.c — CPU fp32 restatement of the reference hot path.  See cnf_oracle.h.
 * TEST INFRASTRUCTURE ONLY; parity unpinned by the reference (no Julia, no golden vectors).
 *
 * One dynamics call (cnf_oracle_aug_f) has the reference's structure: per-layer products over the batch columns, then a
 * hand-written pullback / pushforward of the Dense chain (src/core/icnf.jl:517-559, src/core/utils.jl:150-170), with the one
 * departure that the forward pass is evaluated once per call, not twice (src/core/utils.jl:157-158; same numbers, favours the
 * CPU).  The fixed-step solve (cnf_oracle_integrate_fixed) calls it 4 (RK4) or 6 (Tsit5) times per step with the RK axpys in
 * between - the shape of SciMLBase.solve driving make_ode_func's closure (src/core/base_icnf.jl:62-78, 134-140) - but runs that
 * loop per chunk of 256 columns inside each thread (columns are independent under fixed steps), which is what a CPU needs to
 * keep the stage data in cache; the numbers are the same as a whole-batch loop's, bit for bit.
 */
#include "cnf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CB 64 /* columns (samples) per work block */

enum { ACT_ID = 0, ACT_TANH = 1, ACT_SOFTPLUS = 2 };
enum { MODE_VJP = 0, MODE_JVP = 1, MODE_EXACT = 2 };

int cnf_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int check_cfg(const cnf_oracle_cfg* c) {
    if (!c || c->n_layers < 1 || c->n_layers > CNF_ORACLE_MAX_LAYERS) return -1;
    int D = c->nvars + c->naug;
    int n_in = D + (c->autonomous ? 0 : 1) + c->ncond; /* src/core/icnf.jl:64 */
    if (c->widths[0] != n_in || c->widths[c->n_layers] != D) return -2;
    if (c->mode < 0 || c->mode > 2 || c->nprobes < 1) return -3;
    return 0;
}

/* Lux swaps tanh -> NNlib.tanh_fast on CPU Float32 arrays (upstream): a rational
 * approximation, max abs error 3.3e-7 (checked against tanh in float64).  The parity checks use
 * libm tanhf (default); bench.py's cpu_baseline switches this on so the timed CPU path does the
 * arithmetic the reference's CPU path does (and vectorises). */
static int g_fast_tanh = 0;
void cnf_oracle_set_fast_tanh(int on) { g_fast_tanh = on; }
static inline float tanh_fast_f(float x) {
    const float x2 = x * x;
    const float n = 1.0f + x2 * (0.1346604f + x2 * (0.0035974074f + x2 * (2.2332108e-5f + x2 * 1.587199e-8f)));
    const float d = 1.0f + x2 * (0.4679937f + x2 * (0.026262015f + x2 * (0.0003453992f + x2 * 8.7767893e-7f)));
    const float r = x * (n / d);
    return x2 < 66.f ? r : (x > 0.f ? 1.f : -1.f);
}

/* NNlib.softplus(x) = log1p(exp(-|x|)) + relu(x) */
static inline float softplusf(float x) { return log1pf(expf(-fabsf(x))) + (x > 0.f ? x : 0.f); }
static inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

/* per-thread scratch: pre-activations a_l, activations h_l (h_0 = input), and two
 * delta/tangent ping-pong buffers, all laid out [feature][column-in-block]. */
typedef struct {
    float* a[CNF_ORACLE_MAX_LAYERS + 1];
    float* h[CNF_ORACLE_MAX_LAYERS + 1];
    float* d0;
    float* d1;
    float* base;
} scratch;

static int scratch_init(scratch* s, const cnf_oracle_cfg* c) {
    size_t tot = 0, wmax = 0;
    for (int l = 0; l <= c->n_layers; ++l) {
        tot += 2 * (size_t)c->widths[l];
        if ((size_t)c->widths[l] > wmax) wmax = (size_t)c->widths[l];
    }
    tot += 2 * wmax;
    s->base = (float*)aligned_alloc(64, ((tot * CB * sizeof(float) + 63) / 64) * 64);
    if (!s->base) return -1;
    float* q = s->base;
    for (int l = 0; l <= c->n_layers; ++l) {
        s->a[l] = q; q += (size_t)c->widths[l] * CB;
        s->h[l] = q; q += (size_t)c->widths[l] * CB;
    }
    s->d0 = q; q += wmax * CB;
    s->d1 = q;
    return 0;
}

/* ---- the per-layer products of one column block, register-tiled -------------------------------------------------
 * out (M x CB) = A (M x K) in (K x CB) [+ bias], A(m, k) at A[m * sm + k * sk] (so W and W^T are the same routine).
 * Four output rows x 32 columns live in registers (8 AVX-512 or 16 AVX2 accumulators); per k: two loads of `in`, four
 * broadcast weights, eight FMAs.  Compiled for AVX-512 and for the baseline ISA, selected at load time from cpuid
 * (target_clones), so the timed CPU baseline uses the widest vectors the host has - VERDICT r1: the round-1 loops updated the
 * block's accumulators through memory for every input row and reached about 1 % of the cores' FMA peak. */
typedef float v16f __attribute__((vector_size(64), aligned(4)));
#define CNF_ISA_CLONES __attribute__((target_clones("avx512f", "default")))

CNF_ISA_CLONES
static void gemm_block(const float* A, long sm, long sk, int M, int K, const float* in, float* out, const float* bias) {
    for (int m0 = 0; m0 < M; m0 += 4) {
        const int mb = M - m0 < 4 ? M - m0 : 4;
        for (int c0 = 0; c0 < CB; c0 += 32) {
            v16f acc[4][2];
            for (int r = 0; r < 4; ++r) {
                const float b0 = (bias && r < mb) ? bias[m0 + r] : 0.f;
                for (int e = 0; e < 16; ++e) { acc[r][0][e] = b0; acc[r][1][e] = b0; }
            }
            const float* a0 = A + (long)m0 * sm;
            if (mb == 4) {
                for (int k = 0; k < K; ++k) {
                    const v16f x0 = *(const v16f*)(in + (size_t)k * CB + c0), x1 = *(const v16f*)(in + (size_t)k * CB + c0 + 16);
                    const float* ak = a0 + (long)k * sk;
                    const float w0 = ak[0], w1 = ak[sm], w2 = ak[2 * sm], w3 = ak[3 * sm];
                    acc[0][0] += w0 * x0; acc[0][1] += w0 * x1;
                    acc[1][0] += w1 * x0; acc[1][1] += w1 * x1;
                    acc[2][0] += w2 * x0; acc[2][1] += w2 * x1;
                    acc[3][0] += w3 * x0; acc[3][1] += w3 * x1;
                }
            } else {
                for (int k = 0; k < K; ++k) {
                    const v16f x0 = *(const v16f*)(in + (size_t)k * CB + c0), x1 = *(const v16f*)(in + (size_t)k * CB + c0 + 16);
                    for (int r = 0; r < mb; ++r) {
                        const float w = a0[(long)r * sm + (long)k * sk];
                        acc[r][0] += w * x0; acc[r][1] += w * x1;
                    }
                }
            }
            for (int r = 0; r < mb; ++r) {
                *(v16f*)(out + (size_t)(m0 + r) * CB + c0) = acc[r][0];
                *(v16f*)(out + (size_t)(m0 + r) * CB + c0 + 16) = acc[r][1];
            }
        }
    }
}

CNF_ISA_CLONES
static void act_fwd_block(int act, int fast_tanh, size_t n, const float* a, float* h) {
    if (act == ACT_ID) {
        memcpy(h, a, n * sizeof(float));
    } else if (act == ACT_TANH) {
        if (fast_tanh) {
#pragma omp simd
            for (size_t k = 0; k < n; ++k) h[k] = tanh_fast_f(a[k]);
        } else {
            for (size_t k = 0; k < n; ++k) h[k] = tanhf(a[k]);
        }
    } else {
        for (size_t k = 0; k < n; ++k) h[k] = softplusf(a[k]);
    }
}

/* Dense forward over one column block: a = W h_prev + b, h = act(a).
 * Lux.Dense with weight (out x in) column-major: W(o,i) at o + out*i. */
static void dense_fwd(const float* W, const float* b, int fin, int fout, int act,
                      const float* hp, float* a, float* h) {
    gemm_block(W, 1, fout, fout, fin, hp, a, b);
    act_fwd_block(act, g_fast_tanh, (size_t)fout * CB, a, h);
}

/* multiply d (fout x CB) by act'(a) in place: tanh' = 1-h^2, softplus' = sigmoid(a) */
CNF_ISA_CLONES
static void act_grad_mul(int act, int f, const float* a, const float* h, float* d) {
    const size_t n = (size_t)f * CB;
    if (act == ACT_TANH) {
#pragma omp simd
        for (size_t k = 0; k < n; ++k) d[k] *= (1.f - h[k] * h[k]);
    } else if (act == ACT_SOFTPLUS) {
        for (size_t k = 0; k < n; ++k) d[k] *= sigmoidf(a[k]);
    }
}

/* g_prev = W^T d  (fin x CB):  A(m = i, k = o) = W(o, i) */
static void dense_bwd(const float* W, int fin, int fout, const float* d, float* gp) {
    gemm_block(W, fout, 1, fin, fout, d, gp, NULL);
}

/* tau = W tau_prev  (fout x CB), no bias */
static void dense_tan(const float* W, int fin, int fout, const float* tp, float* tq) {
    gemm_block(W, 1, fout, fout, fin, tp, tq, NULL);
}

/* which clone the loader picked: 2 = AVX-512, 1 = the baseline ISA of the build (x86-64-v3: AVX2 + FMA) */
int cnf_oracle_isa(void) {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") ? 2 : 1;
}

/* pullback through the whole chain: seed (D x CB) in d0 -> returns pointer to the
 * (n_in x CB) input cotangent; only rows 0..D-1 are the z-cotangent (t and ys enter through
 * CondLayer closures and are not differentiated inputs, src/core/icnf.jl:147-153). */
static float* chain_vjp(const cnf_oracle_cfg* c, const float* p, const size_t* w_off, scratch* s) {
    float* d = s->d0;
    float* g = s->d1;
    for (int l = c->n_layers; l >= 1; --l) {
        act_grad_mul(c->acts[l - 1], c->widths[l], s->a[l], s->h[l], d);
        dense_bwd(p + w_off[l - 1], c->widths[l - 1], c->widths[l], d, g);
        float* tmp = d; d = g; g = tmp;
    }
    return d;
}

/* pushforward: tangent of the input in d0 (n_in x CB) -> output tangent (D x CB) */
static float* chain_jvp(const cnf_oracle_cfg* c, const float* p, const size_t* w_off, scratch* s) {
    float* tp = s->d0;
    float* tq = s->d1;
    for (int l = 1; l <= c->n_layers; ++l) {
        dense_tan(p + w_off[l - 1], c->widths[l - 1], c->widths[l], tp, tq);
        act_grad_mul(c->acts[l - 1], c->widths[l], s->a[l], s->h[l], tq);
        float* tmp = tp; tp = tq; tq = tmp;
    }
    return tp;
}

static void aug_f_block(const cnf_oracle_cfg* c, const float* p, const size_t* w_off,
                        const size_t* b_off, const float* u, float t, const float* eps,
                        const float* ys, int64_t col0, int ncols, float* du, scratch* s) {
    const int D = c->nvars + c->naug, S = D + 3, C = c->ncond, K = c->nprobes;
    const int n_in = c->widths[0];
    /* h0 = [z; t; ys]  (src/layers/cond_layer.jl:7-31; wrapping order CondLayer(CondLayer(nn,ys),t)) */
    float* h0 = s->h[0];
    memset(h0, 0, (size_t)n_in * CB * sizeof(float));
    for (int cc = 0; cc < ncols; ++cc) {
        const float* uc = u + (size_t)(col0 + cc) * S;
        for (int i = 0; i < D; ++i) h0[(size_t)i * CB + cc] = uc[i];
        int r = D;
        if (!c->autonomous) h0[(size_t)(r++) * CB + cc] = t;
        for (int i = 0; i < C; ++i) h0[(size_t)(r + i) * CB + cc] = ys[(size_t)(col0 + cc) * C + i];
    }
    for (int l = 1; l <= c->n_layers; ++l)
        dense_fwd(p + w_off[l - 1], p + b_off[l - 1], c->widths[l - 1], c->widths[l],
                  c->acts[l - 1], s->h[l - 1], s->a[l], s->h[l]);
    const float* zd = s->h[c->n_layers];

    float ldot[CB], ndot[CB], edot[CB];
    for (int cc = 0; cc < CB; ++cc) ldot[cc] = ndot[cc] = edot[cc] = 0.f;

    if (c->mode == MODE_EXACT) {
        /* tr J = sum_i (e_i^T J)_i : D pullbacks with one-hot seeds, the DI variant of
         * src/core/utils.jl:35-56; the Lux variant (utils.jl:79-88) gives the same J. */
        for (int i = 0; i < D; ++i) {
            memset(s->d0, 0, (size_t)D * CB * sizeof(float));
            for (int cc = 0; cc < CB; ++cc) s->d0[(size_t)i * CB + cc] = 1.f;
            const float* g = chain_vjp(c, p, w_off, s);
            for (int cc = 0; cc < CB; ++cc) ldot[cc] -= g[(size_t)i * CB + cc];
        }
    } else {
        const float invK = 1.f / (float)K;
        for (int k = 0; k < K; ++k) {
            const float* g;
            if (c->mode == MODE_VJP) {
                for (int cc = 0; cc < CB; ++cc)
                    for (int i = 0; i < D; ++i)
                        s->d0[(size_t)i * CB + cc] =
                            cc < ncols ? eps[(size_t)(col0 + cc) * (K * D) + (size_t)k * D + i] : 0.f;
                g = chain_vjp(c, p, w_off, s);
            } else {
                memset(s->d0, 0, (size_t)n_in * CB * sizeof(float));
                for (int cc = 0; cc < ncols; ++cc)
                    for (int i = 0; i < D; ++i)
                        s->d0[(size_t)i * CB + cc] =
                            eps[(size_t)(col0 + cc) * (K * D) + (size_t)k * D + i];
                g = chain_jvp(c, p, w_off, s);
            }
            for (int cc = 0; cc < ncols; ++cc) {
                const float* e = eps + (size_t)(col0 + cc) * (K * D) + (size_t)k * D;
                float dot = 0.f, nn = 0.f;
                for (int i = 0; i < D; ++i) {
                    const float gi = g[(size_t)i * CB + cc];
                    dot += gi * e[i];   /* ldot = -sum(eJ .* e)   src/core/icnf.jl:532 */
                    nn += gi * gi;      /* ndot = norm(eJ)        src/core/icnf.jl:229-245 */
                }
                ldot[cc] -= invK * dot;
                if (c->reg_j) ndot[cc] += invK * sqrtf(nn);
            }
        }
        if (c->reg_z)
            for (int cc = 0; cc < ncols; ++cc) {
                float nn = 0.f;
                for (int i = 0; i < D; ++i) { const float v = zd[(size_t)i * CB + cc]; nn += v * v; }
                edot[cc] = sqrtf(nn);   /* Edot = norm(zdot)      src/core/icnf.jl:184-199 */
            }
    }
    for (int cc = 0; cc < ncols; ++cc) {
        float* dc = du + (size_t)(col0 + cc) * S;   /* vcat(zdot, ldot, Edot, ndot)  icnf.jl:535 */
        for (int i = 0; i < D; ++i) dc[i] = zd[(size_t)i * CB + cc];
        dc[D] = ldot[cc];
        dc[D + 1] = edot[cc];
        dc[D + 2] = ndot[cc];
    }
}

/* worksharing body: must be called by every thread of an enclosing parallel region */
static void aug_f_ws(const cnf_oracle_cfg* cfg, const float* p, const size_t* w_off,
                     const size_t* b_off, const float* u, float t, const float* eps,
                     const float* ys, int64_t B, float* du, scratch* s) {
    const int64_t nblk = (B + CB - 1) / CB;
#pragma omp for schedule(static)
    for (int64_t blk = 0; blk < nblk; ++blk) {
        const int64_t col0 = blk * CB;
        const int ncols = (int)((B - col0) < CB ? (B - col0) : CB);
        aug_f_block(cfg, p, w_off, b_off, u, t, eps, ys, col0, ncols, du, s);
    }
}

int cnf_oracle_aug_f(const cnf_oracle_cfg* cfg, const float* p, const size_t* w_off,
                     const size_t* b_off, const float* u, float t, const float* eps,
                     const float* ys, int64_t B, float* du, int nthreads) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (cfg->mode != MODE_EXACT && !eps) return -4;
    if (cfg->ncond && !ys) return -5;
    int err = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        scratch s;
        const int bad = scratch_init(&s, cfg);
        if (bad) {
#pragma omp atomic write
            err = -6;
        }
#pragma omp barrier
        if (!err) aug_f_ws(cfg, p, w_off, b_off, u, t, eps, ys, B, du, &s);
        if (!bad) free(s.base);
    }
    return err;
}

/* ---- fixed-step explicit Runge-Kutta (coefficients rounded to float as OrdinaryDiffEq does
 * for T = Float32; tableau values: SURVEY.md §8 A4, Tsitouras 2011) ------------------------ */
static const float RK4_C[4] = {0.f, 0.5f, 0.5f, 1.f};
static const float RK4_A[4][4] = {{0}, {0.5f}, {0.f, 0.5f}, {0.f, 0.f, 1.f}};
static const float RK4_B[4] = {1.f / 6.f, 1.f / 3.f, 1.f / 3.f, 1.f / 6.f};
static const float T5_C[6] = {0.f, 0.161f, 0.327f, 0.9f, 0.9800255409045097f, 1.f};
static const float T5_A[6][6] = {
    {0},
    {0.161f},
    {-0.008480655492356989f, 0.335480655492357f},
    {2.8971530571054935f, -6.359448489975075f, 4.3622954328695815f},
    {5.325864828439257f, -11.748883564062828f, 7.4955393428898365f, -0.09249506636175525f},
    {5.86145544294642f, -12.92096931784711f, 8.159367898576159f, -0.071584973281401f,
     -0.028269050394068383f}};
static const float T5_B[6] = {0.09646076681806523f, 0.01f,  0.4798896504144996f,
                              1.379008574103742f,   -3.290069515436081f, 2.324710524099774f};

int cnf_oracle_integrate_fixed(const cnf_oracle_cfg* cfg, const float* p, const size_t* w_off,
                               const size_t* b_off, int alg, int nsteps, float t0, float t1,
                               const float* u0, const float* eps, const float* ys, int64_t B,
                               float* u1, int nthreads) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (nsteps < 1 || (alg != 0 && alg != 1)) return -7;
    const int ns = alg == 0 ? 4 : 6;
    const float* Cc = alg == 0 ? RK4_C : T5_C;
    const float* Bc = alg == 0 ? RK4_B : T5_B;
    const int S = cfg->nvars + cfg->naug + 3;
    const int C = cfg->ncond, KD = cfg->nprobes * (cfg->nvars + cfg->naug);
    const size_t n = (size_t)S * (size_t)B;
    if (u1 != u0) memcpy(u1, u0, n * sizeof(float));
    const float dt = (t1 - t0) / (float)nsteps;
    if (nthreads < 1) nthreads = 1;
    if (cfg->mode != MODE_EXACT && !eps) rc = -4;
    if (cfg->ncond && !ys) rc = -5;
    int err = rc;
    /* Columns are independent under fixed-step integration (every operation of augmented_f is column-wise,
     * src/core/icnf.jl:530-535), so each thread takes chunks of CHUNK columns through ALL steps with chunk-local stage
     * buffers: per stage a dynamics call over the chunk (the reference's per-call structure, block by block) and the RK
     * axpys on data that stays in the core's L2.  No barriers, no shared whole-state arrays (round 1 kept the whole state
     * and six stage arrays shared and synchronised every stage: 2 GFLOP/s per core on a 128-core, two-socket host whose
     * block products alone run at ~30). */
    enum { CHUNK = 4 * CB };
    const int64_t nchunks = (B + CHUNK - 1) / CHUNK;
#pragma omp parallel num_threads(nthreads)
    {
        scratch s;
        const int bad = scratch_init(&s, cfg);
        float* loc = (float*)aligned_alloc(64, (size_t)(8 * S * CHUNK) * sizeof(float));
        if (bad || !loc) {
#pragma omp atomic write
            err = -6;
        }
#pragma omp barrier
        if (!err) {
            float* ul = loc;                       /* the chunk's state, S x CHUNK (column-major like the ABI) */
            float* us = loc + (size_t)S * CHUNK;   /* stage state */
            float* kl[6];
            for (int i = 0; i < 6; ++i) kl[i] = loc + (size_t)(2 + i) * S * CHUNK;
#pragma omp for schedule(dynamic, 1)
            for (int64_t ch = 0; ch < nchunks; ++ch) {
                const int64_t c0 = ch * CHUNK;
                const int nc = (int)((B - c0) < CHUNK ? (B - c0) : CHUNK);
                const size_t nl = (size_t)S * (size_t)nc;
                memcpy(ul, u1 + (size_t)c0 * S, nl * sizeof(float));
                const float* epsl = eps ? eps + (size_t)c0 * KD : NULL;
                const float* ysl = ys ? ys + (size_t)c0 * C : NULL;
                for (int step = 0; step < nsteps; ++step) {
                    const float tn = t0 + (float)step * dt;
                    for (int i = 0; i < ns; ++i) {
                        const float* Ai = alg == 0 ? RK4_A[i] : T5_A[i];
                        for (size_t e = 0; e < nl; ++e) {
                            float acc = 0.f;
                            for (int j = 0; j < i; ++j) acc += Ai[j] * kl[j][e];
                            us[e] = ul[e] + dt * acc;
                        }
                        for (int b0 = 0; b0 < nc; b0 += CB)
                            aug_f_block(cfg, p, w_off, b_off, us, tn + Cc[i] * dt, epsl, ysl, b0, nc - b0 < CB ? nc - b0 : CB, kl[i], &s);
                    }
                    for (size_t e = 0; e < nl; ++e) {
                        float acc = 0.f;
                        for (int i = 0; i < ns; ++i) acc += Bc[i] * kl[i][e];
                        ul[e] += dt * acc;
                    }
                }
                memcpy(u1 + (size_t)c0 * S, ul, nl * sizeof(float));
            }
        }
        if (!bad) free(s.base);
        free(loc);
    }
    rc = err;
    return rc;
}

int cnf_oracle_inference_fixed(const cnf_oracle_cfg* cfg, const float* p, const size_t* w_off,
                               const size_t* b_off, int alg, int nsteps, float t0, float t1,
                               const float* x, const float* eps, const float* ys, int64_t B,
                               float* logp, float* regs, float* u_final, int nthreads) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    const int nv = cfg->nvars, D = nv + cfg->naug, S = D + 3;
    float* u = (float*)calloc((size_t)S * (size_t)B, sizeof(float));
    if (!u) return -6;
    /* u0 = vcat(xs, zeros(naug + n_aug + 1, B))   src/core/base_icnf.jl:256-266 */
    for (int64_t c = 0; c < B; ++c) memcpy(u + (size_t)c * S, x + (size_t)c * nv, (size_t)nv * sizeof(float));
    rc = cnf_oracle_integrate_fixed(cfg, p, w_off, b_off, alg, nsteps, t0, t1, u, eps, ys, B, u, nthreads);
    if (!rc) {
        const float log2pi = 1.8378770664093453f;
        for (int64_t c = 0; c < B; ++c) {
            const float* uc = u + (size_t)c * S;
            float ss = 0.f, sa = 0.f;
            for (int i = 0; i < D; ++i) ss += uc[i] * uc[i];
            for (int i = nv; i < D; ++i) sa += uc[i] * uc[i];
            /* logp = logpdf(basedist, z) - dlogp   src/core/base_icnf.jl:165-169 */
            logp[c] = (-0.5f * (float)D * log2pi - 0.5f * ss) - uc[D];
            if (regs) {
                regs[c] = uc[D + 1];
                regs[(size_t)B + c] = uc[D + 2];
                regs[2 * (size_t)B + c] = (cfg->reg_aug && cfg->naug > 0) ? sqrtf(sa) : 0.f;
            }
        }
        if (u_final) memcpy(u_final, u, (size_t)S * (size_t)B * sizeof(float));
    }
    free(u);
    return rc;
}
