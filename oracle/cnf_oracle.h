/* cnf_oracle.h — CPU fp32 restatement of the reference's batched augmented-ODE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libcnf_hip.so) never links or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference (impICNF/ContinuousNormalizingFlows.jl
 * v0.31.0) is pure Julia, cannot run in the build container or on the GPU box, and its tests
 * hold no golden vectors for this path.  This restatement is pinned instead against the
 * independent fp64 autograd oracle (oracle/cnf_oracle64.py), the committed fixtures under
 * tests/golden/ and the analytic known-answer tests in tests/test_oracle_kat.py.
 *
 * All matrices use the Julia layout of the reference: column-major, one column per sample
 * (a sample's rows are contiguous; the stride between samples is the row count).
 * Citations are relative to /root/reference.
 */
#ifndef CNF_ORACLE_H
#define CNF_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CNF_ORACLE_MAX_LAYERS 8

typedef struct {
    int32_t nvars, naug, ncond, autonomous;
    int32_t n_layers;                              /* number of Dense layers */
    int32_t widths[CNF_ORACLE_MAX_LAYERS + 1];     /* widths[0]=n_in ... widths[n_layers]=D */
    int32_t acts[CNF_ORACLE_MAX_LAYERS];           /* 0 identity, 1 tanh, 2 softplus */
    int32_t mode;                                  /* 0 hutch vjp, 1 hutch jvp, 2 exact */
    int32_t nprobes;                               /* K (reference: 1) */
    int32_t reg_z, reg_j, reg_aug;                 /* Edot, ndot, Adot switched on */
} cnf_oracle_cfg;

/* augmented_f, MatrixMode: src/core/icnf.jl:517-559 (Hutchinson VJP), :561-603 (JVP),
 * :297-339 (exact).  u, du: S x B with S = D+3.  eps: (K*D) x B.  ys: C x B or NULL. */
int cnf_oracle_aug_f(const cnf_oracle_cfg* cfg, const float* p, const size_t* w_off,
                     const size_t* b_off, const float* u, float t, const float* eps,
                     const float* ys, int64_t B, float* du, int nthreads);

/* base_sol with a fixed-step RK method: src/core/base_icnf.jl:134-140.
 * alg: 0 RK4, 1 Tsit5.  u0, u1: S x B. */
int cnf_oracle_integrate_fixed(const cnf_oracle_cfg* cfg, const float* p, const size_t* w_off,
                               const size_t* b_off, int alg, int nsteps, float t0, float t1,
                               const float* u0, const float* eps, const float* ys, int64_t B,
                               float* u1, int nthreads);

/* inference_prob + inference_sol, MatrixMode: src/core/base_icnf.jl:247-296, 158-172.
 * x: nvars x B.  logp: B.  regs: 3*B (Edot | ndot | Adot, each B long) or NULL.
 * u_final: S x B or NULL. */
int cnf_oracle_inference_fixed(const cnf_oracle_cfg* cfg, const float* p, const size_t* w_off,
                               const size_t* b_off, int alg, int nsteps, float t0, float t1,
                               const float* x, const float* eps, const float* ys, int64_t B,
                               float* logp, float* regs, float* u_final, int nthreads);

int cnf_oracle_max_threads(void);

/* 1: use NNlib.tanh_fast's rational approximation (what Lux runs on CPU Float32); 0: libm tanhf */
void cnf_oracle_set_fast_tanh(int on);

/* which ISA clone of the block products the loader selected from cpuid: 2 = AVX-512, 1 = the build's baseline (AVX2 + FMA) */
int cnf_oracle_isa(void);

#ifdef __cplusplus
}
#endif
#endif
