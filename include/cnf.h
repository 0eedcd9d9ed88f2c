/* cnf.h — C ABI of libcnf_hip.so: the MI355X (gfx950) implementation of the batched
 * augmented-ODE hot path of impICNF/ContinuousNormalizingFlows.jl v0.31.0.
 *
 * The reference has no FFI: its extension point is Julia dispatch on a ComputeMode subtype
 * (src/core/types.jl:9-35).  A `HIPMatrixMode <: MatrixMode` binds the entry
 * points below with `ccall`; each one names the reference method it replaces.  Citations are
 * relative to the reference repository root.
 *
 * Conventions
 *   - plain C: pointers, sizes, scalars; no C++/torch types; nothing throws across the ABI.
 *   - every function returns 0 on success or a negative cnf_status; cnf_last_error() gives the
 *     message of the last failure on the calling thread.
 *   - all matrices are Float32 in the reference's Julia layout: column-major, one column per
 *     sample (a sample's rows contiguous, stride between samples = row count).
 *   - D = nvars + naug, S = D + 3 state rows [z (D); dlogp; E; n]  (src/core/icnf.jl:143-145,
 *     535; src/core/base_icnf.jl:165-167).
 *   - u, du, x, eps, ys, logp, regs, u_final, sums4 are DEVICE pointers (hipMalloc or a torch /
 *     ROCArray allocation on the handle's device).  p in cnf_set_params may be host or device.
 *   - work is enqueued on the caller's HIP stream (`stream` is a hipStream_t passed as void*;
 *     NULL = the null stream) and is asynchronous with respect to the host.
 *   - the caller owns every array; the library keeps no caller pointer after a call returns
 *     (cnf_set_params copies and repacks the weights).  A handle is bound to one device and is
 *     not thread-safe; distinct handles are independent.
 *   - NaN/Inf in outputs is not an error (the reference ignores the solver retcode,
 *     src/core/base_icnf.jl:138-139).
 *
 * Entry points
 *   lifetime / parameters    cnf_create, cnf_destroy, cnf_set_params
 *   boundary A (per call)    cnf_aug_f                      du = augmented_f(u, p, t)
 *   boundary B (whole solve) cnf_integrate_fixed, cnf_inference_fixed, cnf_integrate_fixed_dt, cnf_inference_fixed_dt, cnf_loss_sums, cnf_loss_mean
 *   caller-driven solves     cnf_assemble_u0, cnf_step_embedded (adaptive Tsit5 attempt), cnf_epilogue,
 *                            cnf_vcabm_begin / _attempt / _accept / _state, cnf_solve_vcabm (the reference's default alg VCABM), cnf_solve_tsit5,
 *                            cnf_loss_adaptive (the whole `loss` call under either)
 *   training                 cnf_loss_grad_fixed, cnf_loss_grad_grid, cnf_loss_grad_adaptive  (dloss/dps, optionally dloss/dxs)
 *   column shards (RCCL)     cnf_comm_unique_id, cnf_comm_init, cnf_comm_init_all, cnf_comm_destroy, cnf_comm_rank, cnf_comm_size,
 *                            cnf_allreduce_loss (the mean in `loss`), cnf_allreduce_sum, cnf_comm_group_start / _end
 *   tuning (A/B, tests)      cnf_get_tuning, cnf_set_tuning
 *   introspection            cnf_version, cnf_build_info, cnf_last_error, cnf_kernel_path, cnf_kernel_family, cnf_grad_path, cnf_grad_path_for, cnf_grad_form_for, cnf_repack_on_device, cnf_solve_controller
 */
#ifndef CNF_H
#define CNF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNF_ABI_VERSION 1
#define CNF_MAX_LAYERS 8

typedef enum {
    CNF_OK = 0,
    CNF_ERR_INVALID = -1,      /* bad argument / inconsistent config (Julia: MethodError/DimensionMismatch) */
    CNF_ERR_UNSUPPORTED = -2,  /* configuration outside what the kernels implement */
    CNF_ERR_NO_PARAMS = -3,    /* cnf_set_params has not been called */
    CNF_ERR_HIP = -4,          /* HIP runtime error (message in cnf_last_error) */
    CNF_ERR_NO_DEVICE = -5,    /* no gfx950 device visible */
    CNF_ERR_COMM = -6          /* RCCL error, or librccl.so.1 could not be loaded (message in cnf_last_error) */
} cnf_status;

/* activation ids of Lux.Dense layers (src/core/icnf.jl:67-71) */
enum { CNF_ACT_IDENTITY = 0, CNF_ACT_TANH = 1, CNF_ACT_SOFTPLUS = 2 };

/* trace estimator = ComputeMode x Mode of the reference:
 *   HUTCH_VJP  LuxVecJacMatrixMode / DIVecJacMatrixMode + TrainMode   (src/core/utils.jl:150-159)
 *   HUTCH_JVP  LuxJacVecMatrixMode / DIJacVecMatrixMode + TrainMode   (src/core/utils.jl:161-170)
 *   EXACT      any MatrixMode + TestMode                              (src/core/utils.jl:79-88) */
enum { CNF_MODE_HUTCH_VJP = 0, CNF_MODE_HUTCH_JVP = 1, CNF_MODE_EXACT = 2 };

/* fixed-step integrators a user selects through sol_kwargs=(alg, adaptive=false, dt)
 * (src/core/base_icnf.jl:138) */
enum { CNF_ALG_RK4 = 0, CNF_ALG_TSIT5 = 1, CNF_ALG_VCABM = 2 /* cnf_loss_adaptive only: the multistep solve has its own entry points */ };

/* arithmetic of the hidden-layer products (cnf_config.arith).  Both accumulate in f32.
 *   F32       exact f32 MFMA (v_mfma_f32_16x16x4_f32 = fmaf chain) — default, the validated path
 *   BF16X6    each f32 operand split exactly into three bf16 parts, six bf16 MFMAs per product
 *             (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; dropped terms <= 2^-24 relative):
 *             f32-equivalent accuracy at a higher MFMA rate.  Only shapes the per-wave MFMA
 *             kernel covers; cnf_create fails with CNF_ERR_UNSUPPORTED otherwise. */
enum { CNF_ARITH_F32 = 0, CNF_ARITH_BF16X6 = 1 };

/* kernel families (cnf_kernel_path).  AUTO resolves to MFMA (fused whole-solve kernels) when an
 * instance covers the configuration, else to LAYERED (layer-wise evaluation: every product of the chain a
 * hand-written MFMA kernel with the elementwise work in its epilogue, csrc/cnf_lgemm.hip; any Dense chain
 * whose layers have at most 512 outputs and 639 inputs), else to SIMT (thread-per-sample kernels, any Dense
 * chain; 4-100x slower than LAYERED at every batch size measured, kept as the fallback for wider layers and
 * as an independent implementation for the tests). */
enum { CNF_PATH_AUTO = 0, CNF_PATH_SIMT = 1, CNF_PATH_MFMA = 2, CNF_PATH_LAYERED = 3 };

/* Configuration = the ICNF fields and type parameters that reach the hot path
 * (src/core/icnf.jl:16-141). */
typedef struct {
    int32_t nvars;                        /* nvariables */
    int32_t naug;                         /* naugments (AUGMENTED = naug != 0) */
    int32_t ncond;                        /* nconditions (CONDITIONED = ncond != 0) */
    int32_t autonomous;                   /* 1: no time row in the MLP input */
    int32_t n_layers;                     /* Dense layers in the Chain */
    int32_t widths[CNF_MAX_LAYERS + 1];   /* widths[0]=n_in=D+!autonomous+ncond, ..., widths[n_layers]=D */
    int32_t acts[CNF_MAX_LAYERS];         /* CNF_ACT_* per layer */
    int32_t mode;                         /* CNF_MODE_* */
    int32_t nprobes;                      /* K Hutchinson probes (reference: 1) */
    int32_t reg_z;                        /* Edot=|zdot|_2 on: NORM_Z and TrainMode{true} (icnf.jl:184-199) */
    int32_t reg_j;                        /* ndot=|eps^T J|_2 on: NORM_J and TrainMode{true} (icnf.jl:229-245) */
    int32_t reg_aug;                      /* Adot=|z_aug|_2 on: NORM_Z_AUG, AUGMENTED, TrainMode{true} (base_icnf.jl:106-122) */
    int32_t device_id;                    /* HIP device ordinal */
    int32_t kernel_path;                  /* CNF_PATH_*; AUTO picks MFMA when the shape is supported */
    int32_t arith;                        /* CNF_ARITH_* */
} cnf_config;

typedef struct cnf_handle cnf_handle;

/* Tuning switchboard: every A/B and test switch of the library in one place.  The defaults are what ships; the environment
 * variable CNF_<FIELD IN CAPITALS> overrides a field when the library first consults the board and again on cnf_set_tuning(NULL)
 * - a hook for tests and A/B runs, not configuration; cnf_set_tuning replaces the whole board.  Creating a handle never touches
 * it (the library creates internal handles itself).  Process-wide: the switches select among kernels that compute the same
 * thing (parity tests cross-check them), so they are not part of a handle's identity.  Thread safety: the board is published
 * as an immutable snapshot behind one atomic pointer, so cnf_get_tuning / cnf_set_tuning may be called from any thread; a call
 * of the library that is in flight on another thread while the board changes may run under the old board or the new one. */
typedef struct cnf_tuning {
    int32_t tile_split;            /* CNF_TILE_SPLIT, default 1: 1: whole fixed-step solves of per-wave shapes with at most one 16-sample tile per CU take the tile-split kernel; 0: never; 2: always */
    int32_t solve2;                /* CNF_SOLVE2, default 2: whole fixed-step solves of one-probe VJP flows without conditions on the hand-scheduled per-wave kernel (csrc/cnf_mfma2.hip; same bits) with two waves per SIMD; 1: one wave per SIMD; 0: mfma_solve_kernel */
    int32_t solve2_pair;           /* CNF_SOLVE2_PAIR, default 1: batches of at most two 16-sample tiles per CU of two-tile nets (cfg1) run two waves per tile (forward chain / pullback: csrc/cnf_mfma2.hip); 0: one */
    int32_t coopd;                 /* CNF_COOPD, default 1: dealt cooperative kernels (csrc/cnf_coop_d*.hip): 1 above 4096 columns where they serve the plan; 0 off; 2 at every batch; 3 as 1 with the Runge-Kutta rows of the (2, 12) instance in the plan's global ring instead of LDS (A/B, bit-identical) */
    int32_t coopd_grad;            /* CNF_COOPD_GRAD, default 1: dealt reverse sweep (csrc/cnf_coop_dgrad.hip): 1 where it has an instance; 0 off (section 8.4's sweep); 2 forced */
    int32_t coop_grad;             /* CNF_COOP_GRAD, default 1: cooperative reverse sweep (gradient path 3); 0: those shapes train layer-wise */
    int32_t coop_grad_mid;         /* CNF_COOP_GRAD_MID, default 1: auxiliary cooperative plan for the gradient of the slab shapes of 5 - 8 hidden tiles (7 - 8: every batch size, 5 - 6: up to 8192 columns); N > 1: from N columns on; 0 off */
    int32_t coop_grad3;            /* CNF_COOP_GRAD3, default 1: the cooperative gradient in its second form: the forward solve stores h_l and delta_l of every stage, the sweep (csrc/cnf_coop_grad3.hip) runs the second-order chains alone - with two workgroups per CU on two hidden layers once there are more 32-sample super-tiles than CUs (csrc/cnf_coop_grad3w.hip); 2: the one-workgroup-per-CU sweep for every shape and size, 3: the two-per-CU sweep at every size (A/B, tests; the two agree bit for bit); 0: the sweeps that recompute both chains */
    int32_t coop_grad3_gib;        /* CNF_COOP_GRAD3_GIB, default 96: GiB of HBM the stage store of that form may take (of 288); larger batches take the recomputing sweeps */
    int32_t grad_layered;          /* CNF_GRAD_LAYERED, default 0: 1: every gradient takes the layer-wise path (A/B, cross-checks) */
    int32_t jvp_grad_twin;         /* CNF_JVP_GRAD_TWIN, default 1: JVP mode without the |J eps| regulariser trains through the VJP mode's fused sweeps; 0: its own layer-wise gradient */
    int32_t probe_grad_twin;       /* CNF_PROBE_GRAD_TWIN, default 1: K > 1 probes whose own gradient is layer-wise train probe by probe on the one-probe cooperative sweep - two hidden layers always, three where the one-probe call takes the second form (1.22 x at 3 x 256, K = 4); 2: three hidden layers always (0.96 x on the recomputing sweeps); 0: never */
    int32_t adaptive_ckpt;         /* CNF_ADAPTIVE_CKPT, default 1: the adaptive Tsit5 solve of cnf_loss_grad_adaptive writes the checkpoints of the frozen-grid gradient itself, and the slab-accumulator gradient reads those of the loss solve (no second forward pass); 0: every gradient runs its own */
    int32_t layered_loss_by_solve; /* CNF_LAYERED_LOSS_BY_SOLVE, default 0: 1: the layer-wise gradient takes its loss from a separate solve instead of accumulating it in the sweep */
    int32_t device_controller;     /* CNF_DEVICE_CONTROLLER, default 1: one-launch adaptive Tsit5 / VCABM with the step controller on the device where the batch fits; 0: host loop */
    int32_t dc_per_cu;             /* CNF_DC_PER_CU, default 2: most workgroups per CU of the one-launch adaptive kernels (the occupancy query bounds it: 2 only for the exact-shape VCABM instances of nets of <= 2 hidden tiles with D <= 4, and only for batches that do not fit one per CU) */
    int32_t mfma_coop;             /* CNF_MFMA_COOP, default 0: 1: cnf_create prefers the cooperative kernel where a per-wave instance also fits (tests) */
    int32_t mfma_coopx;            /* CNF_MFMA_COOPX, default 1: extended cooperative kernel (csrc/cnf_coop_x.hip); 0 off */
    int32_t mfma_nt;               /* CNF_MFMA_NT, default 0: threads per workgroup of the per-wave solve kernel (0: the instance's own) */
    int32_t mfma_pre;              /* CNF_MFMA_PRE, default -1: hoisting level of the per-wave solve kernel (-1: the best instance) */
    int32_t mfma_prio;             /* CNF_MFMA_PRIO, default 0: s_setprio scheme of the per-wave solve kernel (A/B; no scheme won) */
    int32_t mfma_queue;            /* CNF_MFMA_QUEUE, default 0: 1: dynamic tile queue instead of the static stride (A/B) */
    int32_t cg_one_per_cu;         /* CNF_CG_ONE_PER_CU, default 0: 1: one workgroup per CU for the cooperative reverse sweep (A/B) */
    int32_t layered_min_b;         /* CNF_LAYERED_MIN_B, default 0: batches below this of an AUTO-resolved layer-wise handle take the SIMT kernels (0: never) */
    int32_t layered_kc;            /* CNF_LAYERED_KC, default 0: column chunk of the layer-wise weight-cotangent products (0: lg_wgrad_chunks' choice) */
    int32_t layered_no_kckpt;      /* CNF_LAYERED_NO_KCKPT, default 0: 1: the layer-wise reverse sweep recomputes the stage derivatives instead of keeping them */
    int32_t layered_act_gib;       /* CNF_LAYERED_ACT_GIB, default 48: GiB of stage activations the layer-wise gradient may keep (of the 288 GB of HBM) */
    int32_t lg_gemm;               /* CNF_LG_GEMM, default 2: generic product kernel: 2 = lg_gemm2 (double-buffered sub-panels), 1 = the round-1 kernel */
    int32_t lg_spw;                /* CNF_LG_SPW, default 0: lg_gemm2: sub-panels per workgroup (0: by shape) */
    int32_t lg_nw;                 /* CNF_LG_NW, default 4: lg_gemm2: waves per workgroup for K <= 272 */
    int32_t lg_gemm2_wide;         /* CNF_LG_GEMM2_WIDE, default 1: lg_gemm2's eight-wave instance for K = 273 .. 512; 0: the round-1 kernel */
    int32_t lg_wgrad_per_cu;       /* CNF_LG_WGRAD_PER_CU, default 0: lg_wgrad workgroups per CU (0: 3 / 6 / 12 by the number of workgroups sharing a chunk) */
    int32_t lg_wgrad_t1;           /* CNF_LG_WGRAD_T1, default 5: lg_wgrad: chunk-sharing threshold for 6 per CU */
    int32_t lg_wgrad_t2;           /* CNF_LG_WGRAD_T2, default 8: lg_wgrad: chunk-sharing threshold for 12 per CU */
} cnf_tuning;
int cnf_get_tuning(cnf_tuning* out);
int cnf_set_tuning(const cnf_tuning* in);   /* NULL: defaults + the CNF_* environment variables, read again */

int cnf_version(void);
const char* cnf_last_error(void);
/* Which compiler made this library: "hipcc <version line>; clang <version>; flags ..." recorded at build time (a log can then say
 * which compiler produced the code objects whose hand-placed wait states tests/test_isa_hazards.py audits).  Never NULL. */
const char* cnf_build_info(void);

/* ICNF(; ...) constructor (src/core/icnf.jl:53-141): validates and binds a config to a device. */
int cnf_create(cnf_handle** out, const cnf_config* cfg);
int cnf_destroy(cnf_handle* h);

/* Parameters `ps`: the ComponentArray of LuxCore.setup (test/ci_tests/smoke_tests.jl:61-62) —
 * a flat Float32 vector holding, per Dense layer, weight (out x in, column-major: W(o,i) at
 * w_off[l] + o + out*i) and bias (out, at b_off[l]).  n = length(p).  The library copies and
 * repacks; call again whenever ps changes.  p_is_device: 1 if p is a device pointer.
 * Ordering: the repack is enqueued on `stream` (for a device pointer on the fused path it is one
 * gather kernel per operand image, no host round trip and no synchronisation, so a training loop
 * that updates ps on the device can call this every step); later calls on the same stream see the
 * new parameters, calls on other streams must be ordered after it by the caller.  A host p may be
 * reused as soon as the call returns. */
int cnf_set_params(cnf_handle* h, const float* p, size_t n, const size_t* w_off,
                   const size_t* b_off, int p_is_device, void* stream);

/* ---- adaptive stepping: the caller drives an embedded Runge-Kutta solve -------------
 * base_sol with an adaptive `alg` (src/core/base_icnf.jl:134-140) lets OrdinaryDiffEq choose the steps from an
 * error norm over the WHOLE S x B state, so samples (and column shards) are coupled through the step-size
 * controller.  The library provides the per-attempt device work; the controller (a few scalars per attempt, and
 * an all-reduce of err_sumsq when the columns are sharded) stays with the caller.
 *
 * cnf_step_embedded: one attempt of Tsit5 from (t, u) with step dt:
 *     u_new = u + dt sum_i b_i k_i                                   (5th order)
 *     err_sumsq[0] = sum over all S*B entries of (dt sum_i btilde_i k_i / (abstol + reltol max(|u|, |u_new|)))^2
 * (OrdinaryDiffEq's calculate_residuals + ODE_DEFAULT_NORM before the division by the length and the root).
 * u, u_new: device S x B, must not alias.  err_sumsq: device double.  flags: CNF_STEP_FSAL = this attempt starts
 * where the previous (accepted) attempt of this handle ended, so its first stage is that attempt's last one;
 * CNF_STEP_RETRY = same (t, u) as the previous (rejected) attempt, the first stage is reused.  0 = evaluate it. */
enum { CNF_STEP_FSAL = 1, CNF_STEP_RETRY = 2 };
int cnf_step_embedded(cnf_handle* h, int alg, int flags, float t, float dt, const float* u, const float* eps,
                      const float* ys, int64_t B, float abstol, float reltol, float* u_new, double* err_sumsq,
                      void* stream);

/* u0 = [x; 0] (S x B) for an integration driven by the caller (inference_prob, src/core/base_icnf.jl:247-270), and
 * the inference_sol epilogue on a final state: logp (B), regs (3B, may be NULL) (src/core/base_icnf.jl:158-172). */
int cnf_assemble_u0(cnf_handle* h, const float* x, int64_t B, float* u0, void* stream);
int cnf_epilogue(cnf_handle* h, const float* u, int64_t B, float* logp, float* regs, void* stream);

/* The reference's DEFAULT solver: `alg = VCABM()`, reltol = abstol = 1e-4 (src/core/icnf.jl:84-89), run by
 * SciMLBase.solve in base_sol (src/core/base_icnf.jl:134-140).  VCABM lives in OrdinaryDiffEqAdamsBashforthMoulton
 * (compat "2", not vendored): the variable-step variable-order Adams predictor-corrector in divided-difference form
 * (Hairer, Noersett, Wanner I, III.5; Shampine & Gordon's PECE step).  The library keeps the multistep state - u_n, f_n,
 * the modified divided differences Phi*_j(n-1), the accepted step sizes - on the device; the caller drives the step-size
 * and order policy (the host side of the reference's solver), reading one to three error sums per step:
 *
 *   cnf_vcabm_begin    u_0 (S x B, device) at t0; evaluates f_0; forgets any earlier history.
 *   cnf_vcabm_attempt  one PEC pass of order k (= number of predictor terms, 1 <= k <= min(12, accepted steps + 1), and at most
 *                      one more than the order of the last accepted step: the stored differences end there) with
 *                      step dt (either sign): p = u_n + dt sum_{j<k} g_j Phi*_j(n), f(p, t_n + dt),
 *                      u_{n+1} = p + dt g_k Phi_k(n+1).  err3 (device, 3 doubles) receives the sums over the whole S x B
 *                      state of the squared local error estimates of orders k, k-1, k-2, each scaled by
 *                      abstol + reltol max(|u_n|, |u_{n+1}|):  dt (g_j - g_{j-1}) Phi_j(n+1), j = k, k-1, k-2 (0 where
 *                      k is too small).  The state is unchanged: a rejected attempt is simply repeated with another dt / k.
 *   cnf_vcabm_accept   commits the pending attempt: f_{n+1} = f(u_{n+1}, t_{n+1}) (the final E), history rotation.
 *                      err_up (device, 1 double, or NULL): squared sum of the order k+1 estimate
 *                      dt gamma*_{k+1} Phi_{k+1}(n+1) from the re-evaluated derivative (needs k accepted steps, k < 12).
 *   cnf_vcabm_state    copies the current state into u_out (S x B, device; may be NULL), current time into *t_out (host).
 *
 * eps / ys as in cnf_aug_f, the same arrays on every call of one solve.  Sharded solves all-reduce the sums. */
#define CNF_VCABM_MAX_ORDER 12
int cnf_vcabm_begin(cnf_handle* h, float t0, const float* u0, const float* eps, const float* ys, int64_t B, void* stream);
int cnf_vcabm_attempt(cnf_handle* h, int order, float dt, const float* eps, const float* ys, int64_t B, float abstol,
                      float reltol, double* err3, void* stream);
int cnf_vcabm_accept(cnf_handle* h, const float* eps, const float* ys, int64_t B, float abstol, float reltol,
                     double* err_up, void* stream);
int cnf_vcabm_state(cnf_handle* h, int64_t B, float* u_out, double* t_out, void* stream);

/* The whole default solve in one call: u1 = solve(u' = augmented_f, u0, (t0, t1), VCABM(); abstol, reltol) - the passes above
 * under the solver's own policy (order ramp 1 -> 3, then order selection from the four error estimates; integral step-size
 * controller, gamma = 9/10, q in [1/5, 10]; Hairer's initial step unless dt_init != 0), run on the host inside the library.
 * For single-process callers (a sharded solve must all-reduce the sums between the passes and drives them itself).
 * u0, u1: S x B device arrays (may alias).  stats (host, may be NULL) and the first record_cap accepted step sizes / orders
 * (host arrays, may be NULL) are filled in; the call synchronises `stream`. */
typedef struct cnf_solve_stats { int32_t naccept, nreject, nf, max_order; } cnf_solve_stats;
int cnf_solve_vcabm(cnf_handle* h, float t0, float t1, const float* u0, const float* eps, const float* ys, int64_t B,
                    float abstol, float reltol, float dt_init, int maxiters, float* u1, cnf_solve_stats* stats,
                    float* dts_out, int32_t* orders_out, int32_t record_cap, void* stream);

/* Adaptive Tsit5 (OrdinaryDiffEq's PI controller: beta1 = 7/50, beta2 = 2/25, gamma = 9/10, q in [1/5, 10]; Hairer's initial
 * step unless dt_init != 0) from t0 to t1 in one call - cnf_step_embedded attempts driven inside the library, or, for batches
 * that fit the chip's wave slots on a fused kernel, the whole solve with the controller on the device in one launch
 * (cnf_solve_controller); arguments as cnf_solve_vcabm (max_order is reported as 5).  Single process; synchronises `stream`. */
int cnf_solve_tsit5(cnf_handle* h, float t0, float t1, const float* u0, const float* eps, const float* ys, int64_t B,
                    float abstol, float reltol, float dt_init, int maxiters, float* u1, cnf_solve_stats* stats,
                    float* dts_out, int32_t record_cap, void* stream);

/* `loss(icnf, mode, xs, ps, st)` under an adaptive solver in ONE call (src/core/icnf.jl:628-649 over src/core/base_icnf.jl:134-172,
 * 256-266): u0 = vcat(xs, 0), the solve (alg = CNF_ALG_VCABM, the reference's default: cnf_solve_vcabm; CNF_ALG_TSIT5:
 * cnf_solve_tsit5), the inference epilogue and mean(-logp + l1 E + l2 n + l3 A) - the same kernels in the same order as
 * cnf_assemble_u0 / cnf_solve_* / cnf_epilogue / cnf_loss_mean, so the same bits; what it saves is the host side of four calls,
 * which at the reference's own batch sizes (2^10 samples, benchmark/benchmarks.jl) is a sixth of the call.
 * x: nvars x B; lambdas: 3 doubles (host); loss: 1 float (device); sums4 (device, may be NULL): the four column sums;
 * logp / regs (device, B and 3 B floats, may be NULL): the per-sample outputs of the epilogue; stats, dts_out, orders_out
 * (host, may be NULL; orders_out is not written under CNF_ALG_TSIT5) as in cnf_solve_vcabm.  B >= 1.  Single process;
 * synchronises `stream` (the solve does). */
int cnf_loss_adaptive(cnf_handle* h, int alg, float t0, float t1, const float* x, const float* eps, const float* ys, int64_t B,
                      float abstol, float reltol, float dt_init, int maxiters, const double* lambdas, float* loss, float* sums4,
                      float* logp, float* regs, cnf_solve_stats* stats, float* dts_out, int32_t* orders_out, int32_t record_cap,
                      void* stream);

/* Which kernel family the handle resolved to (CNF_PATH_SIMT, CNF_PATH_MFMA or CNF_PATH_LAYERED). */
int cnf_kernel_path(const cnf_handle* h);

/* CNF_PATH_MFMA covers four kernel organisations; hosts and tests ask here instead of inferring it from the environment.
 *   PER_WAVE    one wave per 16-sample tile, operand images in LDS (csrc/cnf_mfma_kernel.h)        hidden width <= 128
 *   COOP        a workgroup per 64-sample super-tile, images in L2 (csrc/cnf_coop.hip)             Hutchinson VJP, 1 probe, wide layers
 *   COOPX       its extended form (csrc/cnf_coop_x.hip)                                            conditions / probes / exact / JVP, wide layers
 *   COOPD       the cooperative kernel with its tiles dealt exactly over four owner waves (csrc/cnf_coop_d.hip)   one-probe VJP, the reference's default architecture at nvariables >= 16
 *   TILE_SPLIT  one tile per workgroup, hidden width over the four SIMDs (csrc/cnf_coop.hip)       per-wave shapes at <= 16 CUs' worth of columns
 * cnf_kernel_family: the handle's own family (never TILE_SPLIT).  cnf_kernel_family_for: the kernel a call of B columns takes -
 * whole_solve != 0 for cnf_integrate_fixed / cnf_inference_fixed(_dt) / cnf_loss_*, 0 for cnf_aug_f and the caller-driven steps.
 * Results of two families differ by summation order only (<= 2e-5 in logp); within one family a column's result does not depend
 * on which other columns are in the call - SHARD CONCATENATION IS THEREFORE BIT-IDENTICAL ONLY WHEN EVERY SHARD AND THE
 * UNSHARDED CALL TAKE THE SAME FAMILY: per-wave shapes switch to TILE_SPLIT at B <= 16 x (compute units) = 4096 columns
 * (CNF_TILE_SPLIT=0 in the environment keeps PER_WAVE at every size), extended-kernel shapes of 5 .. 15 hidden tiles to COOPD
 * above it (CNF_COOPD=0 keeps COOPX; 16 .. 24 hidden tiles take COOPD at every size); every BASELINE shard is larger. */
enum { CNF_FAMILY_SIMT = 0, CNF_FAMILY_PER_WAVE = 1, CNF_FAMILY_COOP = 2, CNF_FAMILY_COOPX = 3, CNF_FAMILY_TILE_SPLIT = 4, CNF_FAMILY_LAYERED = 5,
       CNF_FAMILY_COOPD = 6 };
int cnf_kernel_family(const cnf_handle* h);
int cnf_kernel_family_for(cnf_handle* h, int64_t B, int whole_solve);
/* Name of the kernel instance behind the handle ("mfma_vjp<HT=4,L=3,...>", "coopx<...>", "layered", "simt"); static storage
 * owned by the handle. */
const char* cnf_kernel_name(const cnf_handle* h);

/* Where the step (and order) policy of the handle's last cnf_solve_vcabm / cnf_solve_tsit5 / cnf_loss_grad_adaptive ran:
 * 1 = on the device, the whole solve in one cooperative launch (fused per-wave kernels, batches of at most one 16-sample tile
 * per resident wave); 0 = the host loop over per-attempt launches (any other case, or CNF_DEVICE_CONTROLLER=0 in the
 * environment); CNF_ERR_INVALID before the first such call. */
int cnf_solve_controller(const cnf_handle* h);

/* How the last cnf_set_params repacked: 1 = gather kernels on the device (fused path, f32 images),
 * 0 = on the host (SIMT parameter copy, split-bf16 images).  CNF_ERR_NO_PARAMS before the first call. */
int cnf_repack_on_device(const cnf_handle* h);

/* augmented_f(u,p,t, icnf, mode, nn, st, eps) / augmented_f(du,u,p,t, ...) for MatrixMode:
 * src/core/icnf.jl:517-559 (VecJac), :561-603 (JacVec), :297-339 (TestMode) — the callable
 * make_ode_func hands to the ODE solver (src/core/base_icnf.jl:62-78).
 *   u, du : S x B      eps : (K*D) x B (probe k = rows k*D..k*D+D-1; NULL in EXACT mode)
 *   ys    : C x B or NULL      t : scalar time.  du may not alias u. */
int cnf_aug_f(cnf_handle* h, float* du, const float* u, float t, const float* eps,
              const float* ys, int64_t B, void* stream);

/* base_sol(icnf, prob) = last(solve(prob; alg, adaptive=false, dt=(t1-t0)/nsteps).u):
 * src/core/base_icnf.jl:134-140.  u0, u1: S x B (u1 may alias u0).  Stage i of step n is
 * evaluated at t0 + n*dt + c_i*dt.  t1 < t0 integrates backwards (generate_prob's reversed
 * tspan, src/core/base_icnf.jl:351-376). */
int cnf_integrate_fixed(cnf_handle* h, int alg, int nsteps, float t0, float t1, const float* u0,
                        const float* eps, const float* ys, int64_t B, float* u1, void* stream);

/* The same solve as the user states it to OrdinaryDiffEq: sol_kwargs = (alg, adaptive = false, dt) on tspan (t0, t1) —
 * steps of |dt| towards t1 and a SHORTER LAST STEP that lands on t1 (t1 is a tstop of the integrator); when what remains
 * after the full steps is within 100 eps(Float32) max(|t0|, |t1|) — OrdinaryDiffEq's floating-point fix-up for tstops —
 * the span is divided equally instead.  This is what STEER needs: steer_tspan (src/core/base_icnf.jl:23-43) draws t1 per
 * call, so |t1 - t0| is not a multiple of dt.  Two launches of the fused solve kernel (full steps, then the tail).
 * |t1 - t0| < tolerance copies u0 to u1.  cnf_inference_fixed_dt is cnf_inference_fixed on that grid. */
int cnf_integrate_fixed_dt(cnf_handle* h, int alg, float dt, float t0, float t1, const float* u0, const float* eps,
                           const float* ys, int64_t B, float* u1, void* stream);
int cnf_inference_fixed_dt(cnf_handle* h, int alg, float dt, float t0, float t1, const float* x, const float* eps,
                           const float* ys, int64_t B, float* logp, float* regs, float* u_final, void* stream);

/* inference(icnf, mode, xs[, ys], ps, st) for MatrixMode with a fixed-step solver, fused:
 * inference_prob (u0 = [x; 0]) + base_sol + inference_sol (logp = logpdf(N(0,I), z) - dlogp):
 * src/core/base_icnf.jl:247-296, 134-140, 158-172, 406-425.  eps is an explicit input (the
 * reference draws it from icnf.rng at :258-259).
 *   x: nvars x B.  logp: B.  regs: 3*B laid out [Edot (B) | ndot (B) | Adot (B)] or NULL.
 *   u_final: S x B or NULL. */
int cnf_inference_fixed(cnf_handle* h, int alg, int nsteps, float t0, float t1, const float* x,
                        const float* eps, const float* ys, int64_t B, float* logp, float* regs,
                        float* u_final, void* stream);

/* The reduction of loss(icnf, mode, xs[, ys], ps, st) (src/core/icnf.jl:628-649) before the
 * mean: sums4 (device, 4 floats) = [sum(-logp), sum(Edot), sum(ndot), sum(Adot)] over the B
 * columns given.  loss = (s0 + l1*s1 + l2*s2 + l3*s3) / B_global after the caller's
 * all-reduce of these four scalars.  Deterministic (fixed-order tree, no float atomics). */
int cnf_loss_sums(cnf_handle* h, const float* logp, const float* regs, int64_t B, float* sums4,
                  void* stream);

/* The scalar loss(icnf, mode, xs[, ys], ps, st) of an UNSHARDED batch in the same two reduction kernels:
 * loss[0] (device) = (s0 + l1 s1 + l2 s2 + l3 s3) / B, combined in double; lambdas = {l1, l2, l3} (host doubles);
 * sums4 (device, may be NULL) as cnf_loss_sums.  B >= 1. */
int cnf_loss_mean(cnf_handle* h, const float* logp, const float* regs, int64_t B, const double* lambdas, float* sums4,
                  float* loss, void* stream);

/* Gradient of the summed negative log-density with respect to the parameters, through the
 * fixed-step solve (discretise-then-optimise reverse mode):
 *     grad[k] = d/dp_k  sum_j ( -logp_j + l1 Edot_j + l2 ndot_j + l3 Adot_j )    over the B columns given
 * (lambdas = {l1, l2, l3}, host pointer; a term is active only if the handle's reg_z / reg_j /
 * reg_aug flag is set, i.e. for TrainMode{true} with a non-zero lambda, src/core/icnf.jl:628-637),
 * the quantity Zygote obtains for `loss` via QuadratureAdjoint + ZygoteVJP in the reference's
 * training loop (src/core/icnf.jl:90-99; src/exts/mlj_ext/core_icnf.jl:42-51), here exact for the
 * discrete solve.  grad: device, n floats in the layout of cnf_set_params' p (overwritten).
 * grad_x (device, may be NULL): nvars x B, the gradient of the same sum with respect to the data columns
 * (DI.gradient wrt x in test/ci_tests/smoke_tests.jl) - the costate at t0, free in the reverse sweep.
 * sums4 (device, may be NULL): as cnf_loss_sums.  The caller all-reduces grad and sums4 across
 * column shards and divides by the global column count.
 * Every configuration is covered.  Fused reverse-sweep kernels: 1 <= K <= 8 probes,
 * <= 16 conditions, 2 or 3 equal hidden layers (tanh or softplus) of width <= 64, D + !autonomous <= 15;
 * slab-accumulator kernel: two hidden layers of width <= 128, D + !autonomous <= 32, <= 16 conditions, one probe (the
 * reference's default nets for 7..15 variables); every other shape (wider or more layers, mixed activations, larger D): layer-wise
 * reverse sweep on the library's own product kernels (csrc/cnf_lgemm.hip), which also serves the Hutchinson JVP mode.  Exact-trace mode (TestMode):
 * -tr J = -sum_k e_k^T J e_k, the pullback with the D unit vectors as probes of weight 1 (eps is ignored, no
 * regularisers) - on the several-probe fused kernel where the shape has one, layer-wise otherwise.
 * FFJORD and RNODE losses. */
int cnf_loss_grad_fixed(cnf_handle* h, int alg, int nsteps, float t0, float t1, const float* x,
                        const float* eps, const float* ys, int64_t B, const float* lambdas,
                        float* grad, float* grad_x, float* sums4, void* stream);

/* The same sum and gradients on a NON-UNIFORM grid of fixed steps: step n runs from tgrid[n] to tgrid[n+1]
 * (tgrid: HOST array of nsteps + 1 times).  This is how a loss evaluated with the adaptive solver is
 * differentiated: the accepted steps are frozen and the discrete solve on that grid is reversed (the dependence
 * of the step sizes on the parameters is ignored, the usual discretise-then-optimise convention).  Served by the same
 * implementations as cnf_loss_grad_fixed (cnf_grad_path): the fused reverse-sweep kernels read the step times from device
 * memory (their checkpointing forward pass is one launch of the solve kernel per step), the layer-wise path takes them
 * from the host; the loss sums come from the same grid solve. */
int cnf_loss_grad_grid(cnf_handle* h, int alg, int nsteps, const float* tgrid, const float* x, const float* eps,
                       const float* ys, int64_t B, const float* lambdas, float* grad, float* grad_x, float* sums4,
                       void* stream);

/* The training step under an adaptive solver in one call: the adaptive Tsit5 solve from x (cnf_solve_tsit5), its accepted
 * steps frozen into a grid, then cnf_loss_grad_grid on that grid.  This is what serves `Zygote.gradient` of a loss evaluated
 * with the reference's default sol_kwargs (VCABM at 1e-4: a multistep recurrence has no one-step discrete adjoint, so the
 * adaptive Tsit5 discretisation at the same tolerances is differentiated; the reference's own QuadratureAdjoint gradient is
 * likewise a separate solve that matches the forward pass to tolerance).  HOW FAR this gradient is from the gradient of the
 * exact flow's loss - what both it and the reference's adjoint approximate - is measured in the fp64 oracle
 * (profiles/adaptive_gradient_gap.py -> profiles/r6/r6f_adaptive_gradient_gap.json, bounded by
 * tests/test_oracle_gradient.py::test_frozen_grid_gradient_against_the_exact_flow): along random directions, against central
 * differences of the loss on a VCABM solve at tolerance 1e-10, the frozen-grid gradient at 1e-4 is off by 1e-8 / 7e-8 relative for the
 * reference's default architecture at nvariables = 1 / 8 with freshly initialised weights (four Tsit5 steps resolve that flow) and
 * by 3e-5 / 2e-3 with the weights scaled threefold (22 / 39 steps) - while differences of the VCABM solve at its own 1e-4
 * (step decisions moving with the parameters) are off by 1e-4 / 2.5e-2.  tgrid_out (host, may be NULL): the first grid_cap
 * grid times; stats (host, may be NULL).  Single process; synchronises `stream`. */
int cnf_loss_grad_adaptive(cnf_handle* h, float t0, float t1, const float* x, const float* eps, const float* ys, int64_t B,
                           float abstol, float reltol, float dt_init, int maxiters, const float* lambdas, float* grad,
                           float* grad_x, float* sums4, cnf_solve_stats* stats, float* tgrid_out, int32_t grid_cap,
                           void* stream);

/* Which implementation cnf_loss_grad_fixed / cnf_loss_grad_grid use for this handle: 0 = none (CNF_ERR_UNSUPPORTED),
 * 1 = fused reverse-sweep kernel (cnf_grad.hip / cnf_grad_probes.hip / cnf_grad_slab.hip), 2 = layer-wise reverse sweep on
 * the product kernels of cnf_lgemm.hip (cnf_layered.hip), 3 = cooperative reverse sweep for wide tanh nets (FFJORD or the regularised objective) on uniform
 * steps or on the caller's grid (cnf_coop_grad.hip: one launch per RK step + deferred weight-cotangent products). */
int cnf_grad_path(const cnf_handle* h);

/* cnf_grad_path is a batch-independent HINT (the family a handle's shape belongs to).  The implementation a particular call
 * takes also depends on its column count, its solver and whether it runs on a caller's grid: two-hidden-layer nets of 7-8 hidden
 * tiles report 1 but run on 3 from 4096 columns on; a cooperative shape reports 3 but runs on 2 beyond the sweep's 32-bit operand
 * addressing (2^31 / (8 stages (H + 1)) columns), or on a grid its forward kernel cannot checkpoint on.  This entry returns
 * what cnf_loss_grad_fixed (on_grid = 0) / cnf_loss_grad_grid, cnf_loss_grad_adaptive (on_grid = 1) WILL take for B columns
 * with `alg` (CNF_ALG_RK4 / CNF_ALG_TSIT5); same codes. */
int cnf_grad_path_for(const cnf_handle* h, int64_t B, int alg, int on_grid);

/* Which FORM of the cooperative reverse sweep (code 3 above) a call of B columns and `nsteps` steps of `alg` takes: 0 = the call
 * does not take the cooperative sweep; 1 = the sweeps that recompute the forward chain and the first-order pullback per stage
 * (cnf_coop_grad.hip, cnf_coop_dgrad.hip; operands of the weight cotangents as column-major arrays); 2 = the second form
 * (round 6): the checkpointing forward solve stores h_l and delta_l of every stage (2 L H floats per
 * sample and stage of HBM, bounded by cnf_tuning.coop_grad3_gib), the sweep (cnf_coop_grad3.hip) runs the second-order chains
 * alone, the weight cotangents are products over tiles (cnf_wgrad_tiles.hip).  Same gradient to rounding. */
int cnf_grad_form_for(const cnf_handle* h, int64_t B, int alg, int nsteps, int on_grid);

/* ---- column shards: the one exchange step of the path -----------------------------------------
 * Under fixed-step integration every column (sample) is independent (src/core/icnf.jl:530-535 is column-wise), so rank r
 * of G evaluates a contiguous column block with a full weight replica and no data-path collective.  The only exchange is
 * the mean in `loss` (src/core/icnf.jl:636): an all-reduce of the four partial sums of cnf_loss_sums and the column
 * count.  The reference has no multi-device path; these entries are what a sharded Julia host calls in its place.
 * RCCL (ncclAllReduce, ncclSum) over xGMI; librccl.so.1 is resolved with dlopen at first use (CNF_ERR_COMM if absent).
 *
 *   cnf_comm_unique_id   rank 0 obtains the 128-byte RCCL id (ncclGetUniqueId) and ships it to the other ranks by whatever
 *                        channel the host has (MPI, a file, a TCP store).
 *   cnf_comm_init        one rank per process: ncclCommInitRank(nranks, id, rank) on device_id (collective over the ranks).
 *   cnf_comm_init_all    one process driving ndev devices: ncclCommInitAll; out receives ndev communicators (rank i on devs[i]).
 *                        Calls on several of them from one thread must be bracketed by cnf_comm_group_start / _end.
 *   cnf_allreduce_loss   out5 (device, 5 doubles) = sum over ranks of [sums4[0..3] (as double), B_local]: enqueued on
 *                        `stream` behind the cnf_loss_sums that produced sums4 (device, 4 floats).  Every rank then has
 *                        loss = (out5[0] + l1 out5[1] + l2 out5[2] + l3 out5[3]) / out5[4].  40 bytes: latency-bound.
 *   cnf_allreduce_sum    in-place sum of `count` elements (CNF_DTYPE_F32 / CNF_DTYPE_F64) of a device buffer: the gradient of
 *                        cnf_loss_grad_* (nparams floats) and the squared error sums of the adaptive solvers (doubles), which
 *                        couple the shards through the step-size controller. */
#define CNF_COMM_ID_BYTES 128
enum { CNF_DTYPE_F32 = 0, CNF_DTYPE_F64 = 1 };
typedef struct cnf_comm cnf_comm;
int cnf_comm_unique_id(void* id_out /* CNF_COMM_ID_BYTES, host */);
int cnf_comm_init(cnf_comm** out, int rank, int nranks, const void* id /* CNF_COMM_ID_BYTES, host */, int device_id);
int cnf_comm_init_all(cnf_comm** out /* ndev entries */, int ndev, const int* devs);
int cnf_comm_destroy(cnf_comm* c);
int cnf_comm_rank(const cnf_comm* c);
int cnf_comm_size(const cnf_comm* c);
int cnf_comm_group_start(void);
int cnf_comm_group_end(void);
int cnf_allreduce_loss(cnf_comm* c, const float* sums4, int64_t B_local, double* out5, void* stream);
int cnf_allreduce_sum(cnf_comm* c, void* buf, size_t count, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CNF_H */
