// cnf_internal.h — internal interfaces between the C-ABI layer and the kernel files.
#pragma once
#include <string>

#include "cnf_common.h"

namespace cnf {

// the switchboard (cnf_tuning.hip): current values; re-read from the CNF_* environment variables by cnf_create
const cnf_tuning& tuning();
void tuning_from_env();


// Device-side description of the Dense chain (passed by value as a kernel argument).
struct NetDev {
    int D, C, autonomous, n_layers, maxw;
    int widths[CNF_MAX_LAYERS + 1];
    int acts[CNF_MAX_LAYERS];
    int w_off[CNF_MAX_LAYERS];  // offsets into the device copy of the Lux parameter blob
    int b_off[CNF_MAX_LAYERS];
    int mode, K, reg_z, reg_j;
};

// Runge-Kutta stage combination folded into a kernel prologue:
//   value = u + dt * sum_{j<nprev} coef[j] * k[j]
struct StageIn {
    const float* u;
    const float* k[6];
    float coef[6];
    int nprev;
    float dt;
};

// ---- generic SIMT path (cnf_simt.hip) ----
size_t simt_ws_rows(const NetDev& net);
hipError_t simt_aug_f(const NetDev& net, const float* P, const StageIn& in, float t,
                      const float* eps, const float* ys, int64_t B, float* du, float* ws,
                      int64_t ld, hipStream_t st);
hipError_t rk_update(float* u, const StageIn& in, int64_t n, hipStream_t st);
hipError_t assemble_u0(const float* x, int nvars, int S, int64_t B, float* u, hipStream_t st);
// Zero `bytes` (a multiple of 4) at p with a KERNEL on the stream.  Used instead of hipMemsetAsync everywhere on the hot path:
// inside a captured HIP graph a memset NODE on this stack is not reliably ordered with the kernel nodes around it (round 4:
// a graph of cnf_loss_grad_fixed replayed behind other work on the stream returned a quarter of the gradient entries from a
// slab the memset had not - or not yet - cleared; the same calls made eagerly, or with this kernel, are bit-identical).
hipError_t zero_async(void* p, size_t bytes, hipStream_t st);
// device-to-device copy of `bytes` (a multiple of 4, both pointers 4-byte aligned, no overlap) as a kernel, for the same reason
hipError_t copy_async(void* dst, const void* src, size_t bytes, hipStream_t st);
hipError_t epilogue(const float* u, int nvars, int D, int reg_aug, int64_t B, float* logp,
                    float* regs, hipStream_t st);
hipError_t loss_sums(const float* logp, const float* regs, int64_t B, float* partial,
                     float* sums4, hipStream_t st);
hipError_t loss_mean(const float* logp, const float* regs, int64_t B, float* partial, float* sums4, float* mean_out,
                     const double lam[3], hipStream_t st);
constexpr int kErrBlocks = 1024;  // partial sums of embedded_error
hipError_t embedded_error(const float* u, const float* unew, const float* const* k, const float* btilde, int nk, float dt,
                          float abstol, float reltol, int64_t n, double* partial, double* out, hipStream_t st);

// ---- fused MFMA path (cnf_mfma.hip) ----
struct MfmaPlan;  // opaque: packed-weight layout + kernel selection for one config

// Returns nullptr if the configuration is outside what the MFMA kernels cover.
MfmaPlan* mfma_plan_create(const cnf_config& cfg, bool coop_only = false);   // coop_only: the cooperative kernel's plan or null
void mfma_plan_destroy(MfmaPlan* p);
// bytes of the packed weight image (device buffer the plan needs)
size_t mfma_packed_bytes(const MfmaPlan* p);
// host-side repack of the Lux blob into the MFMA operand image
void mfma_pack(const MfmaPlan* p, const float* lux, const size_t* w_off, const size_t* b_off,
               float* packed);
const char* mfma_plan_name(const MfmaPlan* p);
// the one image region that is not a gather of the parameters (Q of the two-hidden-layer exact trace), and its device packer
bool mfma_plan_q_region(const MfmaPlan* p, size_t* off, size_t* len);
hipError_t mfma_pack_q_device(const MfmaPlan* p, const float* lux_dev, const size_t* w_off, float* packed_dev, hipStream_t st);

struct SolveArgs {
    // exactly one of x (nvars x B, u0 = [x;0]) or u0 (S x B) is non-null
    const float* x;
    const float* u0;
    const float* eps;
    const float* ys;
    int64_t B;
    int nsteps;      // 0 = single dynamics call (aug_f): du written to u_out, evaluated at t0
    int alg;
    float t0, t1;
    float* u_out;    // S x B or null
    float* logp;     // B or null
    float* regs;     // 3B or null
    int nvars, reg_aug;
    float* ckpt;     // optional checkpoint buffers (see KArgs::ckpt, ckpt_k, kfull)
    float* ckpt_k;
    float* kfull;
    float dt_exact;  // != 0: the step itself (nsteps = 1 attempts: (t0 + dt) - t0 is not dt in float32); 0: (t1 - t0) / nsteps
    float* ckpt_g;   // optional, cooperative checkpointing solve only (KArgs::ckpt_g)
    const float* tgrid_dev;   // optional, cooperative checkpointing solve only: nsteps + 1 step times on the device
    int ck_tiles;    // 1: checkpoint rows in tile layout (KArgs::ck_tiles) - only where mfma_plan_ckpt_rows_as_tiles(plan, B) says the solve's kernel writes it
};
hipError_t mfma_solve(MfmaPlan* p, const float* packed_dev, const SolveArgs& a, hipStream_t st);
constexpr int kHostRec = 120;   // accepted steps the one-launch solves also record in pinned host memory (AArgs::host_rec)
// adaptive Tsit5 with the step controller on the device (cnf_mfma_kernel.h: mfma_adaptive_kernel): u0 -> u_out over [t0, t1]
int64_t mfma_adaptive_capacity(MfmaPlan* p);
size_t mfma_adaptive_scratch_bytes(int64_t B, int dts_cap);
hipError_t mfma_solve_adaptive(MfmaPlan* p, const float* packed_dev, const SolveArgs& s, float abstol, float reltol, float dt_init,
                               int maxiters, void* scratch, unsigned* epoch, int dts_cap, int** stats_dev, float** dts_dev, int* host_rec, int ckpt_cap,
                               hipStream_t st);   // s.ckpt / s.ckpt_k with ckpt_cap > 0: the accepted steps' checkpoints (host_rec[6] = 1: complete)
// the default solver VCABM with its passes and its step / order policy on the device (mfma_vcabm_kernel)
int64_t mfma_vcabm_capacity(MfmaPlan* p);
hipError_t mfma_solve_vcabm(MfmaPlan* p, const float* packed_dev, const SolveArgs& s, float abstol, float reltol, float dt_init,
                            int maxiters, void* scratch, unsigned* epoch, int dts_cap, int** stats_dev, float** dts_dev, int** orders_dev, int* host_rec, hipStream_t st);

// ---- parameter gradient (cnf_grad.hip) ----
bool grad_supported(const cnf_config& c);
size_t grad_packed_bytes(const cnf_config& c);
size_t grad_slab_floats(const cnf_config& c, int num_cus);
void grad_shape(const cnf_config& c, int* HT, int* L, int* ZR, int* CR);
void grad_pack(const cnf_config& c, const float* lux, const size_t* w_off, const size_t* b_off, float* packed);
hipError_t grad_launch(const cnf_config& c, const float* packed_dev, const float* ckpt, const float* ckpt_k,
                       int ckpt_zr, const float* eps, const float* ys,
                       const size_t* w_off, const size_t* b_off, int alg, int nsteps, float t0, float t1, const float* tgrid_dev,
                       float probe_w, long long B, const float lam[3], float* slab, float* grad, float* grad_x, int num_cus, hipStream_t st);
// layer-wise evaluation / gradient on the product kernels of cnf_lgemm.hip for everything the fused kernels do not cover (cnf_layered.hip)
struct LayeredGrad;
bool layered_available();   // always: the products are the library's own kernels (cnf_lgemm.hip)
bool layered_supports(const cnf_config& c);   // every layer within the product kernels' limits (512 outputs, 639 inputs)
hipError_t layered_aug_f(LayeredGrad** ctx, const cnf_config& c, const float* P_dev, const size_t* w_off,
                         const size_t* b_off, bool rebuild_params, const StageIn& in, float t, const float* eps,
                         const float* ys, long long B, float* du, hipStream_t st, std::string* err);
bool layered_grad_supported(const cnf_config& c);
void layered_grad_destroy(LayeredGrad* g);
hipError_t layered_grad(LayeredGrad** ctx, const cnf_config& c, const float* P_dev, const size_t* w_off,
                        const size_t* b_off, const float* x, const float* eps, const float* ys, int alg,
                        int nsteps, float t0, float t1, const float* tgrid, long long B, const float lam[3], float* grad,
                        float* grad_x, hipStream_t st, std::string* err, float* logp_out = nullptr, float* regs_out = nullptr);
// logp_out (B) / regs_out (3 B): when both are given the reverse sweep also accumulates the loss terms of the same discrete
// solve (dlogp, E, n per column) and runs the epilogue - no separate forward solve for the loss
void mfma_pack_layout(const cnf_config& c, int HT, int L, int ZR, int CR, const float* lux, const size_t* w_off,
                      const size_t* b_off, float* packed);
// slab-accumulator gradient kernel for two-hidden-layer nets of 4..7 hidden tiles (cnf_grad_slab.hip)
bool grad_slab_supported(const cnf_config& c);
size_t grad_slab_packed_bytes(const cnf_config& c);
void grad_slab_pack(const cnf_config& c, const float* lux, const size_t* w_off, const size_t* b_off, float* packed);
size_t grad_slab_ws_floats(const cnf_config& c, int alg, int nsteps, long long B, int num_cus);
hipError_t grad_slab_launch(const cnf_config& c, const float* packed_dev, const float* x, const float* eps, const float* ys,
                            const size_t* w_off, const size_t* b_off, int alg, int nsteps, float t0, float t1,
                            const float* tgrid_dev, long long B, const float lam[3], float* ws, float* grad, float* grad_x, int num_cus, hipStream_t st,
                            const float* pre_ckpt = nullptr, const float* pre_ckpt_k = nullptr, int pre_zr = 0);   // checkpoints an adaptive solve wrote (forward instance's layout, pre_zr state k-steps): no forward sweep
int mfma_plan_zr(const MfmaPlan* p);   // state k-steps of the forward instance (checkpoint stride)
bool mfma_plan_is_per_wave(const MfmaPlan* p);
int mfma_plan_family_for(MfmaPlan* p, long long B, bool whole_solve);   // CNF_FAMILY_*
bool mfma_plan_can_checkpoint(const MfmaPlan* p, bool on_grid);
bool mfma_plan_coop_shape(const MfmaPlan* p, int* HT, int* L, int* ZR, int* ACT);   // true for a cooperative-kernel plan
bool mfma_plan_coop_grad_shape(const MfmaPlan* p, int* HT, int* L, int* ZR, int* ACT, int* CR = nullptr);   // ... or an extended one that can checkpoint
long long mfma_plan_ckpt_tiles(const MfmaPlan* p, long long B, bool on_grid = false);
// cooperative gradient for wide layers (cnf_coop_grad.hip + the deferred weight-cotangent products of cnf_lgemm.hip; host side in
// cnf_layered.hip): loss terms from the checkpointing forward solve, gradient in the Lux layout, dL/dx
bool coop_grad_eligible(const cnf_config& c, const MfmaPlan* plan, const float lam[3], bool on_grid);
int coop_grad_stage_store_tiles(const cnf_config& c, MfmaPlan* plan, long long B, int alg, int nsteps, bool on_grid);   // > 0: the cooperative gradient's second form
long long coop_grad_max_columns(const cnf_config& c, int alg);   // batches beyond it take the layer-wise path (32-bit operand addressing)
bool mfma_plan_ckpt_rows_as_tiles(const MfmaPlan* p, long long B, bool on_grid);   // cnf_mfma.hip: the checkpointing solve of B columns runs on the dealt kernel's 64-sample form, which can write KArgs::ck_tiles
int mfma_plan_stage_store_tiles(const MfmaPlan* p, long long B, bool on_grid);   // cnf_mfma.hip: 0 = the forward solve writes no stage store
hipError_t coop_grad(LayeredGrad** ctx, const cnf_config& c, MfmaPlan* plan, const float* packed_dev, const size_t* w_off,
                     const size_t* b_off, const float* x, const float* eps, const float* ys, int alg, int nsteps, float t0, float t1,
                     const float* tgrid, const float* tgrid_dev, long long B, const float lam[3], float* grad, float* grad_x, float* logp_out, float* regs_out, hipStream_t st, std::string* err);

// ---- variable-coefficient Adams PECE (cnf_vcabm.hip): elementwise passes of one step attempt ----
constexpr int kVcSlots = 13;   // Phi*_0 .. Phi*_12: orders 1..12 plus the difference the order-raising estimate needs
struct VcCoef {
    const float* ps_old;   // Phi*_j(n-1): kVcSlots vectors, stride ld
    float* ps_new;         // Phi*_j(n)
    size_t ld;
    int k, m;              // order (predictor terms); differences the history supports (m <= k + 1)
    float dt;
    float beta[kVcSlots];
    float g[kVcSlots + 1];
    float e0, e1, e2;      // dt (g_k - g_{k-1}), dt (g_{k-1} - g_{k-2}), dt (g_{k-2} - g_{k-3});  errup: e0 = dt gamma*_{k+1}
};
size_t vcabm_partial_doubles();
hipError_t vcabm_scaled_sumsq(const float* a, const float* b, const float* u, float abstol, float reltol, int64_t n,
                              double* partial, double* out1, hipStream_t st);
hipError_t vcabm_predict(const float* f, const float* u, const VcCoef& c, int64_t n, float* p, hipStream_t st);
hipError_t vcabm_correct(const float* d, const float* p, const float* u, const VcCoef& c, float abstol, float reltol,
                         int64_t n, float* unew, double* partial, double* err3, hipStream_t st);
hipError_t vcabm_errup(const float* fnew, const float* u, const float* unew, const VcCoef& c, float abstol, float reltol,
                       int64_t n, double* partial, double* err1, hipStream_t st);

}  // namespace cnf
