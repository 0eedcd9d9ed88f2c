// cnf_handle.h — private to the C-ABI layer (cnf_api.hip: handles, parameters, fixed-step entry points;
// cnf_api_adaptive.hip: caller-driven and whole adaptive solves; cnf_api_grad.hip: the gradient entry points).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "cnf_internal.h"

namespace cnf {

// records the message cnf_last_error() returns (thread-local) and hands the status code back
int api_fail(int code, const std::string& msg);

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ::cnf::api_fail(CNF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Device-side repacking.  Every element of an operand image is either zero padding or ONE Lux parameter
// times a constant (1, or the tanh pre-scale folded into the forward images), so an image is a gather:
// packed[j] = p[idx[j]] * scale[j].  The map is derived from the host packer itself (pack a vector of
// ones -> scale, pack the ramp 1, 2, 3, ... -> idx) and verified bit-for-bit against it on a random vector; an
// image that is not a gather (the split-bf16 hidden images) fails the check and keeps the host path.
// With a map, cnf_set_params on a device pointer is one kernel on the caller's stream: no host round
// trip and no synchronisation in a training loop that updates ps on the device every step.
struct PackMap {
    int* idx = nullptr;       // device, source parameter or -1 (zero padding)
    float* scale = nullptr;   // device
    size_t n = 0;
    bool valid = false;
};

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace cnf

// The handle is one configuration + its parameters + the state each kernel family keeps beside them, grouped by who owns it
// (the flat struct had grown to ~60 fields across six families).
namespace cnf {

// parameters as the host handed them over and as the kernels read them
struct HandleParams {
    size_t n = 0;
    bool have = false;
    float* P_dev = nullptr;              // Lux layout (SIMT path)
    float* packed_dev = nullptr;         // MFMA operand image of the forward plan
    std::vector<size_t> w_off, b_off;    // Lux offsets given to cnf_set_params
    // device-side repacking (see PackMap): the map of the solve image, rebuilt when the layout handed to cnf_set_params changes;
    // `stage` holds host-supplied parameters
    PackMap map_fwd;
    bool maps_built = false;
    bool repack_on_device = false;
    float* stage = nullptr;
    size_t stage_n = 0;
};
// thread-per-sample family: workspaces grown on demand
struct HandleSimt {
    float* ws = nullptr;
    int64_t ws_B = 0;
    float* kbuf = nullptr;               // 6 stage derivatives + 1 state, each S x kbuf_B
    int64_t kbuf_B = 0;
};
// parameter gradient (cnf_loss_grad_*): operand images, checkpoints, the auxiliary cooperative plan
struct HandleGrad {
    float* packed = nullptr;             // plain f32 operand image for the register-accumulator reverse sweep
    float* ws = nullptr;                 // checkpoints + logp + regs
    size_t ws_bytes = 0;
    PackMap map, map_slab;
    float* slab_packed = nullptr;        // operand image of the slab-accumulator gradient kernel (cnf_grad_slab.hip)
    float* slab_ws = nullptr;            // its checkpoints + slabs
    size_t slab_ws_floats = 0;
    // Two-hidden-layer nets of 7 .. 8 hidden tiles keep their per-wave forward plan (the one-launch adaptive solvers hang off it),
    // but at large batches their gradient is faster on the cooperative reverse sweep: a second, cooperative plan + image for it
    MfmaPlan* plan_cg = nullptr;
    float* cg_packed = nullptr;
    PackMap map_cg;
    bool cg_tried = false;
    LayeredGrad* layered = nullptr;      // operand images + workspaces of the layer-wise evaluation / gradient and of the cooperative gradient
    float* tgrid_dev = nullptr;          // step times of a non-uniform grid for the fused gradient kernels
    size_t tgrid_cap = 0;
    float* probe_ws = nullptr;           // several probes served probe by probe through the one-probe twin: one probe's columns, its
    size_t probe_ws_floats = 0;          // gradient, data gradient and loss sums (cnf_api_grad.hip::loss_grad_probe_loop)
};
// embedded-step workspace (cnf_step_embedded): 7 stage derivatives + 1 stage state, each S x B
struct HandleEmbedded {
    float* buf = nullptr;
    int64_t B = 0;
    int k[7] = {0, 1, 2, 3, 4, 5, 6};    // which slot holds k_1 .. k_7 (first-same-as-last swaps slots 0 and 6)
    double* err_partial = nullptr;
};
// multistep solve (cnf_vcabm_*): 6 state-size vectors + 2 x kVcSlots difference vectors, each S x B
struct HandleVcabm {
    float* buf = nullptr;
    double* partial = nullptr;
    double* host_res = nullptr;          // 8 doubles of pinned host memory: the reduction kernels of the library's own policy loops write their
                                         // sums there, so the loop synchronises and reads instead of copying (a small copy costs 25 us)
    int64_t B = -1, cap = 0;             // columns of the solve in progress; columns the allocation holds
    int iu = 0, iun = 2, ifn0 = 3, ifn1 = 5, cur = 0;   // which vector holds u, u_new, f_n, f_{n+1}; live difference half
    int nhist = 0, k = 0;                // accepted steps since begin; order of the pending attempt (0 = none)
    int avail = 0, m = 0;                // differences Phi*_j(n-1) the last accepted step stored; those the pending attempt stores
    double hist[kVcSlots + 1] = {};      // signed sizes of the accepted steps, newest first
    double t = 0.0, dt = 0.0;
};
// adaptive whole solves (cnf_solve_controller, cnf_solve_tsit5)
struct HandleAdaptive {
    int last_controller = -1;
    void* dc_buf = nullptr;              // device-controlled adaptive solve: slots, counter, stats, accepted steps
    size_t dc_bytes = 0;
    unsigned dc_epoch = 0;               // launches on dc_buf since it was allocated / last zeroed (mfma.hip::fill_aargs_scratch)
    int* host_rec = nullptr;             // pinned host memory the one-launch solves write their status words and first steps into (AArgs::host_rec)
    float* buf = nullptr;                // adaptive Tsit5 whole solve: two states + two derivative scratch vectors
    int64_t B = 0;
};

}  // namespace cnf

struct cnf_handle {
    cnf_config cfg{};
    int D = 0, S = 0;
    cnf::NetDev net{};
    int path = CNF_PATH_SIMT;
    int num_cus = 0;
    cnf::MfmaPlan* plan = nullptr;       // the fused forward plan (null on the SIMT / layer-wise paths)
    bool layered_forced = false;         // kernel_path = CNF_PATH_LAYERED given explicitly: GEMM path for every batch
    float* loss_partial = nullptr;       // partial sums of the loss reduction (every family)
    cnf::HandleParams par;
    cnf::HandleSimt simt;
    cnf::HandleGrad grad;
    cnf::HandleEmbedded emb;
    cnf::HandleVcabm vc;
    cnf::HandleAdaptive adp;
    // Hutchinson JVP mode without the Jacobian regulariser: eps^T (J eps) and (eps^T J) eps are the same number, so the loss is the
    // VJP mode's loss and its parameter gradient is served by the VJP mode's fused reverse sweeps through this internal handle of
    // the same configuration with mode = CNF_MODE_HUTCH_VJP (the JVP-specific gradient kernels are layer-wise only).
    // Hutchinson VJP mode with K > 1 probes: the loss is the mean over the probes of the one-probe losses (-eps_k^T J eps_k and
    // |eps_k^T J| enter it as means over k, the state does not depend on eps), so where the K-probe configuration itself has no fused
    // gradient and the one-probe configuration runs on the cooperative reverse sweep, the gradient is K calls of this internal
    // one-probe handle, averaged in a fixed order.  Null otherwise.
    cnf_handle* grad_twin = nullptr;
};

namespace cnf {

// shared helpers of the three files
int api_check_call(cnf_handle* h, const float* eps, const float* ys, int64_t B, const char* who);
// f(u + dt sum coef k, t) on whichever family serves the handle; `stage` is scratch for the fused path, whose single-call
// kernel takes the stage state itself
int api_eval_dynamics(cnf_handle* h, const StageIn& in, float t, const float* eps, const float* ys, int64_t B, float* du,
                      float* stage, bool first, hipStream_t st);
// fixed steps on a given grid: u advanced in place, 6 (Tsit5) or 4 (RK4) evaluations per step on the handle's family
int api_integrate_grid(cnf_handle* h, int alg, int nsteps, const float* tgrid, float* u, const float* eps, const float* ys,
                       int64_t B, hipStream_t st);
// cnf_api_grad.hip
cnf_config api_grad_cfg(const cnf_handle* h);
bool api_grad_is_fused(const cnf_handle* h);
bool api_grad_uses_slab(const cnf_handle* h);
bool api_grad_uses_coop_aux(const cnf_handle* h, int64_t B);   // the auxiliary cooperative plan serves this batch size
// which gradient implementation serves a call of B columns with `alg` on uniform steps / on a caller's grid (cnf_grad_path_for)
struct GradRoute {
    int path = 0;              // 0 none, 1 fused per-wave (register or slab accumulators), 2 layer-wise, 3 cooperative reverse sweep
    bool slab = false;         // path 1 on the slab-accumulator kernel
    bool use_cg_aux = false;   // path 3 on the handle's auxiliary cooperative plan (plan_cg / cg_packed)
};
GradRoute api_grad_route(const cnf_handle* h, int64_t B, int alg, bool on_grid);
// cnf_api_adaptive.hip
int api_ensure_adaptive_buf(cnf_handle* h, int64_t B);
// `ck` (may be null): checkpoint arrays the one-launch solve fills for its accepted steps - z_n per step (cap + 1 slots), the six stage
// derivatives per step (6 cap slots), each slot [tile][lane][ZR of the plan] - so that the frozen-grid gradient needs no forward pass
// of its own; ok = the solve ran in one launch and took at most `cap` steps (otherwise the arrays are not to be used)
struct TsitCkpt { float* ckpt; float* ckpt_k; int cap; bool ok; };
int api_solve_tsit5(cnf_handle* h, float t0, float t1, const float* u0, const float* eps, const float* ys, int64_t B,
                    float abstol, float reltol, float dt_init, int maxiters, float* u1, cnf_solve_stats* stats,
                    std::vector<double>* steps, void* stream, TsitCkpt* ck = nullptr);

}  // namespace cnf
