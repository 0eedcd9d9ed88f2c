// cnf_api_grad.hip — the gradient entry points of the C ABI (include/cnf.h): cnf_loss_grad_fixed / _grid / _adaptive and
// cnf_grad_path, over the three implementations (register-accumulator and several-probe kernels, slab-accumulator kernel,
// layer-wise path) - which one serves a handle, their workspaces, the checkpointing forward pass.
#include "cnf_handle.h"

using namespace cnf;

namespace {
int fail(int code, const std::string& msg) { return cnf::api_fail(code, msg); }
}  // namespace

// The configuration the fused gradient kernels are selected and packed for.  TestMode (exact trace): -tr J is the sum over
// the D unit vectors e_k of -e_k^T J e_k, i.e. the several-probe reverse sweep with K = D one-hot probes of weight 1 and no
// regularisers (the probe loop of cnf_grad2_probes.hip has no capacity limit; shapes outside the fused kernels take the
// layer-wise path).
cnf_config cnf::api_grad_cfg(const cnf_handle* h) {
    cnf_config c = h->cfg;
    if (c.mode == CNF_MODE_EXACT) {
        c.mode = CNF_MODE_HUTCH_VJP;
        c.nprobes = h->D;
        c.reg_z = c.reg_j = c.reg_aug = 0;
    }
    return c;
}

__global__ void unit_probes_kernel(float* __restrict__ eps, int D, long long B) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long DD = (long long)D * D;
    if (i >= DD * B) return;
    const int r = (int)(i % DD);
    eps[i] = (r / D) == (r % D) ? 1.f : 0.f;   // probe k = rows k D .. k D + D - 1 of the column: e_k
}

extern "C" {

// fused reverse-sweep kernel, unless CNF_GRAD_LAYERED=1 forces the layer-wise path (tests, A/B timing)
extern "C++" {
bool cnf::api_grad_is_fused(const cnf_handle* h) {
    return h->path == CNF_PATH_MFMA && grad_supported(api_grad_cfg(h)) && mfma_plan_is_per_wave(h->plan) && tuning().grad_layered == 0;
}
}  // extern "C++"

// slab-accumulator kernel for the mid-width two-hidden-layer nets (CNF_GRAD_LAYERED=1 skips it too)
extern "C++" {
// the cooperative reverse sweep beats the slab kernel on the shapes that have both at every batch size: 86 -> 80 ms at 2 x 104,
// 106 -> 84 ms at 2 x 128, B = 65 536 (round 3, which set a threshold of 4096 columns for the sweep's 40 + 160 launches to
// amortise) - and, measured in round 5, 12.4 -> 7.4 ms at nvariables = 12 / 13 and 18.2 -> 8.4 ms at 14 / 15 at B = 1024 (64 ... 4000
// columns alike: the slab kernel is one wave per 16-sample tile, a latency chain; profiles/r5/r5y_mid_width_small_batches.json)
// 5 - 6 hidden tiles (nvariables = 8 ... 11, on the 8-tile instance): the sweep wins up to 8192 columns = two 16-sample super-tiles per
// CU (B = 1024: 9.0 -> 6.4 ms at nvariables = 10; 8192: 11.4 -> 8.3), the slab kernel beyond (10 240: 11.6 against 13.1 ms; 65 536:
// 47.7 against 55.1) - the auxiliary plan serves them up to 8192 columns.
bool cnf::api_grad_uses_coop_aux(const cnf_handle* h, int64_t B) {
    if (!h->grad.plan_cg || !h->grad.cg_packed || B < 1) return false;
    const int sw = tuning().coop_grad_mid;   // > 1: "from sw columns on" (A/B runs, tests of the slab kernel below it)
    if (sw > 1) return B >= sw;
    return h->cfg.widths[1] > 96 || B <= 8192;
}

bool cnf::api_grad_uses_slab(const cnf_handle* h) {
    return (h->grad.slab_packed || !h->par.have) && grad_slab_supported(h->cfg) && tuning().grad_layered == 0;
}
}  // extern "C++"

// The ONE place that decides which implementation serves a gradient call (the query entries and loss_grad_impl both ask it):
// 1 = fused per-wave kernels (register / slab accumulators), 2 = layer-wise, 3 = cooperative reverse sweep, 0 = none.
extern "C++" {
cnf::GradRoute cnf::api_grad_route(const cnf_handle* h, int64_t B, int alg, bool on_grid) {
    GradRoute r{};
    if (api_grad_is_fused(h) && (h->grad.packed || !h->par.have)) { r.path = 1; return r; }
    const bool slab = api_grad_uses_slab(h);
    // B < 0: "the batch is not known" - the auxiliary cooperative plan of a slab shape is not counted (cnf_grad_path)
    const bool fits = B < 0 || B <= coop_grad_max_columns(h->cfg, alg);
    const float lam0[3] = {0.f, 0.f, 0.f};
    if (B >= 0 && api_grad_uses_coop_aux(h, B) && fits && coop_grad_eligible(h->cfg, h->grad.plan_cg, lam0, on_grid)) {
        r.path = 3; r.use_cg_aux = true; return r;
    }
    if (slab) { r.path = 1; r.slab = true; return r; }
    // CNF_LAYERED_LOSS_BY_SOLVE (A/B switch of the layer-wise path: loss from a separate solve) keeps the call layer-wise
    if (fits && !tuning().layered_loss_by_solve && (h->par.packed_dev || !h->par.have) && coop_grad_eligible(h->cfg, h->plan, lam0, on_grid)) {
        r.path = 3; return r;
    }
    r.path = layered_grad_supported(h->cfg) ? 2 : 0;
    return r;
}
}  // extern "C++"

// Who serves a gradient call (cnf_handle::grad_twin): the handle itself; for JVP mode without the Jacobian regulariser its VJP-mode
// twin when that one has a fused implementation (1 or 3) for the call; for several probes without a fused implementation of their
// own the one-probe twin, once per probe (`nloop` = K), when that one runs on the cooperative reverse sweep.
struct GradServe { const cnf_handle* srv; int path; int nloop; };
static GradServe grad_serve(const cnf_handle* h, int64_t B, int alg, bool on_grid) {
    const int own = api_grad_route(h, B, alg, on_grid).path;
    if (h->grad_twin) {
        if (h->cfg.mode == CNF_MODE_HUTCH_JVP) {
            const GradServe t = grad_serve(h->grad_twin, B, alg, on_grid);
            if (t.path == 1 || t.path == 3) return t;
        } else if (own != 1 && own != 3) {
            const GradRoute tr = api_grad_route(h->grad_twin, B, alg, on_grid);
            // measured at K = 4, B = 32 768 (profiles/probes_wide_timing.py, profiles/r6/r6z_probes_wide_timing.json): 1.76 - 1.78 x the
            // layer-wise path on two hidden layers (the reference's default architecture); on 3 x 256 0.96 x with the recomputing sweeps
            // of round 5 and 1.22 x with the cooperative gradient's second form (345 against 420 ms) - so three hidden layers take the
            // loop where the twin's call takes that form (asked with one step: a store that does not fit HBM falls back to the older
            // sweeps inside the loop), else keep their layer-wise gradient unless CNF_PROBE_GRAD_TWIN=2 asks for the loop
            bool loop = tr.path == 3 && (h->cfg.n_layers == 3 || tuning().probe_grad_twin == 2);
            if (tr.path == 3 && !loop && tuning().probe_grad_twin == 1 && B > 0) {
                const cnf_handle* t = h->grad_twin;
                loop = coop_grad_stage_store_tiles(t->cfg, tr.use_cg_aux ? t->grad.plan_cg : t->plan, B, alg, 1, on_grid) > 0;
            }
            if (loop) return GradServe{h->grad_twin, 3, h->cfg.nprobes};
        }
    }
    return GradServe{h, own, 1};
}

int cnf_grad_path(const cnf_handle* h) {
    if (!h) return CNF_ERR_INVALID;
    return grad_serve(h, -1, CNF_ALG_TSIT5, false).path;
}

int cnf_grad_path_for(const cnf_handle* h, int64_t B, int alg, int on_grid) {
    if (!h || B < 0 || (alg != CNF_ALG_RK4 && alg != CNF_ALG_TSIT5)) return CNF_ERR_INVALID;
    return grad_serve(h, B, alg, on_grid != 0).path;
}

int cnf_grad_form_for(const cnf_handle* h, int64_t B, int alg, int nsteps, int on_grid) {
    if (!h || B < 0 || nsteps < 1 || (alg != CNF_ALG_RK4 && alg != CNF_ALG_TSIT5)) return CNF_ERR_INVALID;
    const GradServe s = grad_serve(h, B, alg, on_grid != 0);
    if (s.path != 3) return 0;
    const GradRoute r = api_grad_route(s.srv, B, alg, on_grid != 0);
    MfmaPlan* plan = r.use_cg_aux ? s.srv->grad.plan_cg : s.srv->plan;
    return coop_grad_stage_store_tiles(s.srv->cfg, plan, B, alg, nsteps, on_grid != 0) > 0 ? 2 : 1;
}

}  // extern "C"

// Layout of the fused per-wave gradient's workspace (cnf_handle::grad.ws) for `steps` steps: z checkpoints (steps + 1 slots), stage
// derivatives (steps x stages slots), logp + regs (4 B), the gradient slabs, the ping-pong states of a grid's step-by-step forward
// pass, the unit probes of TestMode.
struct FusedWs { size_t ckpt_z_floats, ckpt_k_floats, slab_floats, state_floats, unit_floats, need; };
static FusedWs fused_ws(cnf_handle* h, int alg, int steps, int64_t B, bool on_grid) {
    FusedWs W{};
    const size_t ntiles = (size_t)((B + 15) / 16);
    const size_t zslot = ntiles * 64 * (size_t)mfma_plan_zr(h->plan);
    const int nstages = alg == CNF_ALG_RK4 ? 4 : 6;
    W.ckpt_z_floats = (size_t)(steps + 1) * zslot;
    W.ckpt_k_floats = (size_t)steps * nstages * zslot;
    W.slab_floats = grad_slab_floats(api_grad_cfg(h), h->num_cus);
    W.state_floats = on_grid ? 2 * (size_t)h->S * (size_t)B : 0;
    W.unit_floats = h->cfg.mode == CNF_MODE_EXACT ? (size_t)h->D * (size_t)h->D * (size_t)B : 0;
    W.need = (W.ckpt_z_floats + W.ckpt_k_floats + 4 * (size_t)B + W.slab_floats + W.state_floats + W.unit_floats) * sizeof(float);
    return W;
}

// Checkpoints the adaptive solve that found the grid has already written (api_solve_tsit5's TsitCkpt): arrays laid out for `cap`
// steps at the head of the handle's gradient workspace (fused_ws below), and the solve's final state for the loss terms.
// (ckpt / ckpt_k / zr: where the slab-accumulator kernel finds them - behind the loss workspace - and their stride; the fused per-wave
// path derives its own from fused_ws)
struct PreparedCkpt { int cap; const float* u_final; const float* ckpt; const float* ckpt_k; int zr; };
static int loss_grad_impl(cnf_handle* h, const char* who, int alg, int nsteps, float t0, float t1, const float* tgrid,
                          const float* x, const float* eps, const float* ys, int64_t B, const float* lambdas,
                          float* grad, float* grad_x, float* sums4, void* stream, const PreparedCkpt* pc = nullptr);

// rows p D .. p D + D - 1 of every column of the (K D) x B probe array: probe p as a D x B array
__global__ void probe_slice_kernel(const float* __restrict__ eps, int K, int D, int p, long long B, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)D * B) return;
    const long long b = i / D;
    out[i] = eps[b * (long long)K * D + (long long)p * D + (i - b * D)];
}

// acc = first ? w v : acc + w v
__global__ void probe_accum_kernel(float* __restrict__ acc, const float* __restrict__ v, float w, int first, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = first ? w * v[i] : fmaf(w, v[i], acc[i]);
}

// Several probes through the one-probe handle `one` (cnf_handle::grad_twin): loss sums and gradients of the K one-probe calls,
// averaged in probe order.  Every call is a whole forward solve + reverse sweep of the cooperative path.
static int loss_grad_probe_loop(cnf_handle* h, cnf_handle* one, int K, const char* who, int alg, int nsteps, float t0, float t1,
                                const float* tgrid, const float* x, const float* eps, const float* ys, int64_t B,
                                const float* lambdas, float* grad, float* grad_x, float* sums4, void* stream) {
    if ((B > 0 && (!x || !eps)) || !grad || !lambdas) return fail(CNF_ERR_INVALID, std::string(who) + ": null x/eps/grad/lambdas");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    const size_t D = (size_t)h->D, n = h->par.n, nx = grad_x ? (size_t)B * (size_t)h->cfg.nvars : 0;
    const size_t need = D * (size_t)B + n + nx + 4 + 16;
    if (need > h->grad.probe_ws_floats) {
        if (h->grad.probe_ws) HIP_TRY(hipFree(h->grad.probe_ws));
        h->grad.probe_ws = nullptr; h->grad.probe_ws_floats = 0;
        HIP_TRY(hipMalloc((void**)&h->grad.probe_ws, need * sizeof(float)));
        h->grad.probe_ws_floats = need;
    }
    float* eps_p = h->grad.probe_ws;
    float* grad_p = eps_p + (D * (size_t)B + 3) / 4 * 4;
    float* gx_p = grad_x ? grad_p + (n + 3) / 4 * 4 : nullptr;
    float* sums_p = grad_p + (n + 3) / 4 * 4 + (nx + 3) / 4 * 4;
    const float w = 1.f / (float)K;
    auto accum = [&](float* acc, const float* v, size_t cnt, int first) -> hipError_t {
        if (!cnt) return hipSuccess;
        hipLaunchKernelGGL(probe_accum_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, acc, v, w, first, (long long)cnt);
        return hipGetLastError();
    };
    if (B == 0) return loss_grad_impl(one, who, alg, nsteps, t0, t1, tgrid, x, eps, ys, B, lambdas, grad, grad_x, sums4, stream);
    for (int p = 0; p < K; ++p) {
        const long long cnt = (long long)D * B;
        hipLaunchKernelGGL(probe_slice_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, eps, K, (int)D, p, (long long)B, eps_p);
        HIP_TRY(hipGetLastError());
        const int rc = loss_grad_impl(one, who, alg, nsteps, t0, t1, tgrid, x, eps_p, ys, B, lambdas, grad_p, gx_p, sums4 ? sums_p : nullptr, stream);
        if (rc) return rc;
        HIP_TRY(accum(grad, grad_p, n, p == 0));
        if (grad_x) HIP_TRY(accum(grad_x, gx_p, nx, p == 0));
        if (sums4) HIP_TRY(accum(sums4, sums_p, 4, p == 0));
    }
    return CNF_OK;
}

// loss sums + gradient on a uniform grid (tgrid == nullptr: nsteps steps from t0 to t1) or on the caller's non-uniform
// grid (tgrid: host, nsteps + 1 times; t0 / t1 ignored).  The same three gradient implementations serve both.
static int loss_grad_impl(cnf_handle* h, const char* who, int alg, int nsteps, float t0, float t1, const float* tgrid,
                          const float* x, const float* eps, const float* ys, int64_t B, const float* lambdas,
                          float* grad, float* grad_x, float* sums4, void* stream, const PreparedCkpt* pc) {
    int rc = api_check_call(h, eps, ys, B, who);
    if (rc) return rc;
    if (alg == CNF_ALG_RK4 || alg == CNF_ALG_TSIT5) {
        const GradServe gs = grad_serve(h, B, alg, tgrid != nullptr);
        cnf_handle* srv = const_cast<cnf_handle*>(gs.srv);
        if (gs.nloop > 1) return loss_grad_probe_loop(h, srv, gs.nloop, who, alg, nsteps, t0, t1, tgrid, x, eps, ys, B, lambdas, grad, grad_x, sums4, stream);
        if (srv != h) return loss_grad_impl(srv, who, alg, nsteps, t0, t1, tgrid, x, eps, ys, B, lambdas, grad, grad_x, sums4, stream);
    }
    const std::string w(who);
    if (nsteps < 1) return fail(CNF_ERR_INVALID, w + ": nsteps >= 1 required");
    if (alg != CNF_ALG_RK4 && alg != CNF_ALG_TSIT5) return fail(CNF_ERR_INVALID, w + ": unknown alg");
    if ((B > 0 && !x) || !grad || !lambdas) return fail(CNF_ERR_INVALID, w + ": null x/grad/lambdas");
    const GradRoute route = api_grad_route(h, B, alg, tgrid != nullptr);
    const bool fused = route.path == 1 && !route.slab;
    if (route.path == 0) return fail(CNF_ERR_UNSUPPORTED, w + ": no gradient path for this configuration");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(zero_async(grad, h->par.n * sizeof(float), st));
    if (B == 0) {
        if (sums4) HIP_TRY(zero_async(sums4, 4 * sizeof(float), st));
        return CNF_OK;
    }
    const float* tgrid_dev = nullptr;
    if (tgrid) {   // the fused kernels read the step times from device memory (uniform loads, once per step)
        t0 = tgrid[0]; t1 = tgrid[nsteps];
        if ((size_t)nsteps + 1 > h->grad.tgrid_cap) {
            if (h->grad.tgrid_dev) HIP_TRY(hipFree(h->grad.tgrid_dev));
            h->grad.tgrid_dev = nullptr; h->grad.tgrid_cap = 0;
            const size_t cap = ((size_t)nsteps + 1 + 63) / 64 * 64;
            HIP_TRY(hipMalloc((void**)&h->grad.tgrid_dev, cap * sizeof(float)));
            h->grad.tgrid_cap = cap;
        }
        HIP_TRY(hipMemcpyAsync(h->grad.tgrid_dev, tgrid, ((size_t)nsteps + 1) * sizeof(float), hipMemcpyHostToDevice, st));
        tgrid_dev = h->grad.tgrid_dev;
    }
    if (!fused) {
        // the loss sums come from the regular solve on whichever family serves the handle
        // (the cooperative sweep's checkpointing forward solve always yields the loss terms; the layer-wise sweep accumulates
        // them unless CNF_LAYERED_LOSS_BY_SOLVE asks for a separate solve; the slab kernel needs that solve)
        const bool use_cg = route.use_cg_aux;
        const bool loss_in_sweep = sums4 && (route.path == 3 || (route.path == 2 && !tuning().layered_loss_by_solve));
        // slab-accumulator kernel on uniform steps: ONE forward solve serves both the loss terms and - through its checkpoints, in the
        // forward instance's layout, kept behind the loss workspace - the kernel, which then runs no forward sweep of its own
        const bool slab_shared = route.slab && !tgrid && sums4 && tuning().adaptive_ckpt != 0 && h->path == CNF_PATH_MFMA && h->plan &&
                                 mfma_plan_is_per_wave(h->plan) && h->par.packed_dev;
        const size_t sh_zslot = slab_shared ? (size_t)((B + 15) / 16) * 64 * (size_t)mfma_plan_zr(h->plan) : 0;
        const int sh_stages = alg == CNF_ALG_RK4 ? 4 : 6;
        const size_t sh_head = ((size_t)h->S + 4) * (size_t)B;
        const float *sh_ckpt = nullptr, *sh_ckpt_k = nullptr;
        if (sums4) {
            const size_t need = (sh_head + (slab_shared ? (size_t)((sh_stages + 1) * nsteps + 1) * sh_zslot : 0)) * sizeof(float);
            if (need > h->grad.ws_bytes) {
                if (h->grad.ws) HIP_TRY(hipFree(h->grad.ws));
                h->grad.ws = nullptr; h->grad.ws_bytes = 0;
                HIP_TRY(hipMalloc((void**)&h->grad.ws, need));
                h->grad.ws_bytes = need;
            }
            float* logp = h->grad.ws;
            float* regs = logp + B;
            if (loss_in_sweep) {
                // the layer-wise reverse sweep accumulates the loss terms of the solve it differentiates (below)
            } else if (tgrid && pc && pc->u_final) {
                // the adaptive solve that found the grid has the state at t1: its loss terms, no second solve over the grid
                const int ra0 = (h->cfg.mode != CNF_MODE_EXACT && h->cfg.reg_aug && h->cfg.naug > 0) ? 1 : 0;
                HIP_TRY(epilogue(pc->u_final, h->cfg.nvars, h->D, ra0, B, logp, regs, st));
            } else if (tgrid) {   // the loss of the same discrete solve: augmented state advanced over the grid, then the epilogue
                float* u = regs + 3 * (size_t)B;
                const int ra0 = (h->cfg.mode != CNF_MODE_EXACT && h->cfg.reg_aug && h->cfg.naug > 0) ? 1 : 0;
                HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, u, st));
                rc = api_integrate_grid(h, alg, nsteps, tgrid, u, eps, ys, B, st);
                if (rc) return rc;
                HIP_TRY(epilogue(u, h->cfg.nvars, h->D, ra0, B, logp, regs, st));
            } else if (slab_shared) {
                float* ck = h->grad.ws + sh_head;
                float* ckk = ck + (size_t)(nsteps + 1) * sh_zslot;
                SolveArgs a{};
                a.x = x; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = nsteps; a.alg = alg; a.t0 = t0; a.t1 = t1;
                a.logp = logp; a.regs = regs; a.nvars = h->cfg.nvars;
                a.reg_aug = (h->cfg.mode != CNF_MODE_EXACT && h->cfg.reg_aug && h->cfg.naug > 0) ? 1 : 0;
                a.ckpt = ck; a.ckpt_k = ckk;
                HIP_TRY(mfma_solve(h->plan, h->par.packed_dev, a, st));
                sh_ckpt = ck; sh_ckpt_k = ckk;
            } else {
                rc = cnf_inference_fixed(h, alg, nsteps, t0, t1, x, eps, ys, B, logp, regs, nullptr, stream);
                if (rc) return rc;
            }
            if (!h->loss_partial) HIP_TRY(hipMalloc((void**)&h->loss_partial, 256 * 4 * sizeof(float)));
            if (!loss_in_sweep) HIP_TRY(loss_sums(logp, regs, B, h->loss_partial, sums4, st));
        }
        const bool hutch = h->cfg.mode != CNF_MODE_EXACT;   // the exact-trace dynamics carry no regularisers (icnf.jl:297-339)
        const int ra = (hutch && h->cfg.reg_aug && h->cfg.naug > 0) ? 1 : 0;
        const float lam[3] = {hutch && h->cfg.reg_z ? lambdas[0] : 0.f, hutch && h->cfg.reg_j ? lambdas[1] : 0.f, ra ? lambdas[2] : 0.f};
        if (route.slab) {
            // two-hidden-layer nets of 4..7 hidden tiles: tile-fused reverse sweep with slab accumulators (cnf_grad_slab.hip)
            if (h->num_cus == 0) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, h->cfg.device_id));
                h->num_cus = prop.multiProcessorCount;
            }
            const size_t need = grad_slab_ws_floats(h->cfg, alg, nsteps, B, h->num_cus);
            if (need > h->grad.slab_ws_floats) {
                if (h->grad.slab_ws) HIP_TRY(hipFree(h->grad.slab_ws));
                h->grad.slab_ws = nullptr; h->grad.slab_ws_floats = 0;
                HIP_TRY(hipMalloc((void**)&h->grad.slab_ws, need * sizeof(float)));
                h->grad.slab_ws_floats = need;
            }
            const bool pre = pc && pc->cap > 0 && pc->ckpt && pc->ckpt_k && tgrid;
            const float* pk = pre ? pc->ckpt : sh_ckpt;
            const float* pkk = pre ? pc->ckpt_k : sh_ckpt_k;
            const int pzr = pre ? pc->zr : (sh_ckpt ? mfma_plan_zr(h->plan) : 0);
            HIP_TRY(grad_slab_launch(h->cfg, h->grad.slab_packed, x, eps, ys, h->par.w_off.data(), h->par.b_off.data(), alg, nsteps, t0, t1, tgrid_dev, B, lam,
                                     h->grad.slab_ws, grad, grad_x, h->num_cus, st, pk, pkk, pzr));
            return CNF_OK;
        }
        std::string msg;
        MfmaPlan* cgp = use_cg ? h->grad.plan_cg : h->plan;
        const float* cgi = use_cg ? h->grad.cg_packed : h->par.packed_dev;
        if (route.path == 3) {
            if (!cgi) return fail(CNF_ERR_NO_PARAMS, w + ": cnf_set_params has not been called");
            // wide hidden layers on the cooperative kernels: checkpointing forward solve (which also yields the loss terms),
            // one reverse-sweep launch per step, deferred weight-cotangent products (cnf_coop_grad.hip)
            float* cg_logp = sums4 ? h->grad.ws : nullptr;
            hipError_t e = coop_grad(&h->grad.layered, h->cfg, cgp, cgi, h->par.w_off.data(), h->par.b_off.data(), x, eps, ys, alg, nsteps,
                                     t0, t1, tgrid, tgrid_dev, B, lam, grad, grad_x, cg_logp, cg_logp ? cg_logp + B : nullptr, st, &msg);
            if (e != hipSuccess) return fail(CNF_ERR_HIP, w + ": " + msg);
            if (sums4) HIP_TRY(loss_sums(cg_logp, cg_logp + B, B, h->loss_partial, sums4, st));
            return CNF_OK;
        }
        float* lg_logp = loss_in_sweep ? h->grad.ws : nullptr;
        hipError_t e = layered_grad(&h->grad.layered, h->cfg, h->par.P_dev, h->par.w_off.data(), h->par.b_off.data(), x, eps, ys, alg, nsteps,
                                    t0, t1, tgrid, B, lam, grad, grad_x, st, &msg, lg_logp, lg_logp ? lg_logp + B : nullptr);
        if (e == hipErrorNotSupported) return fail(CNF_ERR_UNSUPPORTED, w + ": " + msg);
        if (e != hipSuccess) return fail(CNF_ERR_HIP, w + ": " + msg);
        if (loss_in_sweep) HIP_TRY(loss_sums(lg_logp, lg_logp + B, B, h->loss_partial, sums4, st));
        return CNF_OK;
    }
    if (h->num_cus == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, h->cfg.device_id));
        h->num_cus = prop.multiProcessorCount;
    }
    const long long ntiles = (B + 15) / 16;
    const int ckpt_zr = mfma_plan_zr(h->plan);
    const int nstages = alg == CNF_ALG_RK4 ? 4 : 6;
    const cnf_config gc = api_grad_cfg(h);
    const bool exact = h->cfg.mode == CNF_MODE_EXACT;
    // (checkpoints an adaptive solve has written sit in arrays laid out for pc->cap steps, of which the first nsteps are filled)
    if (pc && pc->cap <= 0) pc = nullptr;   // (a final state without checkpoints serves the other implementations' loss terms only)
    const FusedWs W = fused_ws(h, alg, pc ? pc->cap : nsteps, B, tgrid != nullptr);
    if (W.need > h->grad.ws_bytes) {
        if (pc) return fail(CNF_ERR_INVALID, w + ": prepared checkpoints without their workspace");
        if (h->grad.ws) HIP_TRY(hipFree(h->grad.ws));
        h->grad.ws = nullptr; h->grad.ws_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->grad.ws, W.need));
        h->grad.ws_bytes = W.need;
    }
    const size_t slab_floats = W.slab_floats, state_floats = W.state_floats, unit_floats = W.unit_floats;
    float* ckpt = h->grad.ws;
    float* ckpt_k = ckpt + W.ckpt_z_floats;
    float* logp = ckpt + W.ckpt_z_floats + W.ckpt_k_floats;
    float* regs = logp + B;
    float* slab = regs + 3 * (size_t)B;
    (void)unit_floats;
    const int reg_aug = (!exact && h->cfg.reg_aug && h->cfg.naug > 0) ? 1 : 0;
    if (pc) {
        // the solve that found the grid has left z_n and the stage derivatives of its accepted steps in ckpt / ckpt_k: no forward
        // pass; the loss terms are those of its final state
        HIP_TRY(epilogue(pc->u_final, h->cfg.nvars, h->D, reg_aug, B, logp, regs, st));
    } else if (!tgrid) {
        SolveArgs a{};
        a.x = x; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = nsteps; a.alg = alg; a.t0 = t0; a.t1 = t1;
        a.logp = logp; a.regs = regs; a.nvars = h->cfg.nvars; a.reg_aug = reg_aug; a.ckpt = ckpt; a.ckpt_k = ckpt_k;
        HIP_TRY(mfma_solve(h->plan, h->par.packed_dev, a, st));
    } else {
        // non-uniform grid: the checkpointing forward pass is one launch of the (unchanged) solve kernel per step - the metric
        // kernel keeps its loop-invariant step size; step n writes checkpoint slots n and n + 1 and its stage derivatives
        float* ua = slab + slab_floats;
        float* ub = ua + (size_t)h->S * (size_t)B;
        HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, ua, st));
        const size_t zslot = (size_t)ntiles * 64 * (size_t)ckpt_zr;
        for (int n = 0; n < nsteps; ++n) {
            SolveArgs a{};
            a.u0 = ua; a.u_out = ub; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = 1; a.alg = alg; a.t0 = tgrid[n]; a.t1 = tgrid[n + 1];
            a.nvars = h->cfg.nvars; a.reg_aug = reg_aug;
            a.ckpt = ckpt + (size_t)n * zslot; a.ckpt_k = ckpt_k + (size_t)n * nstages * zslot;
            if (n == nsteps - 1) { a.logp = logp; a.regs = regs; }
            HIP_TRY(mfma_solve(h->plan, h->par.packed_dev, a, st));
            float* tmp = ua; ua = ub; ub = tmp;
        }
    }
    if (sums4) {
        if (!h->loss_partial) HIP_TRY(hipMalloc((void**)&h->loss_partial, 256 * 4 * sizeof(float)));
        HIP_TRY(loss_sums(logp, regs, B, h->loss_partial, sums4, st));
    }
    const float lam[3] = {gc.reg_z ? lambdas[0] : 0.f, gc.reg_j ? lambdas[1] : 0.f, reg_aug ? lambdas[2] : 0.f};
    const float* probes = eps;
    if (exact) {
        float* unit = slab + slab_floats + state_floats;
        const long long nunit = (long long)unit_floats;
        hipLaunchKernelGGL(unit_probes_kernel, dim3((unsigned)((nunit + 255) / 256)), dim3(256), 0, st, unit, h->D, (long long)B);
        HIP_TRY(hipGetLastError());
        probes = unit;
    }
    HIP_TRY(grad_launch(gc, h->grad.packed, ckpt, ckpt_k, ckpt_zr, probes, ys, h->par.w_off.data(), h->par.b_off.data(), alg, nsteps, t0, t1,
                        tgrid_dev, exact ? 1.f : 0.f, B, lam, slab, grad, grad_x, h->num_cus, st));
    return CNF_OK;
}


extern "C" {

int cnf_loss_grad_fixed(cnf_handle* h, int alg, int nsteps, float t0, float t1, const float* x,
                        const float* eps, const float* ys, int64_t B, const float* lambdas,
                        float* grad, float* grad_x, float* sums4, void* stream) {
    return loss_grad_impl(h, "cnf_loss_grad_fixed", alg, nsteps, t0, t1, nullptr, x, eps, ys, B, lambdas, grad, grad_x, sums4, stream);
}

int cnf_loss_grad_grid(cnf_handle* h, int alg, int nsteps, const float* tgrid, const float* x, const float* eps,
                       const float* ys, int64_t B, const float* lambdas, float* grad, float* grad_x, float* sums4,
                       void* stream) {
    if (nsteps < 1 || !tgrid) return fail(CNF_ERR_INVALID, "cnf_loss_grad_grid: nsteps >= 1 and a grid of nsteps + 1 times required");
    return loss_grad_impl(h, "cnf_loss_grad_grid", alg, nsteps, 0.f, 0.f, tgrid, x, eps, ys, B, lambdas, grad, grad_x, sums4, stream);
}

int cnf_loss_grad_adaptive(cnf_handle* h, float t0, float t1, const float* x, const float* eps, const float* ys, int64_t B,
                           float abstol, float reltol, float dt_init, int maxiters, const float* lambdas, float* grad,
                           float* grad_x, float* sums4, cnf_solve_stats* stats, float* tgrid_out, int32_t grid_cap,
                           void* stream) {
    if (stats) *stats = cnf_solve_stats{};
    int rc = api_check_call(h, eps, ys, B, "cnf_loss_grad_adaptive");
    if (rc) return rc;
    if ((B > 0 && !x) || !grad || !lambdas) return fail(CNF_ERR_INVALID, "cnf_loss_grad_adaptive: null x/grad/lambdas");
    if (t0 == t1) return fail(CNF_ERR_INVALID, "cnf_loss_grad_adaptive: empty time span");
    std::vector<float> grid;
    if (B == 0) {   // nothing to step over: the fixed entry zeroes grad / sums4
        return loss_grad_impl(h, "cnf_loss_grad_adaptive", CNF_ALG_TSIT5, 1, t0, t1, nullptr, x, eps, ys, B, lambdas, grad, grad_x, sums4, stream);
    }
    TsitCkpt ck{};
    PreparedCkpt pc{};
    bool slab_route = false;
    cnf_handle* srv = nullptr;   // the handle whose gradient implementation takes the prepared state (h or its twin)
    {
        DeviceGuard g(h->cfg.device_id);
        rc = api_ensure_adaptive_buf(h, B);
        if (rc) return rc;
        const size_t slot = (size_t)h->S * (size_t)h->adp.B;
        float* u = h->adp.buf + 4 * slot;
        // Where the gradient on the frozen grid is the fused per-wave sweep of this very handle, the solve that finds the grid also
        // writes the sweep's checkpoints (z_n and the stage derivatives of every accepted step: the one-launch kernel has them in
        // registers; beyond its capacity the library's host loop lets every fused attempt fill the slots of its step) - the gradient
        // then needs no forward pass of its own.  Up to kAdaptiveCkptSteps steps; a longer solve or CNF_ADAPTIVE_CKPT=0 take the
        // step-by-step forward pass of loss_grad_impl.
        const int kAdaptiveCkptSteps = B <= 32768 ? 32 : 16;   // (7 slots of tiles x 64 x ZR floats a step: 225 MB at 32 768 samples, D <= 8)
        // (gh: the handle whose gradient implementation serves the call - h itself, or its VJP twin for a JVP-mode handle: the solve
        // runs on h either way, and z_n / the stage derivatives do not depend on the trace engine)
        const GradServe gs = grad_serve(h, B, CNF_ALG_TSIT5, true);
        cnf_handle* const gh = const_cast<cnf_handle*>(gs.srv);
        const GradRoute route = api_grad_route(gh, B, CNF_ALG_TSIT5, true);
        const bool eligible = gs.nloop == 1 && route.path == 1 && tuning().adaptive_ckpt != 0 && h->path == CNF_PATH_MFMA && h->plan &&
                              mfma_plan_is_per_wave(h->plan) && gh->path == CNF_PATH_MFMA && gh->plan;
        srv = eligible ? gh : nullptr;
        if (eligible && !route.slab && mfma_plan_zr(h->plan) == mfma_plan_zr(gh->plan)) {
            if (gh->num_cus == 0) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, gh->cfg.device_id));
                gh->num_cus = prop.multiProcessorCount;
            }
            const FusedWs W = fused_ws(gh, CNF_ALG_TSIT5, kAdaptiveCkptSteps, B, true);
            if (W.need > gh->grad.ws_bytes) {
                if (gh->grad.ws) HIP_TRY(hipFree(gh->grad.ws));
                gh->grad.ws = nullptr; gh->grad.ws_bytes = 0;
                HIP_TRY(hipMalloc((void**)&gh->grad.ws, W.need));
                gh->grad.ws_bytes = W.need;
            }
            ck.ckpt = gh->grad.ws; ck.ckpt_k = gh->grad.ws + W.ckpt_z_floats; ck.cap = kAdaptiveCkptSteps;
        } else if (eligible && route.slab) {
            // slab-accumulator gradient (its forward sweep is inside the kernel): the arrays sit behind the loss workspace of the
            // non-fused branch of loss_grad_impl, in the forward instance's layout, which the kernel reads with that stride
            const size_t zslot = (size_t)((B + 15) / 16) * 64 * (size_t)mfma_plan_zr(h->plan);
            const size_t head = ((size_t)gh->S + 4) * (size_t)B;
            const size_t need = (head + (size_t)(7 * kAdaptiveCkptSteps + 1) * zslot) * sizeof(float);
            if (need > gh->grad.ws_bytes) {
                if (gh->grad.ws) HIP_TRY(hipFree(gh->grad.ws));
                gh->grad.ws = nullptr; gh->grad.ws_bytes = 0;
                HIP_TRY(hipMalloc((void**)&gh->grad.ws, need));
                gh->grad.ws_bytes = need;
            }
            ck.ckpt = gh->grad.ws + head; ck.ckpt_k = ck.ckpt + (size_t)(kAdaptiveCkptSteps + 1) * zslot; ck.cap = kAdaptiveCkptSteps;
            slab_route = true;
        }
        HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, u, (hipStream_t)stream));
        std::vector<double> steps;
        rc = api_solve_tsit5(h, t0, t1, u, eps, ys, B, abstol, reltol, dt_init, maxiters, u + slot, stats, &steps, stream, &ck);
        if (rc) return rc;
        double t = t0;
        grid.push_back(t0);
        for (double d : steps) { t += d; grid.push_back((float)t); }
        grid.back() = t1;
        if (ck.ok && (int)steps.size() <= ck.cap) {
            pc.cap = ck.cap;
            if (slab_route) { pc.ckpt = ck.ckpt; pc.ckpt_k = ck.ckpt_k; pc.zr = mfma_plan_zr(h->plan); }
        }
        if (tuning().adaptive_ckpt != 0 && gs.nloop == 1) pc.u_final = u + slot;   // the state at t1: the loss terms of every implementation
        if (!pc.u_final || gs.nloop != 1) srv = nullptr;
        else if (!srv) srv = gh;   // (a route without checkpoints still takes the final state for its loss terms - on the handle that serves it)
    }
    if (tgrid_out)
        for (size_t i = 0; i < grid.size() && (int64_t)i < grid_cap; ++i) tgrid_out[i] = grid[i];
    // (with prepared state / checkpoints the serving handle is called directly: loss_grad_impl's own delegation carries none)
    return loss_grad_impl(srv ? srv : h, "cnf_loss_grad_adaptive", CNF_ALG_TSIT5, (int)grid.size() - 1, 0.f, 0.f, grid.data(), x, eps, ys, B,
                          lambdas, grad, grad_x, sums4, stream, (srv && pc.u_final) ? &pc : nullptr);
}

}  // extern "C"
