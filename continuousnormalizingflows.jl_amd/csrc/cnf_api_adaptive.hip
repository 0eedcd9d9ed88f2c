// cnf_api_adaptive.hip — the adaptive solves of the C ABI (include/cnf.h): cnf_step_embedded (one Tsit5 attempt),
// cnf_vcabm_begin / _attempt / _accept / _state (the passes of the reference's default solver), and the whole solves
// cnf_solve_vcabm / cnf_solve_tsit5 with the solvers' step-size (and order) policies restated on the host side.
#include <cstring>

#include "cnf_handle.h"

using namespace cnf;

namespace {
int fail(int code, const std::string& msg) { return cnf::api_fail(code, msg); }
}  // namespace

extern "C" {

// ckpt / ckpt_k (may be null; per-wave plans only): slots of the frozen-grid gradient's checkpoint arrays for this step - the fused
// attempt then also writes the step's start state, its six stage derivatives and the state it arrives at (api_solve_tsit5's TsitCkpt)
static int step_embedded_impl(cnf_handle* h, int alg, int flags, float t, float dt, const float* u, const float* eps,
                              const float* ys, int64_t B, float abstol, float reltol, float* u_new, double* err_sumsq,
                              void* stream, float* ckpt, float* ckpt_k);

int cnf_step_embedded(cnf_handle* h, int alg, int flags, float t, float dt, const float* u, const float* eps,
                      const float* ys, int64_t B, float abstol, float reltol, float* u_new, double* err_sumsq,
                      void* stream) {
    return step_embedded_impl(h, alg, flags, t, dt, u, eps, ys, B, abstol, reltol, u_new, err_sumsq, stream, nullptr, nullptr);
}

}  // extern "C"

static int step_embedded_impl(cnf_handle* h, int alg, int flags, float t, float dt, const float* u, const float* eps,
                              const float* ys, int64_t B, float abstol, float reltol, float* u_new, double* err_sumsq,
                              void* stream, float* ckpt, float* ckpt_k) {
    int rc = api_check_call(h, eps, ys, B, "cnf_step_embedded");
    if (rc) return rc;
    if (alg != CNF_ALG_TSIT5) return fail(CNF_ERR_INVALID, "cnf_step_embedded: the embedded pair is Tsit5 (alg = CNF_ALG_TSIT5)");
    if (!(abstol >= 0.f) || !(reltol >= 0.f) || (abstol == 0.f && reltol == 0.f))
        return fail(CNF_ERR_INVALID, "cnf_step_embedded: tolerances must be non-negative and not both zero");
    if (B == 0) return CNF_OK;
    if (!u || !u_new || !err_sumsq) return fail(CNF_ERR_INVALID, "cnf_step_embedded: null u/u_new/err_sumsq");
    if (u == u_new) return fail(CNF_ERR_INVALID, "cnf_step_embedded: u_new may not alias u");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)h->S * (size_t)B;
    if (B > h->emb.B) {
        if (h->emb.buf) HIP_TRY(hipFree(h->emb.buf));
        h->emb.buf = nullptr; h->emb.B = 0;
        HIP_TRY(hipMalloc((void**)&h->emb.buf, 8 * n * sizeof(float)));
        h->emb.B = B;
        flags = 0;   // the cached stages went with the old buffer
    }
    if (!h->emb.err_partial) HIP_TRY(hipMalloc((void**)&h->emb.err_partial, kErrBlocks * sizeof(double)));
    const size_t slot = (size_t)h->S * (size_t)h->emb.B;
    float* stage = h->emb.buf + 7 * slot;
    if (flags & CNF_STEP_FSAL) { const int tmp = h->emb.k[0]; h->emb.k[0] = h->emb.k[6]; h->emb.k[6] = tmp; }
    float* k[7];
    for (int i = 0; i < 7; ++i) k[i] = h->emb.buf + (size_t)h->emb.k[i] * slot;
    const Tableau T = make_tableau(CNF_ALG_TSIT5);
    // b - bhat of the embedded 4th-order solution (Tsitouras 2011); satisfies the order-4 conditions to 1e-15
    static const float btilde[7] = {-0.00178001105222577714f, -0.0008164344596567469f, 0.007880878010261995f,
                                    -0.1447110071732629f, 0.5823571654525552f, -0.45808210592918697f,
                                    0.015151515151515152f};
    if (h->path == CNF_PATH_MFMA && mfma_plan_is_per_wave(h->plan)) {
        // fused attempt: the six stages and the update in ONE launch of the solve kernel (nsteps = 1), which also
        // writes every stage derivative; then the 7th stage at u_new and the error reduction.  (The fused step
        // evaluates its own first stage, so the FSAL / RETRY hints save nothing here; 3 launches instead of 14.)
        for (int i = 0; i < 7; ++i) k[i] = h->emb.buf + (size_t)i * n;   // [stage][B][S], packed for this B
        SolveArgs a{};
        a.u0 = u; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = 1; a.alg = CNF_ALG_TSIT5; a.t0 = t; a.t1 = t + dt;
        a.u_out = u_new; a.nvars = h->cfg.nvars; a.reg_aug = 0; a.kfull = h->emb.buf;
        a.dt_exact = dt;   // the step the error estimate is scaled with, not fl(fl(t + dt) - t)
        a.ckpt = ckpt; a.ckpt_k = ckpt_k;
        HIP_TRY(mfma_solve(h->plan, h->par.packed_dev, a, st));
        StageIn last{};
        last.u = u_new; last.nprev = 0; last.dt = 0.f;
        rc = api_eval_dynamics(h, last, t + dt, eps, ys, B, k[6], stage, false, st);
        if (rc) return rc;
        HIP_TRY(embedded_error(u, u_new, k, btilde, 7, dt, abstol, reltol, (int64_t)n, h->emb.err_partial, err_sumsq, st));
        return CNF_OK;
    }
    if (!(flags & (CNF_STEP_FSAL | CNF_STEP_RETRY))) {
        StageIn in{};
        in.u = u; in.nprev = 0; in.dt = 0.f;
        rc = api_eval_dynamics(h, in, t, eps, ys, B, k[0], stage, true, st);
        if (rc) return rc;
    }
    for (int i = 1; i < 6; ++i) {
        StageIn in{};
        in.u = u; in.nprev = i; in.dt = dt;
        for (int j = 0; j < i; ++j) { in.k[j] = k[j]; in.coef[j] = T.a[i][j]; }
        rc = api_eval_dynamics(h, in, t + T.c[i] * dt, eps, ys, B, k[i], stage, false, st);
        if (rc) return rc;
    }
    StageIn fin{};
    fin.u = u; fin.nprev = 6; fin.dt = dt;
    for (int j = 0; j < 6; ++j) { fin.k[j] = k[j]; fin.coef[j] = T.b[j]; }
    HIP_TRY(rk_update(u_new, fin, (int64_t)n, st));
    StageIn last{};
    last.u = u_new; last.nprev = 0; last.dt = 0.f;
    rc = api_eval_dynamics(h, last, t + dt, eps, ys, B, k[6], stage, false, st);   // 7th stage = first stage of the next step
    if (rc) return rc;
    HIP_TRY(embedded_error(u, u_new, k, btilde, 7, dt, abstol, reltol, (int64_t)n, h->emb.err_partial, err_sumsq, st));
    return CNF_OK;
}

extern "C" {

// ---- variable-step variable-order Adams PECE: the reference's default alg = VCABM() ----

static inline float* vc_vec(cnf_handle* h, int i) { return h->vc.buf + (size_t)i * (size_t)h->S * (size_t)h->vc.B; }
static inline float* vc_diffs(cnf_handle* h, int half) { return vc_vec(h, 6 + half * kVcSlots); }

static int vc_check(cnf_handle* h, const float* eps, const float* ys, int64_t B, const char* who) {
    int rc = api_check_call(h, eps, ys, B, who);
    if (rc) return rc;
    if (h->vc.B < 0 || B != h->vc.B) return fail(CNF_ERR_INVALID, std::string(who) + ": cnf_vcabm_begin was not called for this batch");
    return CNF_OK;
}

int cnf_vcabm_begin(cnf_handle* h, float t0, const float* u0, const float* eps, const float* ys, int64_t B, void* stream) {
    int rc = api_check_call(h, eps, ys, B, "cnf_vcabm_begin");
    if (rc) return rc;
    if (B > 0 && !u0) return fail(CNF_ERR_INVALID, "cnf_vcabm_begin: null u0");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    if (B > h->vc.cap) {   // grown on demand only: alternating batch sizes (a shorter last mini-batch) reuse the allocation
        if (h->vc.buf) HIP_TRY(hipFree(h->vc.buf));
        h->vc.buf = nullptr; h->vc.cap = 0; h->vc.B = -1;
        HIP_TRY(hipMalloc((void**)&h->vc.buf, (6 + 2 * kVcSlots) * (size_t)h->S * (size_t)B * sizeof(float)));
        h->vc.cap = B;
    }
    h->vc.B = B;   // the vectors of this solve are packed at stride S x B inside the allocation
    if (!h->vc.partial) HIP_TRY(hipMalloc((void**)&h->vc.partial, (vcabm_partial_doubles() + 8) * sizeof(double)));   // + result slots of cnf_solve_vcabm
    h->vc.iu = 0; h->vc.iun = 2; h->vc.ifn0 = 3; h->vc.ifn1 = 5; h->vc.cur = 0;
    h->vc.nhist = 0; h->vc.k = 0; h->vc.avail = 0; h->vc.m = 0; h->vc.t = t0; h->vc.dt = 0.0;
    for (double& d : h->vc.hist) d = 0.0;
    if (B == 0) return CNF_OK;
    const size_t n = (size_t)h->S * (size_t)B;
    HIP_TRY(hipMemcpyAsync(vc_vec(h, h->vc.iu), u0, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    StageIn in{};
    in.u = vc_vec(h, h->vc.iu); in.nprev = 0; in.dt = 0.f;
    return api_eval_dynamics(h, in, t0, eps, ys, B, vc_vec(h, h->vc.ifn0), nullptr, true, st);
}

int cnf_vcabm_attempt(cnf_handle* h, int order, float dt, const float* eps, const float* ys, int64_t B, float abstol,
                      float reltol, double* err3, void* stream) {
    int rc = vc_check(h, eps, ys, B, "cnf_vcabm_attempt");
    if (rc) return rc;
    if (order < 1 || order > CNF_VCABM_MAX_ORDER || order > h->vc.nhist + 1)
        return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: order must be in 1..12 and at most one more than the accepted steps");
    if (std::min(order, h->vc.nhist) > h->vc.avail)
        return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: the order can rise by at most one per accepted step (the stored differences end there)");
    if (!(abstol >= 0.f) || !(reltol >= 0.f) || (abstol == 0.f && reltol == 0.f))
        return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: tolerances must be non-negative and not both zero");
    if (dt == 0.f || !(dt == dt)) return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: dt must be non-zero");
    if (B == 0) return CNF_OK;
    if (!err3) return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: null err3");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    const int k = order;
    const int m = std::min(k + 1, h->vc.nhist + 1);
    // step sizes newest first, the candidate in front (Hairer, Noersett, Wanner I, III.5: t_{n+1} - t_{n-j+1} = sum of the j newest)
    double dts[kVcSlots + 2];
    dts[0] = dt;
    for (int i = 0; i <= kVcSlots; ++i) dts[i + 1] = h->vc.hist[i];
    VcCoef c{};
    c.ps_old = vc_diffs(h, h->vc.cur);
    c.ps_new = vc_diffs(h, h->vc.cur ^ 1);
    c.ld = (size_t)h->S * (size_t)B;
    c.k = k; c.m = m; c.dt = dt;
    double beta = 1.0, num = 0.0, den = 0.0;
    c.beta[0] = 1.f;
    for (int j = 1; j < m; ++j) {       // beta_j = beta_{j-1} (t_{n+1} - t_{n-j+1}) / (t_n - t_{n-j})
        num += dts[j - 1];
        den += dts[j];
        beta *= num / den;
        c.beta[j] = (float)beta;
    }
    // g_j = c_{j,1};  c_{0,q} = 1/q,  c_{1,q} = 1/(q(q+1)),  c_{j,q} = c_{j-1,q} - c_{j-1,q+1} dt / (t_{n+1} - t_{n-j+1})
    double cq[kVcSlots + 3], gd[kVcSlots + 1];
    const int ng = k + 1;
    gd[0] = 1.0;
    for (int q = 1; q <= ng; ++q) cq[q - 1] = 1.0 / ((double)q * (double)(q + 1));
    double xi = dts[0];
    for (int j = 1; j < ng; ++j) {
        if (j > 1) {
            xi += dts[j - 1];
            for (int q = 0; q < ng - j + 1; ++q) cq[q] = cq[q] - cq[q + 1] * (double)dt / xi;
        }
        gd[j] = cq[0];
    }
    for (int j = 0; j < ng; ++j) c.g[j] = (float)gd[j];
    c.e0 = (float)((double)dt * (gd[k] - gd[k - 1]));
    c.e1 = k >= 2 ? (float)((double)dt * (gd[k - 1] - gd[k - 2])) : 0.f;
    c.e2 = k >= 3 ? (float)((double)dt * (gd[k - 2] - gd[k - 3])) : 0.f;
    const int64_t n = (int64_t)c.ld;
    float *u = vc_vec(h, h->vc.iu), *p = vc_vec(h, 1), *un = vc_vec(h, h->vc.iun), *d = vc_vec(h, 4);
    HIP_TRY(vcabm_predict(vc_vec(h, h->vc.ifn0), u, c, n, p, st));                                    // P
    StageIn in{};
    in.u = p; in.nprev = 0; in.dt = 0.f;
    rc = api_eval_dynamics(h, in, (float)(h->vc.t + (double)dt), eps, ys, B, d, nullptr, false, st);   // E
    if (rc) return rc;
    HIP_TRY(vcabm_correct(d, p, u, c, abstol, reltol, n, un, h->vc.partial, err3, st));             // C
    h->vc.k = k; h->vc.m = m; h->vc.dt = dt;
    return CNF_OK;
}

int cnf_vcabm_accept(cnf_handle* h, const float* eps, const float* ys, int64_t B, float abstol, float reltol,
                     double* err_up, void* stream) {
    int rc = vc_check(h, eps, ys, B, "cnf_vcabm_accept");
    if (rc) return rc;
    if (B == 0) return CNF_OK;
    if (h->vc.k == 0) return fail(CNF_ERR_INVALID, "cnf_vcabm_accept: no pending attempt");
    const int k = h->vc.k;
    if (err_up && (k >= CNF_VCABM_MAX_ORDER || h->vc.nhist < k))
        return fail(CNF_ERR_INVALID, "cnf_vcabm_accept: the order k+1 estimate needs k accepted steps and k < 12");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)h->S * (size_t)B;
    float *u = vc_vec(h, h->vc.iu), *un = vc_vec(h, h->vc.iun), *fnew = vc_vec(h, h->vc.ifn1);
    StageIn in{};
    in.u = un; in.nprev = 0; in.dt = 0.f;
    rc = api_eval_dynamics(h, in, (float)(h->vc.t + h->vc.dt), eps, ys, B, fnew, nullptr, false, st);   // E
    if (rc) return rc;
    if (err_up) {
        // gamma*_j of the Adams-Moulton family: sum_{i<=j} gamma*_i / (j - i + 1) = [j == 0]
        double gs[kVcSlots + 2];
        gs[0] = 1.0;
        for (int j = 1; j <= k + 1; ++j) {
            double a = 0.0;
            for (int i = 0; i < j; ++i) a += gs[i] / (double)(j - i + 1);
            gs[j] = -a;
        }
        VcCoef c{};
        c.ps_new = vc_diffs(h, h->vc.cur ^ 1);
        c.ld = n; c.k = k;
        c.e0 = (float)(h->vc.dt * gs[k + 1]);
        HIP_TRY(vcabm_errup(fnew, u, un, c, abstol, reltol, (int64_t)n, h->vc.partial, err_up, st));
    }
    std::swap(h->vc.iu, h->vc.iun);
    std::swap(h->vc.ifn0, h->vc.ifn1);
    h->vc.cur ^= 1;
    for (int i = kVcSlots; i > 0; --i) h->vc.hist[i] = h->vc.hist[i - 1];
    h->vc.hist[0] = h->vc.dt;
    h->vc.t += h->vc.dt;
    h->vc.nhist += 1;
    h->vc.avail = h->vc.m;
    h->vc.k = 0;
    return CNF_OK;
}

int cnf_vcabm_state(cnf_handle* h, int64_t B, float* u_out, double* t_out, void* stream) {
    if (!h) return fail(CNF_ERR_INVALID, "cnf_vcabm_state: null handle");
    if (h->vc.B < 0 || B != h->vc.B) return fail(CNF_ERR_INVALID, "cnf_vcabm_state: cnf_vcabm_begin was not called for this batch");
    if (t_out) *t_out = h->vc.t;
    if (u_out && B > 0) {
        DeviceGuard g(h->cfg.device_id);
        HIP_TRY(hipMemcpyAsync(u_out, vc_vec(h, h->vc.iu), (size_t)h->S * (size_t)B * sizeof(float), hipMemcpyDeviceToDevice,
                               (hipStream_t)stream));
    }
    return CNF_OK;
}

// The whole default solve in one call: cnf_vcabm_begin / _attempt / _accept driven by the step-size and order policy of
// solvers._vcabm_integrate (the host side of the reference's solver), restated here so that a single-process caller pays one
// library call per solve instead of two per step.  Synchronises `stream` (the policy reads the error sums).
int cnf_solve_vcabm(cnf_handle* h, float t0, float t1, const float* u0, const float* eps, const float* ys, int64_t B,
                    float abstol, float reltol, float dt_init, int maxiters, float* u1, cnf_solve_stats* stats,
                    float* dts_out, int32_t* orders_out, int32_t record_cap, void* stream) {
    if (stats) *stats = cnf_solve_stats{};
    int rc = api_check_call(h, eps, ys, B, "cnf_solve_vcabm");
    if (rc) return rc;
    if (!(abstol >= 0.f) || !(reltol >= 0.f) || (abstol == 0.f && reltol == 0.f))
        return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: tolerances must be non-negative and not both zero");
    if (B > 0 && (!u0 || !u1)) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: null u0/u1");
    if (maxiters < 1) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: maxiters >= 1 required");
    if (B > 0 && t1 != t0 && h->path == CNF_PATH_MFMA && h->plan && B <= mfma_vcabm_capacity(h->plan)) {
        // the batch fits the chip's wave slots: passes, error norms and the step / order policy in one launch
        h->adp.last_controller = 1;
        DeviceGuard g(h->cfg.device_id);
        hipStream_t st = (hipStream_t)stream;
        const int dts_cap = maxiters < (1 << 20) ? maxiters : (1 << 20);
        const size_t need = mfma_adaptive_scratch_bytes(B, dts_cap);
        if (need > h->adp.dc_bytes) {
            if (h->adp.dc_buf) HIP_TRY(hipFree(h->adp.dc_buf));
            h->adp.dc_buf = nullptr; h->adp.dc_bytes = 0; h->adp.dc_epoch = 0;
            HIP_TRY(hipMalloc(&h->adp.dc_buf, need));
            h->adp.dc_bytes = need;
        }
        SolveArgs a{};
        a.u0 = u0; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = 1; a.alg = CNF_ALG_TSIT5; a.t0 = t0; a.t1 = t1;
        a.u_out = u1; a.nvars = h->cfg.nvars; a.reg_aug = 0;
        int *stats_dev = nullptr, *orders_dev = nullptr;
        float* dts_dev = nullptr;
        if (!h->adp.host_rec) HIP_TRY(hipHostMalloc((void**)&h->adp.host_rec, (8 + 2 * kHostRec) * sizeof(int), hipHostMallocDefault));
        const hipError_t le = mfma_solve_vcabm(h->plan, h->par.packed_dev, a, abstol, reltol, dt_init, maxiters, h->adp.dc_buf, &h->adp.dc_epoch, dts_cap,
                                               &stats_dev, &dts_dev, &orders_dev, h->adp.host_rec, st);
        if (le != hipSuccess) {
            (void)hipGetLastError();
            return fail(CNF_ERR_HIP, std::string("cnf_solve_vcabm: launch of the device-resident solve failed: ") + hipGetErrorString(le));
        }
        // the kernel wrote its status words and first accepted steps into pinned host memory: no copy, one synchronisation
        HIP_TRY(hipStreamSynchronize(st));
        const int* hs = h->adp.host_rec;
        if (stats) { stats->naccept = hs[0]; stats->nreject = hs[1]; stats->nf = hs[2]; stats->max_order = hs[4]; }
        if (hs[3] == 1) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: non-finite error estimate (unstable dynamics)");
        if (hs[3] == 2) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: maxiters reached");
        if (hs[3] == 3) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: non-finite state or dynamics at t0 (no initial step)");
        if (hs[3] == 4 || hs[5] != 0) return fail(CNF_ERR_HIP, "cnf_solve_vcabm: the grid-wide sum of the one-launch solve timed out (workgroups not all resident); CNF_DEVICE_CONTROLLER=0 selects the host loop");
        const int na = std::min(std::min(hs[0], dts_cap), (int)record_cap);
        if (na <= kHostRec) {
            for (int i = 0; i < na; ++i) {
                if (dts_out) memcpy(dts_out + i, hs + 8 + 2 * i, sizeof(float));
                if (orders_out) orders_out[i] = hs[9 + 2 * i];
            }
        } else {
            if (dts_out) HIP_TRY(hipMemcpyAsync(dts_out, dts_dev, (size_t)na * sizeof(float), hipMemcpyDeviceToHost, st));
            if (orders_out) HIP_TRY(hipMemcpyAsync(orders_out, orders_dev, (size_t)na * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            if (dts_out || orders_out) HIP_TRY(hipStreamSynchronize(st));
        }
        h->vc.B = -1;   // the step-wise entry points have no state from this solve
        return CNF_OK;
    }
    h->adp.last_controller = 0;
    rc = cnf_vcabm_begin(h, t0, u0, eps, ys, B, stream);
    if (rc) return rc;
    int nf = B > 0 ? 1 : 0, naccept = 0, nreject = 0, max_order = 0;
    const double span = std::fabs((double)t1 - (double)t0), tdir = t1 >= t0 ? 1.0 : -1.0;
    if (B == 0 || span == 0.0) {
        if (B > 0) return cnf_vcabm_state(h, B, u1, nullptr, stream);
        return CNF_OK;
    }
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)h->S * (size_t)B;
    const double ntot = (double)n;
    if (!h->vc.host_res) HIP_TRY(hipHostMalloc((void**)&h->vc.host_res, 8 * sizeof(double), hipHostMallocDefault));
    double* res = h->vc.host_res;   // result slots in pinned host memory: the reduction kernels write them, the loop synchronises and reads
    double host[4];
    auto fetch = [&](int cnt) -> int {
        HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < cnt; ++i) host[i] = res[i];
        return CNF_OK;
    };
    double dt;
    if (dt_init != 0.f) {
        dt = std::min((double)std::fabs(dt_init), span);
    } else {   // ode_determine_initdt (Hairer, Noersett, Wanner I, II.4), RMS norm over all S*B entries; the exponent is 1 / get_current_alg_order = 1 / (the cache's current order) = 1 at the start of a VCABM solve (recalled, not read)
        float *u = vc_vec(h, h->vc.iu), *f0 = vc_vec(h, h->vc.ifn0), *ue = vc_vec(h, 1), *f1 = vc_vec(h, 4);
        HIP_TRY(vcabm_scaled_sumsq(u, nullptr, u, abstol, reltol, (int64_t)n, h->vc.partial, res, st));
        HIP_TRY(vcabm_scaled_sumsq(f0, nullptr, u, abstol, reltol, (int64_t)n, h->vc.partial, res + 1, st));
        rc = fetch(2);
        if (rc) return rc;
        const double d0 = std::sqrt(host[0] / ntot), d1 = std::sqrt(host[1] / ntot);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = std::min(h0, span);
        StageIn eu{};
        eu.u = u; eu.nprev = 1; eu.k[0] = f0; eu.coef[0] = 1.f; eu.dt = (float)(tdir * h0);
        HIP_TRY(rk_update(ue, eu, (int64_t)n, st));
        StageIn in{};
        in.u = ue; in.nprev = 0; in.dt = 0.f;
        rc = api_eval_dynamics(h, in, (float)((double)t0 + tdir * h0), eps, ys, B, f1, nullptr, false, st);
        if (rc) return rc;
        ++nf;
        HIP_TRY(vcabm_scaled_sumsq(f1, f0, u, abstol, reltol, (int64_t)n, h->vc.partial, res, st));
        rc = fetch(1);
        if (rc) return rc;
        const double d2 = std::sqrt(host[0] / ntot) / h0, dmax = std::max(d1, d2);
        const double h1 = dmax <= 1e-15 ? std::max(1e-6, h0 * 1e-3) : std::pow(10.0, -(2.0 + std::log10(dmax)) / 1.0);
        dt = std::min(std::min(100.0 * h0, h1), span);
        if (!(std::isfinite(dt) && dt > 0.0)) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: non-finite state or dynamics at t0 (no initial step)");
    }
    const double gamma = 0.9, qmin = 0.2, qmax = 10.0;
    double t = t0;
    int k = 1, step = 1, it = 0;
    for (; it < maxiters; ++it) {
        if (std::fabs((double)t1 - t) <= 1e-7 * std::max(1.0, span)) break;
        const bool last = dt >= std::fabs((double)t1 - t) * (1.0 - 1e-6);
        const double hstep = last ? std::fabs((double)t1 - t) : dt;     // tstop: never step over t1
        rc = cnf_vcabm_attempt(h, k, (float)(tdir * hstep), eps, ys, B, abstol, reltol, res, stream);
        if (rc) return rc;
        ++nf;
        rc = fetch(3);
        if (rc) return rc;
        double eest = std::sqrt(host[0] / ntot);
        if (!std::isfinite(eest)) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: non-finite error estimate (unstable dynamics)");
        if (eest > 1.0) {   // reject: same state, smaller step, same order
            ++nreject;
            dt = hstep / std::max(1.0 / qmax, std::min(1.0 / qmin, std::pow(eest, 1.0 / (k + 1)) / gamma));
            continue;
        }
        const bool select = step > 4 && k >= 3;
        const bool lower = select && std::max(std::sqrt(host[2] / ntot), std::sqrt(host[1] / ntot)) <= eest;
        const bool want_up = select && !lower && k < CNF_VCABM_MAX_ORDER;
        rc = cnf_vcabm_accept(h, eps, ys, B, abstol, reltol, want_up ? res : nullptr, stream);
        if (rc) return rc;
        ++nf;
        int knew = k;
        if (!select) knew = std::min(k + 1, 3);
        else if (lower) knew = k - 1;
        else if (want_up) {
            rc = fetch(1);
            if (rc) return rc;
            if (std::sqrt(host[0] / ntot) < eest) { knew = k + 1; eest = 1.0; }
        }
        const double q = eest == 0.0 ? 1.0 / qmax : std::max(1.0 / qmax, std::min(1.0 / qmin, std::pow(eest, 1.0 / (knew + 1)) / gamma));
        t = last ? (double)t1 : t + tdir * hstep;
        if (naccept < record_cap) {
            if (dts_out) dts_out[naccept] = (float)(tdir * hstep);
            if (orders_out) orders_out[naccept] = k;
        }
        ++naccept;
        if (k > max_order) max_order = k;
        k = knew; ++step;
        dt = hstep / q;
    }
    if (stats) { stats->naccept = naccept; stats->nreject = nreject; stats->nf = nf; stats->max_order = max_order; }
    if (it == maxiters) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: maxiters reached");
    return cnf_vcabm_state(h, B, u1, nullptr, stream);
}

// Adaptive Tsit5 from t0 to t1 in one call: cnf_step_embedded attempts under OrdinaryDiffEq's PI controller - the loop of
// solvers._adaptive_integrate restated inside the library (single process).  Synchronises `stream`.
extern "C++" {
int cnf::api_ensure_adaptive_buf(cnf_handle* h, int64_t B) {
    if (B <= h->adp.B) return CNF_OK;
    if (h->adp.buf) HIP_TRY(hipFree(h->adp.buf));
    h->adp.buf = nullptr; h->adp.B = 0;
    HIP_TRY(hipMalloc((void**)&h->adp.buf, 6 * (size_t)h->S * (size_t)B * sizeof(float)));   // 4 for the solve, 2 for cnf_loss_grad_adaptive
    h->adp.B = B;
    return CNF_OK;
}
}  // extern "C++"

extern "C++" {
int cnf::api_solve_tsit5(cnf_handle* h, float t0, float t1, const float* u0, const float* eps, const float* ys, int64_t B,
                         float abstol, float reltol, float dt_init, int maxiters, float* u1, cnf_solve_stats* stats,
                         std::vector<double>* steps, void* stream, TsitCkpt* ck) {
    if (stats) *stats = cnf_solve_stats{};
    if (ck) ck->ok = false;
    int rc = api_check_call(h, eps, ys, B, "cnf_solve_tsit5");
    if (rc) return rc;
    if (!(abstol >= 0.f) || !(reltol >= 0.f) || (abstol == 0.f && reltol == 0.f))
        return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: tolerances must be non-negative and not both zero");
    if (B > 0 && (!u0 || !u1)) return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: null u0/u1");
    if (maxiters < 1) return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: maxiters >= 1 required");
    if (B == 0) return CNF_OK;
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)h->S * (size_t)B;
    const double span = std::fabs((double)t1 - (double)t0), tdir = t1 >= t0 ? 1.0 : -1.0, ntot = (double)n;
    if (span == 0.0) {
        if (u1 != u0) HIP_TRY(hipMemcpyAsync(u1, u0, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        return CNF_OK;
    }
    if (h->path == CNF_PATH_MFMA && h->plan && B <= mfma_adaptive_capacity(h->plan)) {
        // the batch fits the chip's wave slots: the whole solve, step controller included, in one launch
        h->adp.last_controller = 1;
        const int dts_cap = maxiters < (1 << 20) ? maxiters : (1 << 20);
        const size_t need = mfma_adaptive_scratch_bytes(B, dts_cap);
        if (need > h->adp.dc_bytes) {
            if (h->adp.dc_buf) HIP_TRY(hipFree(h->adp.dc_buf));
            h->adp.dc_buf = nullptr; h->adp.dc_bytes = 0; h->adp.dc_epoch = 0;
            HIP_TRY(hipMalloc(&h->adp.dc_buf, need));
            h->adp.dc_bytes = need;
        }
        SolveArgs a{};
        a.u0 = u0; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = 1; a.alg = CNF_ALG_TSIT5; a.t0 = t0; a.t1 = t1;
        a.u_out = u1; a.nvars = h->cfg.nvars; a.reg_aug = 0;
        if (ck && ck->cap > 0) { a.ckpt = ck->ckpt; a.ckpt_k = ck->ckpt_k; }
        int* stats_dev = nullptr;
        float* dts_dev = nullptr;
        if (!h->adp.host_rec) HIP_TRY(hipHostMalloc((void**)&h->adp.host_rec, (8 + 2 * kHostRec) * sizeof(int), hipHostMallocDefault));
        const hipError_t le = mfma_solve_adaptive(h->plan, h->par.packed_dev, a, abstol, reltol, dt_init, maxiters, h->adp.dc_buf, &h->adp.dc_epoch, dts_cap, &stats_dev, &dts_dev,
                                                  h->adp.host_rec, ck ? ck->cap : 0, st);
        if (le != hipSuccess) {
            (void)hipGetLastError();   // not sticky: nothing was launched
            return fail(CNF_ERR_HIP, std::string("cnf_solve_tsit5: launch of the device-controlled solve failed: ") + hipGetErrorString(le));
        }
        // the kernel wrote its status words and first accepted steps into pinned host memory: no copy, one synchronisation
        HIP_TRY(hipStreamSynchronize(st));
        const int* host = h->adp.host_rec;
        if (stats) { stats->naccept = host[0]; stats->nreject = host[1]; stats->nf = host[2]; stats->max_order = 5; }
        if (host[3] == 1) return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: non-finite error estimate (unstable dynamics)");
        if (host[3] == 2) return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: maxiters reached");
        if (host[3] == 3) return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: non-finite state or dynamics at t0 (no initial step)");
        if (host[3] == 4 || host[5] != 0) return fail(CNF_ERR_HIP, "cnf_solve_tsit5: the grid-wide sum of the one-launch solve timed out (workgroups not all resident); CNF_DEVICE_CONTROLLER=0 selects the host loop");
        if (ck) ck->ok = ck->cap > 0 && host[6] == 1;
        if (steps) {
            const int na = host[0] < dts_cap ? host[0] : dts_cap;
            std::vector<float> all((size_t)na);
            if (na <= kHostRec) {
                for (int i = 0; i < na; ++i) memcpy(&all[i], host + 8 + 2 * i, sizeof(float));
            } else {
                HIP_TRY(hipMemcpyAsync(all.data(), dts_dev, (size_t)na * sizeof(float), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
            for (int i = 0; i < na; ++i) steps->push_back((double)all[i]);
        }
        return CNF_OK;
    }
    h->adp.last_controller = 0;
    rc = api_ensure_adaptive_buf(h, B);
    if (rc) return rc;
    if (!h->vc.partial) HIP_TRY(hipMalloc((void**)&h->vc.partial, (vcabm_partial_doubles() + 8) * sizeof(double)));
    const size_t slot = (size_t)h->S * (size_t)h->adp.B;
    float *ua = h->adp.buf, *ub = ua + slot, *f0 = ub + slot, *f1 = f0 + slot;
    HIP_TRY(hipMemcpyAsync(ua, u0, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (!h->vc.host_res) HIP_TRY(hipHostMalloc((void**)&h->vc.host_res, 8 * sizeof(double), hipHostMallocDefault));
    double* res = h->vc.host_res;   // pinned host memory, as in cnf_solve_vcabm
    double host[2];
    auto fetch = [&](int cnt) -> int {
        HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < cnt; ++i) host[i] = res[i];
        return CNF_OK;
    };
    int nf = 0, naccept = 0, nreject = 0;
    double dt;
    if (dt_init != 0.f) {
        dt = std::min((double)std::fabs(dt_init), span);
    } else {   // ode_determine_initdt (Hairer, Noersett, Wanner I, II.4), order 5
        StageIn in{};
        in.u = ua; in.nprev = 0; in.dt = 0.f;
        rc = api_eval_dynamics(h, in, t0, eps, ys, B, f0, nullptr, true, st);
        if (rc) return rc;
        ++nf;
        HIP_TRY(vcabm_scaled_sumsq(ua, nullptr, ua, abstol, reltol, (int64_t)n, h->vc.partial, res, st));
        HIP_TRY(vcabm_scaled_sumsq(f0, nullptr, ua, abstol, reltol, (int64_t)n, h->vc.partial, res + 1, st));
        rc = fetch(2);
        if (rc) return rc;
        const double d0 = std::sqrt(host[0] / ntot), d1 = std::sqrt(host[1] / ntot);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = std::min(h0, span);
        if (!(std::isfinite(h0) && h0 > 0.0)) return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: non-finite state or dynamics at t0 (no initial step)");
        StageIn eu{};
        eu.u = ua; eu.nprev = 1; eu.k[0] = f0; eu.coef[0] = 1.f; eu.dt = (float)(tdir * h0);
        HIP_TRY(rk_update(ub, eu, (int64_t)n, st));
        StageIn in1{};
        in1.u = ub; in1.nprev = 0; in1.dt = 0.f;
        rc = api_eval_dynamics(h, in1, (float)((double)t0 + tdir * h0), eps, ys, B, f1, nullptr, false, st);
        if (rc) return rc;
        ++nf;
        HIP_TRY(vcabm_scaled_sumsq(f1, f0, ua, abstol, reltol, (int64_t)n, h->vc.partial, res, st));
        rc = fetch(1);
        if (rc) return rc;
        const double d2 = std::sqrt(host[0] / ntot) / h0, dmax = std::max(d1, d2);
        const double h1 = dmax <= 1e-15 ? std::max(1e-6, h0 * 1e-3) : std::pow(10.0, -(2.0 + std::log10(dmax)) / 5.0);
        dt = std::min(std::min(100.0 * h0, h1), span);
        if (!(std::isfinite(dt) && dt > 0.0)) return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: non-finite state or dynamics at t0 (no initial step)");
    }
    const double beta1 = 7.0 / 50.0, beta2 = 2.0 / 25.0, gamma = 0.9, qmin = 0.2, qmax = 10.0;
    double qold = 1e-4, t = t0;
    int flags = 0, it = 0;
    const bool ck_live = ck && ck->cap > 0 && ck->ckpt && ck->ckpt_k && h->path == CNF_PATH_MFMA && mfma_plan_is_per_wave(h->plan);
    const size_t ck_slot = ck_live ? (size_t)((B + 15) / 16) * 64 * (size_t)mfma_plan_zr(h->plan) : 0;
    for (; it < maxiters; ++it) {
        if (std::fabs((double)t1 - t) <= 1e-7 * std::max(1.0, span)) break;
        const bool last = dt >= std::fabs((double)t1 - t) * (1.0 - 1e-6);
        const double step = last ? std::fabs((double)t1 - t) : dt;     // tstop: never step over t1
        // (per-wave plans: the fused attempt of step `naccept` also fills that step's checkpoint slots - z_n, the six stage
        // derivatives, z_{n+1}; a retry overwrites them)
        float *cz = nullptr, *ckk = nullptr;
        if (ck_live && naccept < ck->cap) { cz = ck->ckpt + (size_t)naccept * ck_slot; ckk = ck->ckpt_k + (size_t)naccept * 6 * ck_slot; }
        rc = step_embedded_impl(h, CNF_ALG_TSIT5, flags, (float)t, (float)(tdir * step), ua, eps, ys, B, abstol, reltol, ub, res, stream, cz, ckk);
        if (rc) return rc;
        nf += flags ? 6 : 7;
        rc = fetch(1);
        if (rc) return rc;
        const double eest = std::sqrt(host[0] / ntot);
        if (!std::isfinite(eest)) return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: non-finite error estimate (unstable dynamics)");
        const double q11 = eest > 0.0 ? std::pow(eest, beta1) : 0.0;
        const double q = eest == 0.0 ? 1.0 / qmax : std::max(1.0 / qmax, std::min(1.0 / qmin, (q11 / std::pow(qold, beta2)) / gamma));
        if (eest <= 1.0) {   // accept
            t = last ? (double)t1 : t + tdir * step;
            std::swap(ua, ub);
            if (steps) steps->push_back(tdir * step);
            ++naccept;
            qold = std::max(eest, 1e-4);
            dt = step / q;
            flags = CNF_STEP_FSAL;
        } else {             // reject: same (t, u), smaller step
            ++nreject;
            dt = step / std::min(1.0 / qmin, q11 / gamma);
            flags = CNF_STEP_RETRY;
        }
    }
    if (stats) { stats->naccept = naccept; stats->nreject = nreject; stats->nf = nf; stats->max_order = 5; }
    if (it == maxiters) return fail(CNF_ERR_INVALID, "cnf_solve_tsit5: maxiters reached");
    HIP_TRY(hipMemcpyAsync(u1, ua, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (ck) ck->ok = ck_live && naccept <= ck->cap;
    return CNF_OK;
}
}  // extern "C++"

int cnf_loss_adaptive(cnf_handle* h, int alg, float t0, float t1, const float* x, const float* eps, const float* ys, int64_t B,
                      float abstol, float reltol, float dt_init, int maxiters, const double* lambdas, float* loss, float* sums4,
                      float* logp_out, float* regs_out, cnf_solve_stats* stats, float* dts_out, int32_t* orders_out,
                      int32_t record_cap, void* stream) {
    if (stats) *stats = cnf_solve_stats{};
    int rc = api_check_call(h, eps, ys, B, "cnf_loss_adaptive");
    if (rc) return rc;
    if (alg != CNF_ALG_VCABM && alg != CNF_ALG_TSIT5) return fail(CNF_ERR_INVALID, "cnf_loss_adaptive: alg must be CNF_ALG_VCABM or CNF_ALG_TSIT5");
    if (B < 1) return fail(CNF_ERR_INVALID, "cnf_loss_adaptive: the mean of an empty batch is undefined");
    if (!x || !lambdas || !loss) return fail(CNF_ERR_INVALID, "cnf_loss_adaptive: null x/lambdas/loss");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    rc = api_ensure_adaptive_buf(h, B);
    if (rc) return rc;
    const size_t slot = (size_t)h->S * (size_t)h->adp.B;
    float* u0 = h->adp.buf + 4 * slot;     // the two slots the solves themselves do not use (as cnf_loss_grad_adaptive)
    float* u1 = u0 + slot;
    HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, u0, st));
    if (alg == CNF_ALG_VCABM) {
        rc = cnf_solve_vcabm(h, t0, t1, u0, eps, ys, B, abstol, reltol, dt_init, maxiters, u1, stats, dts_out, orders_out, record_cap, stream);
    } else {
        rc = cnf_solve_tsit5(h, t0, t1, u0, eps, ys, B, abstol, reltol, dt_init, maxiters, u1, stats, dts_out, record_cap, stream);
    }
    if (rc) return rc;
    // u0 is spent: its slot (S >= 4 floats per column) takes the per-sample outputs the caller did not ask for
    float* logp = logp_out ? logp_out : u0;
    float* regs = regs_out ? regs_out : u0 + B;
    const int reg_aug = (h->cfg.reg_aug && h->cfg.naug > 0 && h->cfg.mode != CNF_MODE_EXACT) ? 1 : 0;
    HIP_TRY(epilogue(u1, h->cfg.nvars, h->D, reg_aug, B, logp, regs, st));
    if (!h->loss_partial) HIP_TRY(hipMalloc((void**)&h->loss_partial, 256 * 4 * sizeof(float)));
    HIP_TRY(loss_mean(logp, regs, B, h->loss_partial, sums4, loss, lambdas, st));
    return CNF_OK;
}

int cnf_solve_tsit5(cnf_handle* h, float t0, float t1, const float* u0, const float* eps, const float* ys, int64_t B,
                    float abstol, float reltol, float dt_init, int maxiters, float* u1, cnf_solve_stats* stats,
                    float* dts_out, int32_t record_cap, void* stream) {
    std::vector<double> steps;
    const int rc = api_solve_tsit5(h, t0, t1, u0, eps, ys, B, abstol, reltol, dt_init, maxiters, u1, stats, &steps, stream);
    if (dts_out)
        for (size_t i = 0; i < steps.size() && (int64_t)i < record_cap; ++i) dts_out[i] = (float)steps[i];
    return rc;
}

}  // extern "C"
