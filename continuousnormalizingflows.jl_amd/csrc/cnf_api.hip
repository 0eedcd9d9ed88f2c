// cnf_api.hip — the C ABI of libcnf_hip.so (see include/cnf.h for the contract): handles, parameters and the
// fixed-step entry points.  (cnf_api_adaptive.hip: adaptive solves; cnf_api_grad.hip: gradients; cnf_handle.h: shared.)
//
// Host-side only: handle lifetime, validation, parameter repacking, workspace management and
// dispatch to the kernel families (cnf_mfma.hip: fused whole-solve MFMA kernels;
// cnf_simt.hip: generic per-call kernels).  There is no CPU fallback: without a gfx950 device
// every entry point that would compute returns CNF_ERR_NO_DEVICE / CNF_ERR_HIP.
#include "cnf_handle.h"

using namespace cnf;

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) { return cnf::api_fail(code, msg); }
}  // namespace

int cnf::api_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

namespace {

__global__ void gather_pack_kernel(const float* __restrict__ p, const int* __restrict__ idx,
                                   const float* __restrict__ scale, float* __restrict__ out, size_t n) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int i = idx[j];
    out[j] = i >= 0 ? p[i] * scale[j] : 0.f;
}

// returns false (map left invalid) when the image is not a verified gather of the parameters
bool build_pack_map(PackMap& m, size_t nparams, size_t npacked,
                    const std::function<void(const float*, float*)>& pack, size_t skip_off = 0, size_t skip_len = 0) {
    m.valid = false;
    if (nparams == 0 || nparams >= (1u << 20)) return false;   // the ramp must stay exact after scaling
    std::vector<float> ones(nparams, 1.f), ramp(nparams), rnd(nparams);
    for (size_t i = 0; i < nparams; ++i) {
        ramp[i] = (float)(i + 1);
        rnd[i] = (float)((double)((i * 2654435761u) & 0xffffu) / 65536.0 - 0.5) * 1.7f;
    }
    std::vector<float> p1(npacked, 0.f), p2(npacked, 0.f), p3(npacked, 0.f);
    pack(ones.data(), p1.data());
    pack(ramp.data(), p2.data());
    pack(rnd.data(), p3.data());
    std::vector<int> idx(npacked);
    for (size_t j = 0; j < npacked; ++j) {
        if (j >= skip_off && j < skip_off + skip_len) {   // region filled by its own device packer (not a gather)
            idx[j] = -1; p1[j] = 0.f;
            continue;
        }
        if (p1[j] == 0.f) {
            if (p2[j] != 0.f || p3[j] != 0.f) return false;
            idx[j] = -1;
            continue;
        }
        const double q = (double)p2[j] / (double)p1[j];
        const long long i = llround(q) - 1;
        if (!(i >= 0 && (size_t)i < nparams) || std::fabs(q - (double)(i + 1)) > 0.25) return false;
        idx[j] = (int)i;
        const float want = rnd[(size_t)i] * p1[j];   // what the kernel will compute
        if (std::memcmp(&want, &p3[j], sizeof(float)) != 0) return false;
    }
    if (m.n != npacked) {
        if (m.idx) (void)hipFree(m.idx);
        if (m.scale) (void)hipFree(m.scale);
        m.idx = nullptr; m.scale = nullptr; m.n = 0;
        if (hipMalloc((void**)&m.idx, npacked * sizeof(int)) != hipSuccess) return false;
        if (hipMalloc((void**)&m.scale, npacked * sizeof(float)) != hipSuccess) return false;
        m.n = npacked;
    }
    if (hipMemcpy(m.idx, idx.data(), npacked * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return false;
    if (hipMemcpy(m.scale, p1.data(), npacked * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return false;
    m.valid = true;
    return true;
}

void free_pack_map(PackMap& m) {
    if (m.idx) (void)hipFree(m.idx);
    if (m.scale) (void)hipFree(m.scale);
    m = PackMap{};
}

}  // namespace


static int ensure_ws(cnf_handle* h, int64_t B) {
    if (B <= h->simt.ws_B) return CNF_OK;
    if (h->simt.ws) HIP_TRY(hipFree(h->simt.ws));
    h->simt.ws = nullptr;
    h->simt.ws_B = 0;
    const int64_t Bp = (B + 255) / 256 * 256;
    HIP_TRY(hipMalloc((void**)&h->simt.ws, simt_ws_rows(h->net) * (size_t)Bp * sizeof(float)));
    h->simt.ws_B = Bp;
    return CNF_OK;
}

static int ensure_kbuf(cnf_handle* h, int64_t B) {
    if (B <= h->simt.kbuf_B) return CNF_OK;
    if (h->simt.kbuf) HIP_TRY(hipFree(h->simt.kbuf));
    h->simt.kbuf = nullptr;
    h->simt.kbuf_B = 0;
    HIP_TRY(hipMalloc((void**)&h->simt.kbuf, 7 * (size_t)h->S * (size_t)B * sizeof(float)));
    h->simt.kbuf_B = B;
    return CNF_OK;
}

// One dynamics evaluation on the generic families: CNF_LAYERED_MIN_B=n sends batches below n of an AUTO-resolved LAYERED handle
// to the SIMT kernels instead
static int64_t layered_min_batch() {
    const char* e = getenv("CNF_LAYERED_MIN_B");
    return (e && *e) ? atoll(e) : 0;   // measured: the GEMM path wins at every batch size (profiles/archive/r1h_generic_small.json)
}

extern "C" {

int cnf_version(void) { return CNF_ABI_VERSION; }

const char* cnf_last_error(void) { return g_err.c_str(); }

int cnf_create(cnf_handle** out, const cnf_config* cfg) {
    if (!out || !cfg) return fail(CNF_ERR_INVALID, "cnf_create: null argument");
    *out = nullptr;
    const cnf_config& c = *cfg;
    if (c.nvars < 1 || c.naug < 0 || c.ncond < 0)
        return fail(CNF_ERR_INVALID, "cnf_create: nvars >= 1, naug >= 0, ncond >= 0 required");
    if (c.n_layers < 1 || c.n_layers > CNF_MAX_LAYERS)
        return fail(CNF_ERR_INVALID, "cnf_create: n_layers out of range");
    const int D = c.nvars + c.naug;
    const int n_in = D + (c.autonomous ? 0 : 1) + c.ncond;  // src/core/icnf.jl:64
    if (c.widths[0] != n_in)
        return fail(CNF_ERR_INVALID, "cnf_create: widths[0] must equal nvars+naug+!autonomous+ncond");
    if (c.widths[c.n_layers] != D)
        return fail(CNF_ERR_INVALID, "cnf_create: last width must equal nvars+naug");
    for (int l = 0; l <= c.n_layers; ++l)
        if (c.widths[l] < 1) return fail(CNF_ERR_INVALID, "cnf_create: widths must be positive");
    for (int l = 0; l < c.n_layers; ++l)
        if (c.acts[l] < CNF_ACT_IDENTITY || c.acts[l] > CNF_ACT_SOFTPLUS)
            return fail(CNF_ERR_INVALID, "cnf_create: unknown activation id");
    if (c.mode < CNF_MODE_HUTCH_VJP || c.mode > CNF_MODE_EXACT)
        return fail(CNF_ERR_INVALID, "cnf_create: unknown mode");
    if (c.nprobes < 1) return fail(CNF_ERR_INVALID, "cnf_create: nprobes >= 1 required");
    if (c.kernel_path < CNF_PATH_AUTO || c.kernel_path > CNF_PATH_LAYERED)
        return fail(CNF_ERR_INVALID, "cnf_create: unknown kernel_path");
    if (c.arith < CNF_ARITH_F32 || c.arith > CNF_ARITH_BF16X6)
        return fail(CNF_ERR_INVALID, "cnf_create: unknown arith");
    if (c.arith != CNF_ARITH_F32 && (c.kernel_path == CNF_PATH_SIMT || c.kernel_path == CNF_PATH_LAYERED))
        return fail(CNF_ERR_INVALID, "cnf_create: arith = BF16X6 is an MFMA-path option");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(CNF_ERR_NO_DEVICE, "cnf_create: no HIP device visible (libcnf_hip has no CPU fallback)");
    if (c.device_id < 0 || c.device_id >= ndev)
        return fail(CNF_ERR_INVALID, "cnf_create: device_id out of range");

    cnf_handle* h = new cnf_handle();
    h->cfg = c;
    h->D = D;
    h->S = D + 3;
    NetDev& n = h->net;
    n.D = D; n.C = c.ncond; n.autonomous = c.autonomous; n.n_layers = c.n_layers;
    n.maxw = 0;
    for (int l = 0; l <= c.n_layers; ++l) {
        n.widths[l] = c.widths[l];
        if (c.widths[l] > n.maxw) n.maxw = c.widths[l];
    }
    for (int l = 0; l < c.n_layers; ++l) n.acts[l] = c.acts[l];
    n.mode = c.mode; n.K = c.nprobes; n.reg_z = c.reg_z; n.reg_j = c.reg_j;

    h->path = CNF_PATH_SIMT;
    if (c.kernel_path == CNF_PATH_LAYERED) {
        if (!layered_supports(c)) {
            delete h;
            return fail(CNF_ERR_UNSUPPORTED, "cnf_create: CNF_PATH_LAYERED covers layers of up to 512 outputs and 639 inputs");
        }
        h->path = CNF_PATH_LAYERED;
        h->layered_forced = true;
    } else if (c.kernel_path != CNF_PATH_SIMT) {
        h->plan = mfma_plan_create(c);
        if (h->plan) {
            h->path = CNF_PATH_MFMA;
        } else if (c.kernel_path == CNF_PATH_MFMA || c.arith != CNF_ARITH_F32) {
            delete h;
            return fail(CNF_ERR_UNSUPPORTED, "cnf_create: configuration not covered by the MFMA kernels");
        } else if (layered_supports(c)) {
            h->path = CNF_PATH_LAYERED;
        }
    }
    *out = h;
    return CNF_OK;
}

int cnf_destroy(cnf_handle* h) {
    if (!h) return CNF_OK;
    DeviceGuard g(h->cfg.device_id);
    if (h->par.P_dev) (void)hipFree(h->par.P_dev);
    if (h->par.packed_dev) (void)hipFree(h->par.packed_dev);
    if (h->simt.ws) (void)hipFree(h->simt.ws);
    if (h->simt.kbuf) (void)hipFree(h->simt.kbuf);
    if (h->loss_partial) (void)hipFree(h->loss_partial);
    if (h->grad.packed) (void)hipFree(h->grad.packed);
    if (h->grad.ws) (void)hipFree(h->grad.ws);
    if (h->par.stage) (void)hipFree(h->par.stage);
    if (h->emb.buf) (void)hipFree(h->emb.buf);
    if (h->emb.err_partial) (void)hipFree(h->emb.err_partial);
    if (h->vc.buf) (void)hipFree(h->vc.buf);
    if (h->grad.tgrid_dev) (void)hipFree(h->grad.tgrid_dev);
    if (h->adp.buf) (void)hipFree(h->adp.buf);
    if (h->adp.dc_buf) (void)hipFree(h->adp.dc_buf);
    if (h->vc.partial) (void)hipFree(h->vc.partial);
    layered_grad_destroy(h->grad.layered);
    free_pack_map(h->par.map_fwd);
    free_pack_map(h->grad.map);
    free_pack_map(h->grad.map_slab);
    if (h->grad.slab_packed) (void)hipFree(h->grad.slab_packed);
    if (h->grad.slab_ws) (void)hipFree(h->grad.slab_ws);
    free_pack_map(h->grad.map_cg);
    if (h->grad.cg_packed) (void)hipFree(h->grad.cg_packed);
    if (h->grad.plan_cg) mfma_plan_destroy(h->grad.plan_cg);
    if (h->plan) mfma_plan_destroy(h->plan);
    delete h;
    return CNF_OK;
}

int cnf_kernel_path(const cnf_handle* h) { return h ? h->path : CNF_ERR_INVALID; }
int cnf_kernel_family_for(cnf_handle* h, int64_t B, int whole_solve) {
    if (!h || B < 0) return CNF_ERR_INVALID;
    if (h->path == CNF_PATH_MFMA) {
        DeviceGuard g(h->cfg.device_id);
        return mfma_plan_family_for(h->plan, B, whole_solve != 0);
    }
    if (h->path == CNF_PATH_LAYERED) return (h->layered_forced || B >= layered_min_batch()) ? CNF_FAMILY_LAYERED : CNF_FAMILY_SIMT;
    return CNF_FAMILY_SIMT;
}
int cnf_kernel_family(const cnf_handle* h) {
    if (!h) return CNF_ERR_INVALID;
    if (h->path == CNF_PATH_MFMA) return mfma_plan_family_for(h->plan, 0, false);
    return h->path == CNF_PATH_LAYERED ? CNF_FAMILY_LAYERED : CNF_FAMILY_SIMT;
}
const char* cnf_kernel_name(const cnf_handle* h) {
    if (!h) return "";
    if (h->path == CNF_PATH_MFMA) return mfma_plan_name(h->plan);
    return h->path == CNF_PATH_LAYERED ? "layered" : "simt";
}
int cnf_solve_controller(const cnf_handle* h) { return (h && h->adp.last_controller >= 0) ? h->adp.last_controller : CNF_ERR_INVALID; }

int cnf_repack_on_device(const cnf_handle* h) {
    if (!h) return CNF_ERR_INVALID;
    if (!h->par.have) return CNF_ERR_NO_PARAMS;
    return h->par.repack_on_device ? 1 : 0;
}

int cnf_set_params(cnf_handle* h, const float* p, size_t n, const size_t* w_off,
                   const size_t* b_off, int p_is_device, void* stream) {
    if (!h || !p || !w_off || !b_off) return fail(CNF_ERR_INVALID, "cnf_set_params: null argument");
    const cnf_config& c = h->cfg;
    for (int l = 0; l < c.n_layers; ++l) {
        const size_t wn = (size_t)c.widths[l] * (size_t)c.widths[l + 1];
        if (w_off[l] + wn > n || b_off[l] + (size_t)c.widths[l + 1] > n)
            return fail(CNF_ERR_INVALID, "cnf_set_params: layer offsets exceed the parameter vector");
    }
    DeviceGuard g(c.device_id);
    if (!g.ok) return fail(CNF_ERR_HIP, "cnf_set_params: hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const bool mfma = h->path == CNF_PATH_MFMA;
    const cnf_config gc = api_grad_cfg(h);
    const bool want_grad = mfma && mfma_plan_is_per_wave(h->plan) && grad_supported(gc);
    const bool same_layout = h->par.have && h->par.n == n &&
                             std::equal(w_off, w_off + c.n_layers, h->par.w_off.begin()) &&
                             std::equal(b_off, b_off + c.n_layers, h->par.b_off.begin());
    if (!same_layout) {
        h->par.w_off.assign(w_off, w_off + c.n_layers);
        h->par.b_off.assign(b_off, b_off + c.n_layers);
        h->par.maps_built = false;
    }
    if (mfma && !h->par.packed_dev) HIP_TRY(hipMalloc((void**)&h->par.packed_dev, mfma_packed_bytes(h->plan)));
    if (want_grad && !h->grad.packed) HIP_TRY(hipMalloc((void**)&h->grad.packed, grad_packed_bytes(gc)));
    if (mfma && !h->par.maps_built) {
        // one-time (per layout): derive and verify the gather maps from the host packers
        size_t qo = 0, ql = 0;
        (void)mfma_plan_q_region(h->plan, &qo, &ql);
        build_pack_map(h->par.map_fwd, n, mfma_packed_bytes(h->plan) / sizeof(float),
                       [&](const float* src, float* dst) { mfma_pack(h->plan, src, w_off, b_off, dst); }, qo, ql);
        if (want_grad)
            build_pack_map(h->grad.map, n, grad_packed_bytes(gc) / sizeof(float),
                           [&](const float* src, float* dst) { grad_pack(gc, src, w_off, b_off, dst); });
        h->par.maps_built = true;
    }
    const bool dev_pack = mfma && h->par.map_fwd.valid && (!want_grad || h->grad.map.valid);
    h->par.repack_on_device = dev_pack;
    if (mfma && dev_pack) {
        // device path: (host p: one H2D copy into the staging buffer, then) gather kernels on `stream`
        const float* src = p;
        if (!p_is_device) {
            if (h->par.stage_n < n) {
                if (h->par.stage) HIP_TRY(hipFree(h->par.stage));
                h->par.stage = nullptr; h->par.stage_n = 0;
                HIP_TRY(hipMalloc((void**)&h->par.stage, n * sizeof(float)));
                h->par.stage_n = n;
            }
            HIP_TRY(hipMemcpyAsync(h->par.stage, p, n * sizeof(float), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));   // the caller may reuse its host buffer on return
            src = h->par.stage;
        }
        // Lux-layout device copy (the layer-wise gradient reads the plain parameters)
        if (h->par.P_dev && h->par.n != n) {
            HIP_TRY(hipFree(h->par.P_dev));
            h->par.P_dev = nullptr;
        }
        if (!h->par.P_dev) HIP_TRY(hipMalloc((void**)&h->par.P_dev, n * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(h->par.P_dev, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        const PackMap* maps[2] = {&h->par.map_fwd, want_grad ? &h->grad.map : nullptr};
        float* outs[2] = {h->par.packed_dev, h->grad.packed};
        for (int i = 0; i < 2; ++i) {
            if (!maps[i]) continue;
            const size_t np = maps[i]->n;
            hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, src,
                               maps[i]->idx, maps[i]->scale, outs[i], np);
            HIP_TRY(hipGetLastError());
        }
        size_t qo = 0, ql = 0;
        if (mfma_plan_q_region(h->plan, &qo, &ql)) HIP_TRY(mfma_pack_q_device(h->plan, src, w_off, h->par.packed_dev, st));
    } else {
        // host path: SIMT parameters (plain copy) and images that are not a gather (split-bf16)
        std::vector<float> host(n);
        if (p_is_device) {
            HIP_TRY(hipMemcpyAsync(host.data(), p, n * sizeof(float), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        } else {
            std::memcpy(host.data(), p, n * sizeof(float));
        }
        if (mfma) {
            if (h->par.P_dev && h->par.n != n) {
                HIP_TRY(hipFree(h->par.P_dev));
                h->par.P_dev = nullptr;
            }
            if (!h->par.P_dev) HIP_TRY(hipMalloc((void**)&h->par.P_dev, n * sizeof(float)));
            HIP_TRY(hipMemcpyAsync(h->par.P_dev, host.data(), n * sizeof(float), hipMemcpyHostToDevice, st));
            const size_t bytes = mfma_packed_bytes(h->plan);
            std::vector<float> packed(bytes / sizeof(float), 0.f);
            mfma_pack(h->plan, host.data(), w_off, b_off, packed.data());
            HIP_TRY(hipMemcpyAsync(h->par.packed_dev, packed.data(), bytes, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (want_grad) {
                const size_t gb = grad_packed_bytes(gc);
                std::vector<float> gp(gb / sizeof(float), 0.f);
                grad_pack(gc, host.data(), w_off, b_off, gp.data());
                HIP_TRY(hipMemcpyAsync(h->grad.packed, gp.data(), gb, hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
        } else {
            if (h->par.P_dev && h->par.n != n) {
                HIP_TRY(hipFree(h->par.P_dev));
                h->par.P_dev = nullptr;
            }
            if (!h->par.P_dev) HIP_TRY(hipMalloc((void**)&h->par.P_dev, n * sizeof(float)));
            HIP_TRY(hipMemcpyAsync(h->par.P_dev, host.data(), n * sizeof(float), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (int l = 0; l < c.n_layers; ++l) {
                h->net.w_off[l] = (int)w_off[l];
                h->net.b_off[l] = (int)b_off[l];
            }
        }
    }
    // image of the slab-accumulator gradient kernel: a gather from the Lux-layout device copy kept above
    if (grad_slab_supported(c) && !(mfma && want_grad)) {
        const size_t sb = grad_slab_packed_bytes(c);
        if (!h->grad.slab_packed) HIP_TRY(hipMalloc((void**)&h->grad.slab_packed, sb));
        if (!h->grad.map_slab.valid || !same_layout)
            build_pack_map(h->grad.map_slab, n, sb / sizeof(float),
                           [&](const float* src, float* dst) { grad_slab_pack(c, src, w_off, b_off, dst); });
        if (h->grad.map_slab.valid) {
            const size_t np = h->grad.map_slab.n;
            hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, h->par.P_dev,
                               h->grad.map_slab.idx, h->grad.map_slab.scale, h->grad.slab_packed, np);
            HIP_TRY(hipGetLastError());
        } else {
            HIP_TRY(hipFree(h->grad.slab_packed));
            h->grad.slab_packed = nullptr;
        }
    }
    // image of the auxiliary cooperative plan (see cnf_handle::plan_cg): wide two-layer slab shapes, one-probe VJP, no conditions
    if (grad_slab_supported(c) && !(mfma && want_grad) && c.mode == CNF_MODE_HUTCH_VJP && c.nprobes == 1 && c.ncond == 0 &&
        c.widths[1] > 96 && c.widths[1] == c.widths[2] && c.widths[1] % 4 == 0) {
        if (!h->grad.cg_tried) {
            h->grad.cg_tried = true;
            const char* e = getenv("CNF_COOP_GRAD_MID");
            if (!(e && *e == '0')) h->grad.plan_cg = mfma_plan_create(c, true);
        }
        if (h->grad.plan_cg) {
            const size_t cb = mfma_packed_bytes(h->grad.plan_cg);
            if (!h->grad.cg_packed) HIP_TRY(hipMalloc((void**)&h->grad.cg_packed, cb));
            if (!h->grad.map_cg.valid || !same_layout)
                build_pack_map(h->grad.map_cg, n, cb / sizeof(float),
                               [&](const float* src, float* dst) { mfma_pack(h->grad.plan_cg, src, w_off, b_off, dst); });
            if (h->grad.map_cg.valid) {
                const size_t np = h->grad.map_cg.n;
                hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, h->par.P_dev,
                                   h->grad.map_cg.idx, h->grad.map_cg.scale, h->grad.cg_packed, np);
                HIP_TRY(hipGetLastError());
            } else {
                HIP_TRY(hipFree(h->grad.cg_packed));
                h->grad.cg_packed = nullptr;
            }
        }
    }
    h->par.n = n;
    h->par.have = true;
    return CNF_OK;
}

// One dynamics evaluation on the generic families: layer-wise GEMMs (cnf_layered.hip) for a LAYERED handle
// (CNF_LAYERED_MIN_B=n sends batches below n to the SIMT kernels instead), the thread-per-sample kernels
// (cnf_simt.hip) otherwise.
static int generic_aug_f(cnf_handle* h, const StageIn& in, float t, const float* eps, const float* ys, int64_t B,
                         float* du, bool first_of_solve, hipStream_t st) {
    if (h->path == CNF_PATH_LAYERED && (h->layered_forced || B >= layered_min_batch())) {
        std::string msg;
        hipError_t e = layered_aug_f(&h->grad.layered, h->cfg, h->par.P_dev, h->par.w_off.data(), h->par.b_off.data(), first_of_solve, in, t,
                                     eps, ys, B, du, st, &msg);
        if (e == hipErrorNotSupported) return fail(CNF_ERR_UNSUPPORTED, msg);
        if (e != hipSuccess) return fail(CNF_ERR_HIP, msg);
        return CNF_OK;
    }
    int rc = ensure_ws(h, B);
    if (rc) return rc;
    HIP_TRY(simt_aug_f(h->net, h->par.P_dev, in, t, eps, ys, B, du, h->simt.ws, h->simt.ws_B, st));
    return CNF_OK;
}

extern "C++" {
int cnf::api_check_call(cnf_handle* h, const float* eps, const float* ys, int64_t B, const char* who) {
    if (!h) return fail(CNF_ERR_INVALID, std::string(who) + ": null handle");
    if (!h->par.have) return fail(CNF_ERR_NO_PARAMS, std::string(who) + ": cnf_set_params not called");
    if (B < 0) return fail(CNF_ERR_INVALID, std::string(who) + ": negative batch");
    if (h->cfg.mode != CNF_MODE_EXACT && !eps && B > 0)
        return fail(CNF_ERR_INVALID, std::string(who) + ": eps is required in Hutchinson modes");
    if (h->cfg.ncond > 0 && !ys && B > 0)
        return fail(CNF_ERR_INVALID, std::string(who) + ": ys is required when ncond > 0");
    return CNF_OK;
}
}  // extern "C++"

int cnf_aug_f(cnf_handle* h, float* du, const float* u, float t, const float* eps,
              const float* ys, int64_t B, void* stream) {
    int rc = api_check_call(h, eps, ys, B, "cnf_aug_f");
    if (rc) return rc;
    if (B == 0) return CNF_OK;
    if (!du || !u) return fail(CNF_ERR_INVALID, "cnf_aug_f: null u/du");
    if (du == u) return fail(CNF_ERR_INVALID, "cnf_aug_f: du may not alias u");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    if (h->path == CNF_PATH_MFMA) {
        SolveArgs a{};
        a.u0 = u; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = 0; a.alg = 0; a.t0 = t; a.t1 = t;
        a.u_out = du; a.nvars = h->cfg.nvars; a.reg_aug = 0;
        HIP_TRY(mfma_solve(h->plan, h->par.packed_dev, a, st));
        return CNF_OK;
    }
    StageIn in{};
    in.u = u; in.nprev = 0; in.dt = 0.f;
    return generic_aug_f(h, in, t, eps, ys, B, du, true, st);
}

static int simt_integrate(cnf_handle* h, int alg, int nsteps, float t0, float t1, float* u,
                          const float* eps, const float* ys, int64_t B, hipStream_t st) {
    // u is integrated in place.  Unfused structure: one evaluation per stage + one update per step.
    int rc = ensure_kbuf(h, B);
    if (rc) return rc;
    const Tableau T = make_tableau(alg);
    const size_t n = (size_t)h->S * (size_t)B;
    float* k[6];
    for (int i = 0; i < 6; ++i) k[i] = h->simt.kbuf + (size_t)i * n;
    const float dt = (t1 - t0) / (float)nsteps;
    for (int step = 0; step < nsteps; ++step) {
        const float tn = t0 + (float)step * dt;
        for (int i = 0; i < T.ns; ++i) {
            StageIn in{};
            in.u = u; in.nprev = i; in.dt = dt;
            for (int j = 0; j < i; ++j) { in.k[j] = k[j]; in.coef[j] = T.a[i][j]; }
            rc = generic_aug_f(h, in, tn + T.c[i] * dt, eps, ys, B, k[i], step == 0 && i == 0, st);
            if (rc) return rc;
        }
        StageIn fin{};
        fin.u = u; fin.nprev = T.ns; fin.dt = dt;
        for (int j = 0; j < T.ns; ++j) { fin.k[j] = k[j]; fin.coef[j] = T.b[j]; }
        HIP_TRY(rk_update(u, fin, (int64_t)n, st));
    }
    return CNF_OK;
}

int cnf_integrate_fixed(cnf_handle* h, int alg, int nsteps, float t0, float t1, const float* u0,
                        const float* eps, const float* ys, int64_t B, float* u1, void* stream) {
    int rc = api_check_call(h, eps, ys, B, "cnf_integrate_fixed");
    if (rc) return rc;
    if (nsteps < 1) return fail(CNF_ERR_INVALID, "cnf_integrate_fixed: nsteps >= 1 required");
    if (alg != CNF_ALG_RK4 && alg != CNF_ALG_TSIT5)
        return fail(CNF_ERR_INVALID, "cnf_integrate_fixed: unknown alg");
    if (B == 0) return CNF_OK;
    if (!u0 || !u1) return fail(CNF_ERR_INVALID, "cnf_integrate_fixed: null u0/u1");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    if (h->path == CNF_PATH_MFMA) {
        SolveArgs a{};
        a.u0 = u0; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = nsteps; a.alg = alg;
        a.t0 = t0; a.t1 = t1; a.u_out = u1; a.nvars = h->cfg.nvars; a.reg_aug = 0;
        HIP_TRY(mfma_solve(h->plan, h->par.packed_dev, a, st));
        return CNF_OK;
    }
    if (u1 != u0)
        HIP_TRY(copy_async(u1, u0, (size_t)h->S * (size_t)B * sizeof(float), st));
    return simt_integrate(h, alg, nsteps, t0, t1, u1, eps, ys, B, st);
}

int cnf_inference_fixed(cnf_handle* h, int alg, int nsteps, float t0, float t1, const float* x,
                        const float* eps, const float* ys, int64_t B, float* logp, float* regs,
                        float* u_final, void* stream) {
    int rc = api_check_call(h, eps, ys, B, "cnf_inference_fixed");
    if (rc) return rc;
    if (nsteps < 1) return fail(CNF_ERR_INVALID, "cnf_inference_fixed: nsteps >= 1 required");
    if (alg != CNF_ALG_RK4 && alg != CNF_ALG_TSIT5)
        return fail(CNF_ERR_INVALID, "cnf_inference_fixed: unknown alg");
    if (B == 0) return CNF_OK;
    if (!x || !logp) return fail(CNF_ERR_INVALID, "cnf_inference_fixed: null x/logp");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    const int reg_aug = (h->cfg.reg_aug && h->cfg.naug > 0 && h->cfg.mode != CNF_MODE_EXACT) ? 1 : 0;
    if (h->path == CNF_PATH_MFMA) {
        SolveArgs a{};
        a.x = x; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = nsteps; a.alg = alg;
        a.t0 = t0; a.t1 = t1; a.u_out = u_final; a.logp = logp; a.regs = regs;
        a.nvars = h->cfg.nvars; a.reg_aug = reg_aug;
        HIP_TRY(mfma_solve(h->plan, h->par.packed_dev, a, st));
        return CNF_OK;
    }
    rc = ensure_kbuf(h, B);
    if (rc) return rc;
    float* u = u_final ? u_final : h->simt.kbuf + 6 * (size_t)h->S * (size_t)h->simt.kbuf_B;
    HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, u, st));
    rc = simt_integrate(h, alg, nsteps, t0, t1, u, eps, ys, B, st);
    if (rc) return rc;
    HIP_TRY(epilogue(u, h->cfg.nvars, h->D, reg_aug, B, logp, regs, st));
    return CNF_OK;
}

// OrdinaryDiffEq's fixed-dt stepping on (t0, t1): n_full steps of |dt|, then a shorter tail step that lands on t1 (a tstop);
// a remainder within 100 eps(Float32) of the larger end point counts as having reached it (fixed_t_for_floatingpoint_error!).
struct FixedDtPlan { int n_full; bool tail; float t_mid; };
static FixedDtPlan fixed_dt_plan(float t0, float t1, float dt) {
    const double span = std::fabs((double)t1 - (double)t0), adt = std::fabs((double)dt);
    const double tdir = t1 >= t0 ? 1.0 : -1.0;
    FixedDtPlan p{};
    p.n_full = (int)std::floor(span / adt + 1e-9);
    const double tol = 100.0 * 1.1920928955078125e-7 * std::fmax(std::fabs((double)t0), std::fabs((double)t1));
    // a Float32 dt a hair above span / n (0.1f on (0, 1)) leaves a "remainder" of one whole step less that hair: the stepper
    // snaps that step onto t1 (same tolerance), so it is the last of n_full + 1 equal steps, not a tail
    if (span - p.n_full * adt > tol && adt - (span - p.n_full * adt) <= tol) ++p.n_full;
    p.tail = span - p.n_full * adt > tol;
    p.t_mid = p.tail ? (float)((double)t0 + tdir * p.n_full * adt) : t1;
    return p;
}

int cnf_integrate_fixed_dt(cnf_handle* h, int alg, float dt, float t0, float t1, const float* u0, const float* eps,
                           const float* ys, int64_t B, float* u1, void* stream) {
    int rc = api_check_call(h, eps, ys, B, "cnf_integrate_fixed_dt");
    if (rc) return rc;
    if (!(std::fabs(dt) > 0.f) || !std::isfinite(dt) || !std::isfinite(t0) || !std::isfinite(t1))
        return fail(CNF_ERR_INVALID, "cnf_integrate_fixed_dt: dt must be non-zero and finite, t0 / t1 finite");
    if (std::fabs((double)t1 - t0) / std::fabs((double)dt) > 1e8) return fail(CNF_ERR_INVALID, "cnf_integrate_fixed_dt: more than 1e8 steps");
    if (B == 0) return CNF_OK;
    if (!u0 || !u1) return fail(CNF_ERR_INVALID, "cnf_integrate_fixed_dt: null u0/u1");
    const FixedDtPlan p = fixed_dt_plan(t0, t1, dt);
    if (p.n_full == 0 && !p.tail) {       // nothing to integrate
        DeviceGuard g(h->cfg.device_id);
        if (u1 != u0) HIP_TRY(copy_async(u1, u0, (size_t)h->S * (size_t)B * sizeof(float), (hipStream_t)stream));
        return CNF_OK;
    }
    const float* from = u0;
    if (p.n_full > 0) {
        rc = cnf_integrate_fixed(h, alg, p.n_full, t0, p.t_mid, u0, eps, ys, B, u1, stream);
        if (rc) return rc;
        from = u1;
    }
    if (p.tail) return cnf_integrate_fixed(h, alg, 1, p.t_mid, t1, from, eps, ys, B, u1, stream);
    return CNF_OK;
}

int cnf_inference_fixed_dt(cnf_handle* h, int alg, float dt, float t0, float t1, const float* x, const float* eps,
                           const float* ys, int64_t B, float* logp, float* regs, float* u_final, void* stream) {
    int rc = api_check_call(h, eps, ys, B, "cnf_inference_fixed_dt");
    if (rc) return rc;
    if (!(std::fabs(dt) > 0.f) || !std::isfinite(dt) || !std::isfinite(t0) || !std::isfinite(t1))
        return fail(CNF_ERR_INVALID, "cnf_inference_fixed_dt: dt must be non-zero and finite, t0 / t1 finite");
    if (std::fabs((double)t1 - t0) / std::fabs((double)dt) > 1e8) return fail(CNF_ERR_INVALID, "cnf_inference_fixed_dt: more than 1e8 steps");
    if (B == 0) return CNF_OK;
    if (!x || !logp) return fail(CNF_ERR_INVALID, "cnf_inference_fixed_dt: null x/logp");
    const FixedDtPlan p = fixed_dt_plan(t0, t1, dt);
    if (!p.tail && p.n_full > 0) return cnf_inference_fixed(h, alg, p.n_full, t0, t1, x, eps, ys, B, logp, regs, u_final, stream);
    // a state buffer between the two launches (or for the degenerate empty span): the caller's u_final, else the embedded-step scratch
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    float* u = u_final;
    if (!u) {
        rc = api_ensure_adaptive_buf(h, B);
        if (rc) return rc;
        u = h->adp.buf;
    }
    HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, u, st));
    if (p.n_full > 0) {
        rc = cnf_integrate_fixed(h, alg, p.n_full, t0, p.t_mid, u, eps, ys, B, u, stream);
        if (rc) return rc;
    }
    if (p.tail) {
        rc = cnf_integrate_fixed(h, alg, 1, p.t_mid, t1, u, eps, ys, B, u, stream);
        if (rc) return rc;
    }
    return cnf_epilogue(h, u, B, logp, regs, stream);
}

// f(u + dt sum coef k, t) on whichever family serves the handle; `stage` is scratch for the fused path, whose
// single-call kernel takes the stage state itself
extern "C++" {
int cnf::api_eval_dynamics(cnf_handle* h, const StageIn& in, float t, const float* eps, const float* ys, int64_t B, float* du,
                           float* stage, bool first, hipStream_t st) {
    if (h->path == CNF_PATH_MFMA) {
        const float* uin = in.u;
        if (in.nprev > 0) {
            HIP_TRY(rk_update(stage, in, (int64_t)h->S * B, st));
            uin = stage;
        }
        SolveArgs a{};
        a.u0 = uin; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = 0; a.alg = 0; a.t0 = t; a.t1 = t;
        a.u_out = du; a.nvars = h->cfg.nvars; a.reg_aug = 0;
        HIP_TRY(mfma_solve(h->plan, h->par.packed_dev, a, st));
        return CNF_OK;
    }
    return generic_aug_f(h, in, t, eps, ys, B, du, first, st);
}
}  // extern "C++"

extern "C++" {
int cnf::api_integrate_grid(cnf_handle* h, int alg, int nsteps, const float* tgrid, float* u, const float* eps, const float* ys,
                            int64_t B, hipStream_t st) {
    const size_t n = (size_t)h->S * (size_t)B;
    if (B > h->emb.B) {
        if (h->emb.buf) HIP_TRY(hipFree(h->emb.buf));
        h->emb.buf = nullptr; h->emb.B = 0;
        HIP_TRY(hipMalloc((void**)&h->emb.buf, 8 * n * sizeof(float)));
        h->emb.B = B;
    }
    const size_t slot = (size_t)h->S * (size_t)h->emb.B;
    float* stage = h->emb.buf + 7 * slot;
    float* k[6];
    for (int i = 0; i < 6; ++i) k[i] = h->emb.buf + (size_t)i * slot;
    const Tableau T = make_tableau(alg);
    for (int s = 0; s < nsteps; ++s) {
        const float tn = tgrid[s], dt = tgrid[s + 1] - tgrid[s];
        for (int i = 0; i < T.ns; ++i) {
            StageIn in{};
            in.u = u; in.nprev = i; in.dt = dt;
            for (int j = 0; j < i; ++j) { in.k[j] = k[j]; in.coef[j] = T.a[i][j]; }
            int rc = api_eval_dynamics(h, in, tn + T.c[i] * dt, eps, ys, B, k[i], stage, s == 0 && i == 0, st);
            if (rc) return rc;
        }
        StageIn fin{};
        fin.u = u; fin.nprev = T.ns; fin.dt = dt;
        for (int j = 0; j < T.ns; ++j) { fin.k[j] = k[j]; fin.coef[j] = T.b[j]; }
        HIP_TRY(rk_update(u, fin, (int64_t)n, st));
    }
    return CNF_OK;
}
}  // extern "C++"

int cnf_assemble_u0(cnf_handle* h, const float* x, int64_t B, float* u0, void* stream) {
    if (!h || B < 0) return fail(CNF_ERR_INVALID, "cnf_assemble_u0: null handle or negative batch");
    if (B == 0) return CNF_OK;
    if (!x || !u0) return fail(CNF_ERR_INVALID, "cnf_assemble_u0: null x/u0");
    DeviceGuard g(h->cfg.device_id);
    HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, u0, (hipStream_t)stream));
    return CNF_OK;
}

int cnf_epilogue(cnf_handle* h, const float* u, int64_t B, float* logp, float* regs, void* stream) {
    if (!h || B < 0) return fail(CNF_ERR_INVALID, "cnf_epilogue: null handle or negative batch");
    if (B == 0) return CNF_OK;
    if (!u || !logp) return fail(CNF_ERR_INVALID, "cnf_epilogue: null u/logp");
    DeviceGuard g(h->cfg.device_id);
    const int reg_aug = (h->cfg.reg_aug && h->cfg.naug > 0 && h->cfg.mode != CNF_MODE_EXACT) ? 1 : 0;
    HIP_TRY(epilogue(u, h->cfg.nvars, h->D, reg_aug, B, logp, regs, (hipStream_t)stream));
    return CNF_OK;
}

int cnf_loss_sums(cnf_handle* h, const float* logp, const float* regs, int64_t B, float* sums4,
                  void* stream) {
    if (!h || !logp || !sums4) return fail(CNF_ERR_INVALID, "cnf_loss_sums: null argument");
    if (B < 0) return fail(CNF_ERR_INVALID, "cnf_loss_sums: negative batch");
    DeviceGuard g(h->cfg.device_id);
    if (!h->loss_partial) HIP_TRY(hipMalloc((void**)&h->loss_partial, 256 * 4 * sizeof(float)));
    HIP_TRY(loss_sums(logp, regs, B, h->loss_partial, sums4, (hipStream_t)stream));
    return CNF_OK;
}

int cnf_loss_mean(cnf_handle* h, const float* logp, const float* regs, int64_t B, const double* lambdas, float* sums4,
                  float* loss, void* stream) {
    if (!h || !logp || !lambdas || !loss) return fail(CNF_ERR_INVALID, "cnf_loss_mean: null argument");
    if (B < 1) return fail(CNF_ERR_INVALID, "cnf_loss_mean: the mean of an empty batch is undefined (use cnf_loss_sums on shards)");
    DeviceGuard g(h->cfg.device_id);
    if (!h->loss_partial) HIP_TRY(hipMalloc((void**)&h->loss_partial, 256 * 4 * sizeof(float)));
    HIP_TRY(loss_mean(logp, regs, B, h->loss_partial, sums4, loss, lambdas, (hipStream_t)stream));
    return CNF_OK;
}

}  // extern "C"
