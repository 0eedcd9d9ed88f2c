// cnf_api.hip — the C ABI of libcnf_hip.so (see include/cnf.h for the contract).
//
// Host-side only: handle lifetime, validation, parameter repacking, workspace management and
// dispatch to the kernel families (cnf_mfma.hip: fused whole-solve MFMA kernels;
// cnf_simt.hip: generic per-call kernels).  There is no CPU fallback: without a gfx950 device
// every entry point that would compute returns CNF_ERR_NO_DEVICE / CNF_ERR_HIP.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "cnf_internal.h"

using namespace cnf;

namespace {
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return fail(CNF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
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
}  // namespace

struct cnf_handle {
    cnf_config cfg{};
    int D = 0, S = 0;
    NetDev net{};
    size_t nparams = 0;
    bool have_params = false;
    int path = CNF_PATH_SIMT;
    // device copies of the parameters
    float* P_dev = nullptr;       // Lux layout (SIMT path)
    MfmaPlan* plan = nullptr;
    float* packed_dev = nullptr;  // MFMA operand image
    // SIMT workspaces, grown on demand
    float* ws = nullptr;
    int64_t ws_B = 0;
    float* kbuf = nullptr;        // 6 stage derivatives + 1 state, each S x kbuf_B
    int64_t kbuf_B = 0;
    float* loss_partial = nullptr;
    // parameter gradient (cnf_loss_grad_fixed)
    std::vector<size_t> w_off, b_off;   // Lux offsets given to cnf_set_params
    float* grad_packed = nullptr;        // plain f32 operand image for the reverse sweep
    float* grad_ws = nullptr;            // checkpoints + logp + regs
    size_t grad_ws_bytes = 0;
    int num_cus = 0;
    // device-side repacking (see PackMap): maps for the solve image and the gradient image, rebuilt
    // when the layout handed to cnf_set_params changes; p_stage holds host-supplied parameters
    PackMap map_fwd, map_grad, map_slab;
    float* slab_packed = nullptr;        // operand image of the slab-accumulator gradient kernel (cnf_grad_slab.hip)
    float* slab_ws = nullptr;            // its checkpoints + slabs
    size_t slab_ws_floats = 0;
    LayeredGrad* layered = nullptr;      // rocBLAS context + workspaces of the layer-wise evaluation / gradient
    // embedded-step workspace (cnf_step_embedded): 7 stage derivatives + 1 stage state, each S x ebuf_B
    float* ebuf = nullptr;
    int64_t ebuf_B = 0;
    int ek[7] = {0, 1, 2, 3, 4, 5, 6};   // which slot holds k_1 .. k_7 (first-same-as-last swaps slots 0 and 6)
    double* err_partial = nullptr;
    // multistep solve (cnf_vcabm_*): 6 state-size vectors + 2 x kVcSlots difference vectors, each S x vc_B
    float* vc_buf = nullptr;
    double* vc_partial = nullptr;
    int64_t vc_B = -1;
    int vc_iu = 0, vc_iun = 2, vc_if = 3, vc_ifn = 5, vc_cur = 0;   // which vector holds u, u_new, f_n, f_{n+1}; live difference half
    int vc_nhist = 0, vc_k = 0;          // accepted steps since begin; order of the pending attempt (0 = none)
    int vc_avail = 0, vc_m = 0;          // differences Phi*_j(n-1) the last accepted step stored; those the pending attempt stores
    double vc_hist[kVcSlots + 1] = {};   // signed sizes of the accepted steps, newest first
    double vc_t = 0.0, vc_dt = 0.0;
    float* ad_buf = nullptr;             // adaptive Tsit5 whole solve (cnf_solve_tsit5): two states + two derivative scratch vectors
    int64_t ad_B = 0;
    float* tgrid_dev = nullptr;          // step times of a non-uniform grid for the fused gradient kernels
    size_t tgrid_cap = 0;
    bool layered_forced = false;         // kernel_path = CNF_PATH_LAYERED given explicitly: GEMM path for every batch
    bool maps_built = false;
    bool repack_on_device = false;
    float* p_stage = nullptr;
    size_t p_stage_n = 0;
};

// The configuration the fused gradient kernels are selected and packed for.  TestMode (exact trace): -tr J is the sum over
// the D unit vectors e_k of -e_k^T J e_k, i.e. the several-probe reverse sweep with K = D one-hot probes of weight 1 and no
// regularisers (up to the kernel's probe capacity; wider states take the layer-wise path).
static cnf_config grad_cfg(const cnf_handle* h) {
    cnf_config c = h->cfg;
    if (c.mode == CNF_MODE_EXACT) {
        c.mode = CNF_MODE_HUTCH_VJP;
        c.nprobes = h->D;
        c.reg_z = c.reg_j = c.reg_aug = 0;
        if (h->D > 8) c.nprobes = 0;   // no fused instance: grad_supported() rejects nprobes < 1
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

static int ensure_ws(cnf_handle* h, int64_t B) {
    if (B <= h->ws_B) return CNF_OK;
    if (h->ws) HIP_TRY(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_B = 0;
    const int64_t Bp = (B + 255) / 256 * 256;
    HIP_TRY(hipMalloc((void**)&h->ws, simt_ws_rows(h->net) * (size_t)Bp * sizeof(float)));
    h->ws_B = Bp;
    return CNF_OK;
}

static int ensure_kbuf(cnf_handle* h, int64_t B) {
    if (B <= h->kbuf_B) return CNF_OK;
    if (h->kbuf) HIP_TRY(hipFree(h->kbuf));
    h->kbuf = nullptr;
    h->kbuf_B = 0;
    HIP_TRY(hipMalloc((void**)&h->kbuf, 7 * (size_t)h->S * (size_t)B * sizeof(float)));
    h->kbuf_B = B;
    return CNF_OK;
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
        if (!layered_available()) {
            delete h;
            return fail(CNF_ERR_UNSUPPORTED, "cnf_create: CNF_PATH_LAYERED needs librocblas.so.5 (dlopen failed)");
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
        } else if (layered_available()) {
            h->path = CNF_PATH_LAYERED;
        }
    }
    *out = h;
    return CNF_OK;
}

int cnf_destroy(cnf_handle* h) {
    if (!h) return CNF_OK;
    DeviceGuard g(h->cfg.device_id);
    if (h->P_dev) (void)hipFree(h->P_dev);
    if (h->packed_dev) (void)hipFree(h->packed_dev);
    if (h->ws) (void)hipFree(h->ws);
    if (h->kbuf) (void)hipFree(h->kbuf);
    if (h->loss_partial) (void)hipFree(h->loss_partial);
    if (h->grad_packed) (void)hipFree(h->grad_packed);
    if (h->grad_ws) (void)hipFree(h->grad_ws);
    if (h->p_stage) (void)hipFree(h->p_stage);
    if (h->ebuf) (void)hipFree(h->ebuf);
    if (h->err_partial) (void)hipFree(h->err_partial);
    if (h->vc_buf) (void)hipFree(h->vc_buf);
    if (h->tgrid_dev) (void)hipFree(h->tgrid_dev);
    if (h->ad_buf) (void)hipFree(h->ad_buf);
    if (h->vc_partial) (void)hipFree(h->vc_partial);
    layered_grad_destroy(h->layered);
    free_pack_map(h->map_fwd);
    free_pack_map(h->map_grad);
    free_pack_map(h->map_slab);
    if (h->slab_packed) (void)hipFree(h->slab_packed);
    if (h->slab_ws) (void)hipFree(h->slab_ws);
    if (h->plan) mfma_plan_destroy(h->plan);
    delete h;
    return CNF_OK;
}

int cnf_kernel_path(const cnf_handle* h) { return h ? h->path : CNF_ERR_INVALID; }

int cnf_repack_on_device(const cnf_handle* h) {
    if (!h) return CNF_ERR_INVALID;
    if (!h->have_params) return CNF_ERR_NO_PARAMS;
    return h->repack_on_device ? 1 : 0;
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
    const cnf_config gc = grad_cfg(h);
    const bool want_grad = mfma && mfma_plan_is_per_wave(h->plan) && grad_supported(gc);
    const bool same_layout = h->have_params && h->nparams == n &&
                             std::equal(w_off, w_off + c.n_layers, h->w_off.begin()) &&
                             std::equal(b_off, b_off + c.n_layers, h->b_off.begin());
    if (!same_layout) {
        h->w_off.assign(w_off, w_off + c.n_layers);
        h->b_off.assign(b_off, b_off + c.n_layers);
        h->maps_built = false;
    }
    if (mfma && !h->packed_dev) HIP_TRY(hipMalloc((void**)&h->packed_dev, mfma_packed_bytes(h->plan)));
    if (want_grad && !h->grad_packed) HIP_TRY(hipMalloc((void**)&h->grad_packed, grad_packed_bytes(gc)));
    if (mfma && !h->maps_built) {
        // one-time (per layout): derive and verify the gather maps from the host packers
        size_t qo = 0, ql = 0;
        (void)mfma_plan_q_region(h->plan, &qo, &ql);
        build_pack_map(h->map_fwd, n, mfma_packed_bytes(h->plan) / sizeof(float),
                       [&](const float* src, float* dst) { mfma_pack(h->plan, src, w_off, b_off, dst); }, qo, ql);
        if (want_grad)
            build_pack_map(h->map_grad, n, grad_packed_bytes(gc) / sizeof(float),
                           [&](const float* src, float* dst) { grad_pack(gc, src, w_off, b_off, dst); });
        h->maps_built = true;
    }
    const bool dev_pack = mfma && h->map_fwd.valid && (!want_grad || h->map_grad.valid);
    h->repack_on_device = dev_pack;
    if (mfma && dev_pack) {
        // device path: (host p: one H2D copy into the staging buffer, then) gather kernels on `stream`
        const float* src = p;
        if (!p_is_device) {
            if (h->p_stage_n < n) {
                if (h->p_stage) HIP_TRY(hipFree(h->p_stage));
                h->p_stage = nullptr; h->p_stage_n = 0;
                HIP_TRY(hipMalloc((void**)&h->p_stage, n * sizeof(float)));
                h->p_stage_n = n;
            }
            HIP_TRY(hipMemcpyAsync(h->p_stage, p, n * sizeof(float), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));   // the caller may reuse its host buffer on return
            src = h->p_stage;
        }
        // Lux-layout device copy (the layer-wise gradient reads the plain parameters)
        if (h->P_dev && h->nparams != n) {
            HIP_TRY(hipFree(h->P_dev));
            h->P_dev = nullptr;
        }
        if (!h->P_dev) HIP_TRY(hipMalloc((void**)&h->P_dev, n * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(h->P_dev, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        const PackMap* maps[2] = {&h->map_fwd, want_grad ? &h->map_grad : nullptr};
        float* outs[2] = {h->packed_dev, h->grad_packed};
        for (int i = 0; i < 2; ++i) {
            if (!maps[i]) continue;
            const size_t np = maps[i]->n;
            hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, src,
                               maps[i]->idx, maps[i]->scale, outs[i], np);
            HIP_TRY(hipGetLastError());
        }
        size_t qo = 0, ql = 0;
        if (mfma_plan_q_region(h->plan, &qo, &ql)) HIP_TRY(mfma_pack_q_device(h->plan, src, w_off, h->packed_dev, st));
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
            if (h->P_dev && h->nparams != n) {
                HIP_TRY(hipFree(h->P_dev));
                h->P_dev = nullptr;
            }
            if (!h->P_dev) HIP_TRY(hipMalloc((void**)&h->P_dev, n * sizeof(float)));
            HIP_TRY(hipMemcpyAsync(h->P_dev, host.data(), n * sizeof(float), hipMemcpyHostToDevice, st));
            const size_t bytes = mfma_packed_bytes(h->plan);
            std::vector<float> packed(bytes / sizeof(float), 0.f);
            mfma_pack(h->plan, host.data(), w_off, b_off, packed.data());
            HIP_TRY(hipMemcpyAsync(h->packed_dev, packed.data(), bytes, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (want_grad) {
                const size_t gb = grad_packed_bytes(gc);
                std::vector<float> gp(gb / sizeof(float), 0.f);
                grad_pack(gc, host.data(), w_off, b_off, gp.data());
                HIP_TRY(hipMemcpyAsync(h->grad_packed, gp.data(), gb, hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
        } else {
            if (h->P_dev && h->nparams != n) {
                HIP_TRY(hipFree(h->P_dev));
                h->P_dev = nullptr;
            }
            if (!h->P_dev) HIP_TRY(hipMalloc((void**)&h->P_dev, n * sizeof(float)));
            HIP_TRY(hipMemcpyAsync(h->P_dev, host.data(), n * sizeof(float), hipMemcpyHostToDevice, st));
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
        if (!h->slab_packed) HIP_TRY(hipMalloc((void**)&h->slab_packed, sb));
        if (!h->map_slab.valid || !same_layout)
            build_pack_map(h->map_slab, n, sb / sizeof(float),
                           [&](const float* src, float* dst) { grad_slab_pack(c, src, w_off, b_off, dst); });
        if (h->map_slab.valid) {
            const size_t np = h->map_slab.n;
            hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, h->P_dev,
                               h->map_slab.idx, h->map_slab.scale, h->slab_packed, np);
            HIP_TRY(hipGetLastError());
        } else {
            HIP_TRY(hipFree(h->slab_packed));
            h->slab_packed = nullptr;
        }
    }
    h->nparams = n;
    h->have_params = true;
    return CNF_OK;
}

// One dynamics evaluation on the generic families: layer-wise GEMMs (cnf_layered.hip) for a LAYERED handle
// (CNF_LAYERED_MIN_B=n sends batches below n to the SIMT kernels instead), the thread-per-sample kernels
// (cnf_simt.hip) otherwise.
static int64_t layered_min_batch() {
    const char* e = getenv("CNF_LAYERED_MIN_B");
    return (e && *e) ? atoll(e) : 0;   // measured: the GEMM path wins at every batch size (profiles/r1h_generic_small.json)
}
static int generic_aug_f(cnf_handle* h, const StageIn& in, float t, const float* eps, const float* ys, int64_t B,
                         float* du, bool first_of_solve, hipStream_t st) {
    if (h->path == CNF_PATH_LAYERED && (h->layered_forced || B >= layered_min_batch())) {
        std::string msg;
        hipError_t e = layered_aug_f(&h->layered, h->cfg, h->P_dev, h->w_off.data(), h->b_off.data(), first_of_solve, in, t,
                                     eps, ys, B, du, st, &msg);
        if (e == hipErrorNotSupported) return fail(CNF_ERR_UNSUPPORTED, msg);
        if (e != hipSuccess) return fail(CNF_ERR_HIP, msg);
        return CNF_OK;
    }
    int rc = ensure_ws(h, B);
    if (rc) return rc;
    HIP_TRY(simt_aug_f(h->net, h->P_dev, in, t, eps, ys, B, du, h->ws, h->ws_B, st));
    return CNF_OK;
}

static int check_call(cnf_handle* h, const float* eps, const float* ys, int64_t B, const char* who) {
    if (!h) return fail(CNF_ERR_INVALID, std::string(who) + ": null handle");
    if (!h->have_params) return fail(CNF_ERR_NO_PARAMS, std::string(who) + ": cnf_set_params not called");
    if (B < 0) return fail(CNF_ERR_INVALID, std::string(who) + ": negative batch");
    if (h->cfg.mode != CNF_MODE_EXACT && !eps && B > 0)
        return fail(CNF_ERR_INVALID, std::string(who) + ": eps is required in Hutchinson modes");
    if (h->cfg.ncond > 0 && !ys && B > 0)
        return fail(CNF_ERR_INVALID, std::string(who) + ": ys is required when ncond > 0");
    return CNF_OK;
}

int cnf_aug_f(cnf_handle* h, float* du, const float* u, float t, const float* eps,
              const float* ys, int64_t B, void* stream) {
    int rc = check_call(h, eps, ys, B, "cnf_aug_f");
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
        HIP_TRY(mfma_solve(h->plan, h->packed_dev, a, st));
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
    for (int i = 0; i < 6; ++i) k[i] = h->kbuf + (size_t)i * n;
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
    int rc = check_call(h, eps, ys, B, "cnf_integrate_fixed");
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
        HIP_TRY(mfma_solve(h->plan, h->packed_dev, a, st));
        return CNF_OK;
    }
    if (u1 != u0)
        HIP_TRY(hipMemcpyAsync(u1, u0, (size_t)h->S * (size_t)B * sizeof(float),
                               hipMemcpyDeviceToDevice, st));
    return simt_integrate(h, alg, nsteps, t0, t1, u1, eps, ys, B, st);
}

int cnf_inference_fixed(cnf_handle* h, int alg, int nsteps, float t0, float t1, const float* x,
                        const float* eps, const float* ys, int64_t B, float* logp, float* regs,
                        float* u_final, void* stream) {
    int rc = check_call(h, eps, ys, B, "cnf_inference_fixed");
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
        HIP_TRY(mfma_solve(h->plan, h->packed_dev, a, st));
        return CNF_OK;
    }
    rc = ensure_kbuf(h, B);
    if (rc) return rc;
    float* u = u_final ? u_final : h->kbuf + 6 * (size_t)h->S * (size_t)h->kbuf_B;
    HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, u, st));
    rc = simt_integrate(h, alg, nsteps, t0, t1, u, eps, ys, B, st);
    if (rc) return rc;
    HIP_TRY(epilogue(u, h->cfg.nvars, h->D, reg_aug, B, logp, regs, st));
    return CNF_OK;
}

// fixed steps on a given grid: u advanced in place, 6 (Tsit5) or 4 (RK4) evaluations per step on the handle's family
static int integrate_grid(cnf_handle* h, int alg, int nsteps, const float* tgrid, float* u, const float* eps, const float* ys,
                          int64_t B, hipStream_t st);

// f(u + dt sum coef k, t) on whichever family serves the handle; `stage` is scratch for the fused path, whose
// single-call kernel takes the stage state itself
static int eval_dynamics(cnf_handle* h, const StageIn& in, float t, const float* eps, const float* ys, int64_t B, float* du,
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
        HIP_TRY(mfma_solve(h->plan, h->packed_dev, a, st));
        return CNF_OK;
    }
    return generic_aug_f(h, in, t, eps, ys, B, du, first, st);
}

static int integrate_grid(cnf_handle* h, int alg, int nsteps, const float* tgrid, float* u, const float* eps, const float* ys,
                          int64_t B, hipStream_t st) {
    const size_t n = (size_t)h->S * (size_t)B;
    if (B > h->ebuf_B) {
        if (h->ebuf) HIP_TRY(hipFree(h->ebuf));
        h->ebuf = nullptr; h->ebuf_B = 0;
        HIP_TRY(hipMalloc((void**)&h->ebuf, 8 * n * sizeof(float)));
        h->ebuf_B = B;
    }
    const size_t slot = (size_t)h->S * (size_t)h->ebuf_B;
    float* stage = h->ebuf + 7 * slot;
    float* k[6];
    for (int i = 0; i < 6; ++i) k[i] = h->ebuf + (size_t)i * slot;
    const Tableau T = make_tableau(alg);
    for (int s = 0; s < nsteps; ++s) {
        const float tn = tgrid[s], dt = tgrid[s + 1] - tgrid[s];
        for (int i = 0; i < T.ns; ++i) {
            StageIn in{};
            in.u = u; in.nprev = i; in.dt = dt;
            for (int j = 0; j < i; ++j) { in.k[j] = k[j]; in.coef[j] = T.a[i][j]; }
            int rc = eval_dynamics(h, in, tn + T.c[i] * dt, eps, ys, B, k[i], stage, s == 0 && i == 0, st);
            if (rc) return rc;
        }
        StageIn fin{};
        fin.u = u; fin.nprev = T.ns; fin.dt = dt;
        for (int j = 0; j < T.ns; ++j) { fin.k[j] = k[j]; fin.coef[j] = T.b[j]; }
        HIP_TRY(rk_update(u, fin, (int64_t)n, st));
    }
    return CNF_OK;
}

int cnf_step_embedded(cnf_handle* h, int alg, int flags, float t, float dt, const float* u, const float* eps,
                      const float* ys, int64_t B, float abstol, float reltol, float* u_new, double* err_sumsq,
                      void* stream) {
    int rc = check_call(h, eps, ys, B, "cnf_step_embedded");
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
    if (B > h->ebuf_B) {
        if (h->ebuf) HIP_TRY(hipFree(h->ebuf));
        h->ebuf = nullptr; h->ebuf_B = 0;
        HIP_TRY(hipMalloc((void**)&h->ebuf, 8 * n * sizeof(float)));
        h->ebuf_B = B;
        flags = 0;   // the cached stages went with the old buffer
    }
    if (!h->err_partial) HIP_TRY(hipMalloc((void**)&h->err_partial, kErrBlocks * sizeof(double)));
    const size_t slot = (size_t)h->S * (size_t)h->ebuf_B;
    float* stage = h->ebuf + 7 * slot;
    if (flags & CNF_STEP_FSAL) { const int tmp = h->ek[0]; h->ek[0] = h->ek[6]; h->ek[6] = tmp; }
    float* k[7];
    for (int i = 0; i < 7; ++i) k[i] = h->ebuf + (size_t)h->ek[i] * slot;
    const Tableau T = make_tableau(CNF_ALG_TSIT5);
    // b - bhat of the embedded 4th-order solution (Tsitouras 2011); satisfies the order-4 conditions to 1e-15
    static const float btilde[7] = {-0.00178001105222577714f, -0.0008164344596567469f, 0.007880878010261995f,
                                    -0.1447110071732629f, 0.5823571654525552f, -0.45808210592918697f,
                                    0.015151515151515152f};
    if (h->path == CNF_PATH_MFMA && mfma_plan_is_per_wave(h->plan)) {
        // fused attempt: the six stages and the update in ONE launch of the solve kernel (nsteps = 1), which also
        // writes every stage derivative; then the 7th stage at u_new and the error reduction.  (The fused step
        // evaluates its own first stage, so the FSAL / RETRY hints save nothing here; 3 launches instead of 14.)
        for (int i = 0; i < 7; ++i) k[i] = h->ebuf + (size_t)i * n;   // [stage][B][S], packed for this B
        SolveArgs a{};
        a.u0 = u; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = 1; a.alg = CNF_ALG_TSIT5; a.t0 = t; a.t1 = t + dt;
        a.u_out = u_new; a.nvars = h->cfg.nvars; a.reg_aug = 0; a.kfull = h->ebuf;
        HIP_TRY(mfma_solve(h->plan, h->packed_dev, a, st));
        StageIn last{};
        last.u = u_new; last.nprev = 0; last.dt = 0.f;
        rc = eval_dynamics(h, last, t + dt, eps, ys, B, k[6], stage, false, st);
        if (rc) return rc;
        HIP_TRY(embedded_error(u, u_new, k, btilde, 7, dt, abstol, reltol, (int64_t)n, h->err_partial, err_sumsq, st));
        return CNF_OK;
    }
    if (!(flags & (CNF_STEP_FSAL | CNF_STEP_RETRY))) {
        StageIn in{};
        in.u = u; in.nprev = 0; in.dt = 0.f;
        rc = eval_dynamics(h, in, t, eps, ys, B, k[0], stage, true, st);
        if (rc) return rc;
    }
    for (int i = 1; i < 6; ++i) {
        StageIn in{};
        in.u = u; in.nprev = i; in.dt = dt;
        for (int j = 0; j < i; ++j) { in.k[j] = k[j]; in.coef[j] = T.a[i][j]; }
        rc = eval_dynamics(h, in, t + T.c[i] * dt, eps, ys, B, k[i], stage, false, st);
        if (rc) return rc;
    }
    StageIn fin{};
    fin.u = u; fin.nprev = 6; fin.dt = dt;
    for (int j = 0; j < 6; ++j) { fin.k[j] = k[j]; fin.coef[j] = T.b[j]; }
    HIP_TRY(rk_update(u_new, fin, (int64_t)n, st));
    StageIn last{};
    last.u = u_new; last.nprev = 0; last.dt = 0.f;
    rc = eval_dynamics(h, last, t + dt, eps, ys, B, k[6], stage, false, st);   // 7th stage = first stage of the next step
    if (rc) return rc;
    HIP_TRY(embedded_error(u, u_new, k, btilde, 7, dt, abstol, reltol, (int64_t)n, h->err_partial, err_sumsq, st));
    return CNF_OK;
}

// ---- variable-step variable-order Adams PECE: the reference's default alg = VCABM() ----

static inline float* vc_vec(cnf_handle* h, int i) { return h->vc_buf + (size_t)i * (size_t)h->S * (size_t)h->vc_B; }
static inline float* vc_diffs(cnf_handle* h, int half) { return vc_vec(h, 6 + half * kVcSlots); }

static int vc_check(cnf_handle* h, const float* eps, const float* ys, int64_t B, const char* who) {
    int rc = check_call(h, eps, ys, B, who);
    if (rc) return rc;
    if (h->vc_B < 0 || B != h->vc_B) return fail(CNF_ERR_INVALID, std::string(who) + ": cnf_vcabm_begin was not called for this batch");
    return CNF_OK;
}

int cnf_vcabm_begin(cnf_handle* h, float t0, const float* u0, const float* eps, const float* ys, int64_t B, void* stream) {
    int rc = check_call(h, eps, ys, B, "cnf_vcabm_begin");
    if (rc) return rc;
    if (B > 0 && !u0) return fail(CNF_ERR_INVALID, "cnf_vcabm_begin: null u0");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    if (B != h->vc_B) {
        if (h->vc_buf) HIP_TRY(hipFree(h->vc_buf));
        h->vc_buf = nullptr; h->vc_B = -1;
        const size_t n = (size_t)h->S * (size_t)B;
        if (n) HIP_TRY(hipMalloc((void**)&h->vc_buf, (6 + 2 * kVcSlots) * n * sizeof(float)));
        h->vc_B = B;
    }
    if (!h->vc_partial) HIP_TRY(hipMalloc((void**)&h->vc_partial, (vcabm_partial_doubles() + 8) * sizeof(double)));   // + result slots of cnf_solve_vcabm
    h->vc_iu = 0; h->vc_iun = 2; h->vc_if = 3; h->vc_ifn = 5; h->vc_cur = 0;
    h->vc_nhist = 0; h->vc_k = 0; h->vc_avail = 0; h->vc_m = 0; h->vc_t = t0; h->vc_dt = 0.0;
    for (double& d : h->vc_hist) d = 0.0;
    if (B == 0) return CNF_OK;
    const size_t n = (size_t)h->S * (size_t)B;
    HIP_TRY(hipMemcpyAsync(vc_vec(h, h->vc_iu), u0, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    StageIn in{};
    in.u = vc_vec(h, h->vc_iu); in.nprev = 0; in.dt = 0.f;
    return eval_dynamics(h, in, t0, eps, ys, B, vc_vec(h, h->vc_if), nullptr, true, st);
}

int cnf_vcabm_attempt(cnf_handle* h, int order, float dt, const float* eps, const float* ys, int64_t B, float abstol,
                      float reltol, double* err3, void* stream) {
    int rc = vc_check(h, eps, ys, B, "cnf_vcabm_attempt");
    if (rc) return rc;
    if (order < 1 || order > CNF_VCABM_MAX_ORDER || order > h->vc_nhist + 1)
        return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: order must be in 1..12 and at most one more than the accepted steps");
    if (std::min(order, h->vc_nhist) > h->vc_avail)
        return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: the order can rise by at most one per accepted step (the stored differences end there)");
    if (!(abstol >= 0.f) || !(reltol >= 0.f) || (abstol == 0.f && reltol == 0.f))
        return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: tolerances must be non-negative and not both zero");
    if (dt == 0.f || !(dt == dt)) return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: dt must be non-zero");
    if (B == 0) return CNF_OK;
    if (!err3) return fail(CNF_ERR_INVALID, "cnf_vcabm_attempt: null err3");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    const int k = order;
    const int m = std::min(k + 1, h->vc_nhist + 1);
    // step sizes newest first, the candidate in front (Hairer, Noersett, Wanner I, III.5: t_{n+1} - t_{n-j+1} = sum of the j newest)
    double dts[kVcSlots + 2];
    dts[0] = dt;
    for (int i = 0; i <= kVcSlots; ++i) dts[i + 1] = h->vc_hist[i];
    VcCoef c{};
    c.ps_old = vc_diffs(h, h->vc_cur);
    c.ps_new = vc_diffs(h, h->vc_cur ^ 1);
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
    float *u = vc_vec(h, h->vc_iu), *p = vc_vec(h, 1), *un = vc_vec(h, h->vc_iun), *d = vc_vec(h, 4);
    HIP_TRY(vcabm_predict(vc_vec(h, h->vc_if), u, c, n, p, st));                                    // P
    StageIn in{};
    in.u = p; in.nprev = 0; in.dt = 0.f;
    rc = eval_dynamics(h, in, (float)(h->vc_t + (double)dt), eps, ys, B, d, nullptr, false, st);   // E
    if (rc) return rc;
    HIP_TRY(vcabm_correct(d, p, u, c, abstol, reltol, n, un, h->vc_partial, err3, st));             // C
    h->vc_k = k; h->vc_m = m; h->vc_dt = dt;
    return CNF_OK;
}

int cnf_vcabm_accept(cnf_handle* h, const float* eps, const float* ys, int64_t B, float abstol, float reltol,
                     double* err_up, void* stream) {
    int rc = vc_check(h, eps, ys, B, "cnf_vcabm_accept");
    if (rc) return rc;
    if (B == 0) return CNF_OK;
    if (h->vc_k == 0) return fail(CNF_ERR_INVALID, "cnf_vcabm_accept: no pending attempt");
    const int k = h->vc_k;
    if (err_up && (k >= CNF_VCABM_MAX_ORDER || h->vc_nhist < k))
        return fail(CNF_ERR_INVALID, "cnf_vcabm_accept: the order k+1 estimate needs k accepted steps and k < 12");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)h->S * (size_t)B;
    float *u = vc_vec(h, h->vc_iu), *un = vc_vec(h, h->vc_iun), *fnew = vc_vec(h, h->vc_ifn);
    StageIn in{};
    in.u = un; in.nprev = 0; in.dt = 0.f;
    rc = eval_dynamics(h, in, (float)(h->vc_t + h->vc_dt), eps, ys, B, fnew, nullptr, false, st);   // E
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
        c.ps_new = vc_diffs(h, h->vc_cur ^ 1);
        c.ld = n; c.k = k;
        c.e0 = (float)(h->vc_dt * gs[k + 1]);
        HIP_TRY(vcabm_errup(fnew, u, un, c, abstol, reltol, (int64_t)n, h->vc_partial, err_up, st));
    }
    std::swap(h->vc_iu, h->vc_iun);
    std::swap(h->vc_if, h->vc_ifn);
    h->vc_cur ^= 1;
    for (int i = kVcSlots; i > 0; --i) h->vc_hist[i] = h->vc_hist[i - 1];
    h->vc_hist[0] = h->vc_dt;
    h->vc_t += h->vc_dt;
    h->vc_nhist += 1;
    h->vc_avail = h->vc_m;
    h->vc_k = 0;
    return CNF_OK;
}

int cnf_vcabm_state(cnf_handle* h, int64_t B, float* u_out, double* t_out, void* stream) {
    if (!h) return fail(CNF_ERR_INVALID, "cnf_vcabm_state: null handle");
    if (h->vc_B < 0 || B != h->vc_B) return fail(CNF_ERR_INVALID, "cnf_vcabm_state: cnf_vcabm_begin was not called for this batch");
    if (t_out) *t_out = h->vc_t;
    if (u_out && B > 0) {
        DeviceGuard g(h->cfg.device_id);
        HIP_TRY(hipMemcpyAsync(u_out, vc_vec(h, h->vc_iu), (size_t)h->S * (size_t)B * sizeof(float), hipMemcpyDeviceToDevice,
                               (hipStream_t)stream));
    }
    return CNF_OK;
}

// The whole default solve in one call: cnf_vcabm_begin / _attempt / _accept driven by the step-size and order policy of
// icnf._vcabm_integrate (the host side of the reference's solver), restated here so that a single-process caller pays one
// library call per solve instead of two per step.  Synchronises `stream` (the policy reads the error sums).
int cnf_solve_vcabm(cnf_handle* h, float t0, float t1, const float* u0, const float* eps, const float* ys, int64_t B,
                    float abstol, float reltol, float dt_init, int maxiters, float* u1, cnf_solve_stats* stats,
                    float* dts_out, int32_t* orders_out, int32_t record_cap, void* stream) {
    if (stats) *stats = cnf_solve_stats{};
    int rc = cnf_vcabm_begin(h, t0, u0, eps, ys, B, stream);
    if (rc) return rc;
    if (!(abstol >= 0.f) || !(reltol >= 0.f) || (abstol == 0.f && reltol == 0.f))
        return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: tolerances must be non-negative and not both zero");
    if (B > 0 && !u1) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: null u1");
    if (maxiters < 1) return fail(CNF_ERR_INVALID, "cnf_solve_vcabm: maxiters >= 1 required");
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
    double* res = h->vc_partial + vcabm_partial_doubles();   // device result slots
    double host[4];
    auto fetch = [&](int cnt) -> int {
        HIP_TRY(hipMemcpyAsync(host, res, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return CNF_OK;
    };
    double dt;
    if (dt_init != 0.f) {
        dt = std::min((double)std::fabs(dt_init), span);
    } else {   // ode_determine_initdt (Hairer, Noersett, Wanner I, II.4) with the algorithm order 7, RMS norm over all S*B entries
        float *u = vc_vec(h, h->vc_iu), *f0 = vc_vec(h, h->vc_if), *ue = vc_vec(h, 1), *f1 = vc_vec(h, 4);
        HIP_TRY(vcabm_scaled_sumsq(u, nullptr, u, abstol, reltol, (int64_t)n, h->vc_partial, res, st));
        HIP_TRY(vcabm_scaled_sumsq(f0, nullptr, u, abstol, reltol, (int64_t)n, h->vc_partial, res + 1, st));
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
        rc = eval_dynamics(h, in, (float)((double)t0 + tdir * h0), eps, ys, B, f1, nullptr, false, st);
        if (rc) return rc;
        ++nf;
        HIP_TRY(vcabm_scaled_sumsq(f1, f0, u, abstol, reltol, (int64_t)n, h->vc_partial, res, st));
        rc = fetch(1);
        if (rc) return rc;
        const double d2 = std::sqrt(host[0] / ntot) / h0, dmax = std::max(d1, d2);
        const double h1 = dmax <= 1e-15 ? std::max(1e-6, h0 * 1e-3) : std::pow(10.0, -(2.0 + std::log10(dmax)) / 8.0);
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
// icnf._adaptive_integrate restated inside the library (single process).  Synchronises `stream`.
static int ensure_adaptive_buf(cnf_handle* h, int64_t B) {
    if (B <= h->ad_B) return CNF_OK;
    if (h->ad_buf) HIP_TRY(hipFree(h->ad_buf));
    h->ad_buf = nullptr; h->ad_B = 0;
    HIP_TRY(hipMalloc((void**)&h->ad_buf, 6 * (size_t)h->S * (size_t)B * sizeof(float)));   // 4 for the solve, 2 for cnf_loss_grad_adaptive
    h->ad_B = B;
    return CNF_OK;
}

static int solve_tsit5_impl(cnf_handle* h, float t0, float t1, const float* u0, const float* eps, const float* ys, int64_t B,
                            float abstol, float reltol, float dt_init, int maxiters, float* u1, cnf_solve_stats* stats,
                            std::vector<double>* steps, void* stream) {
    if (stats) *stats = cnf_solve_stats{};
    int rc = check_call(h, eps, ys, B, "cnf_solve_tsit5");
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
    rc = ensure_adaptive_buf(h, B);
    if (rc) return rc;
    if (!h->vc_partial) HIP_TRY(hipMalloc((void**)&h->vc_partial, (vcabm_partial_doubles() + 8) * sizeof(double)));
    const size_t slot = (size_t)h->S * (size_t)h->ad_B;
    float *ua = h->ad_buf, *ub = ua + slot, *f0 = ub + slot, *f1 = f0 + slot;
    HIP_TRY(hipMemcpyAsync(ua, u0, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    double* res = h->vc_partial + vcabm_partial_doubles();
    double host[2];
    auto fetch = [&](int cnt) -> int {
        HIP_TRY(hipMemcpyAsync(host, res, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return CNF_OK;
    };
    int nf = 0, naccept = 0, nreject = 0;
    double dt;
    if (dt_init != 0.f) {
        dt = std::min((double)std::fabs(dt_init), span);
    } else {   // ode_determine_initdt (Hairer, Noersett, Wanner I, II.4), order 5
        StageIn in{};
        in.u = ua; in.nprev = 0; in.dt = 0.f;
        rc = eval_dynamics(h, in, t0, eps, ys, B, f0, nullptr, true, st);
        if (rc) return rc;
        ++nf;
        HIP_TRY(vcabm_scaled_sumsq(ua, nullptr, ua, abstol, reltol, (int64_t)n, h->vc_partial, res, st));
        HIP_TRY(vcabm_scaled_sumsq(f0, nullptr, ua, abstol, reltol, (int64_t)n, h->vc_partial, res + 1, st));
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
        rc = eval_dynamics(h, in1, (float)((double)t0 + tdir * h0), eps, ys, B, f1, nullptr, false, st);
        if (rc) return rc;
        ++nf;
        HIP_TRY(vcabm_scaled_sumsq(f1, f0, ua, abstol, reltol, (int64_t)n, h->vc_partial, res, st));
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
    for (; it < maxiters; ++it) {
        if (std::fabs((double)t1 - t) <= 1e-7 * std::max(1.0, span)) break;
        const bool last = dt >= std::fabs((double)t1 - t) * (1.0 - 1e-6);
        const double step = last ? std::fabs((double)t1 - t) : dt;     // tstop: never step over t1
        rc = cnf_step_embedded(h, CNF_ALG_TSIT5, flags, (float)t, (float)(tdir * step), ua, eps, ys, B, abstol, reltol, ub, res, stream);
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
    return CNF_OK;
}

int cnf_solve_tsit5(cnf_handle* h, float t0, float t1, const float* u0, const float* eps, const float* ys, int64_t B,
                    float abstol, float reltol, float dt_init, int maxiters, float* u1, cnf_solve_stats* stats,
                    float* dts_out, int32_t record_cap, void* stream) {
    std::vector<double> steps;
    const int rc = solve_tsit5_impl(h, t0, t1, u0, eps, ys, B, abstol, reltol, dt_init, maxiters, u1, stats, &steps, stream);
    if (dts_out)
        for (size_t i = 0; i < steps.size() && (int64_t)i < record_cap; ++i) dts_out[i] = (float)steps[i];
    return rc;
}

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

// fused reverse-sweep kernel, unless CNF_GRAD_LAYERED=1 forces the layer-wise path (tests, A/B timing)
static bool grad_is_fused(const cnf_handle* h) {
    const char* force = getenv("CNF_GRAD_LAYERED");
    return h->path == CNF_PATH_MFMA && grad_supported(grad_cfg(h)) && mfma_plan_is_per_wave(h->plan) &&
           !(force && *force && *force != '0');
}

// slab-accumulator kernel for the mid-width two-hidden-layer nets (CNF_GRAD_LAYERED=1 skips it too)
static bool grad_uses_slab(const cnf_handle* h) {
    const char* force = getenv("CNF_GRAD_LAYERED");
    return (h->slab_packed || !h->have_params) && grad_slab_supported(h->cfg) && !(force && *force && *force != '0');
}

int cnf_grad_path(const cnf_handle* h) {
    if (!h) return CNF_ERR_INVALID;
    if (grad_is_fused(h) || grad_uses_slab(h)) return 1;
    return layered_grad_supported(h->cfg) ? 2 : 0;
}

}  // extern "C"

// loss sums + gradient on a uniform grid (tgrid == nullptr: nsteps steps from t0 to t1) or on the caller's non-uniform
// grid (tgrid: host, nsteps + 1 times; t0 / t1 ignored).  The same three gradient implementations serve both.
static int loss_grad_impl(cnf_handle* h, const char* who, int alg, int nsteps, float t0, float t1, const float* tgrid,
                          const float* x, const float* eps, const float* ys, int64_t B, const float* lambdas,
                          float* grad, float* grad_x, float* sums4, void* stream) {
    int rc = check_call(h, eps, ys, B, who);
    if (rc) return rc;
    const std::string w(who);
    if (nsteps < 1) return fail(CNF_ERR_INVALID, w + ": nsteps >= 1 required");
    if (alg != CNF_ALG_RK4 && alg != CNF_ALG_TSIT5) return fail(CNF_ERR_INVALID, w + ": unknown alg");
    if (!x || !grad || !lambdas) return fail(CNF_ERR_INVALID, w + ": null x/grad/lambdas");
    const bool fused = grad_is_fused(h) && h->grad_packed;
    if (!fused && !layered_grad_supported(h->cfg))
        return fail(CNF_ERR_UNSUPPORTED, w + ": no gradient path for this configuration");
    DeviceGuard g(h->cfg.device_id);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(grad, 0, h->nparams * sizeof(float), st));
    if (B == 0) {
        if (sums4) HIP_TRY(hipMemsetAsync(sums4, 0, 4 * sizeof(float), st));
        return CNF_OK;
    }
    const float* tgrid_dev = nullptr;
    if (tgrid) {   // the fused kernels read the step times from device memory (uniform loads, once per step)
        t0 = tgrid[0]; t1 = tgrid[nsteps];
        if ((size_t)nsteps + 1 > h->tgrid_cap) {
            if (h->tgrid_dev) HIP_TRY(hipFree(h->tgrid_dev));
            h->tgrid_dev = nullptr; h->tgrid_cap = 0;
            const size_t cap = ((size_t)nsteps + 1 + 63) / 64 * 64;
            HIP_TRY(hipMalloc((void**)&h->tgrid_dev, cap * sizeof(float)));
            h->tgrid_cap = cap;
        }
        HIP_TRY(hipMemcpyAsync(h->tgrid_dev, tgrid, ((size_t)nsteps + 1) * sizeof(float), hipMemcpyHostToDevice, st));
        tgrid_dev = h->tgrid_dev;
    }
    if (!fused) {
        // the loss sums come from the regular solve on whichever family serves the handle
        if (sums4) {
            const size_t need = ((size_t)h->S + 4) * (size_t)B * sizeof(float);
            if (need > h->grad_ws_bytes) {
                if (h->grad_ws) HIP_TRY(hipFree(h->grad_ws));
                h->grad_ws = nullptr; h->grad_ws_bytes = 0;
                HIP_TRY(hipMalloc((void**)&h->grad_ws, need));
                h->grad_ws_bytes = need;
            }
            float* logp = h->grad_ws;
            float* regs = logp + B;
            if (tgrid) {   // the loss of the same discrete solve: augmented state advanced over the grid, then the epilogue
                float* u = regs + 3 * (size_t)B;
                const int ra0 = (h->cfg.mode != CNF_MODE_EXACT && h->cfg.reg_aug && h->cfg.naug > 0) ? 1 : 0;
                HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, u, st));
                rc = integrate_grid(h, alg, nsteps, tgrid, u, eps, ys, B, st);
                if (rc) return rc;
                HIP_TRY(epilogue(u, h->cfg.nvars, h->D, ra0, B, logp, regs, st));
            } else {
                rc = cnf_inference_fixed(h, alg, nsteps, t0, t1, x, eps, ys, B, logp, regs, nullptr, stream);
                if (rc) return rc;
            }
            if (!h->loss_partial) HIP_TRY(hipMalloc((void**)&h->loss_partial, 256 * 4 * sizeof(float)));
            HIP_TRY(loss_sums(logp, regs, B, h->loss_partial, sums4, st));
        }
        const bool hutch = h->cfg.mode != CNF_MODE_EXACT;   // the exact-trace dynamics carry no regularisers (icnf.jl:297-339)
        const int ra = (hutch && h->cfg.reg_aug && h->cfg.naug > 0) ? 1 : 0;
        const float lam[3] = {hutch && h->cfg.reg_z ? lambdas[0] : 0.f, hutch && h->cfg.reg_j ? lambdas[1] : 0.f, ra ? lambdas[2] : 0.f};
        if (grad_uses_slab(h)) {
            // two-hidden-layer nets of 4..7 hidden tiles: tile-fused reverse sweep with slab accumulators (cnf_grad_slab.hip)
            if (h->num_cus == 0) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, h->cfg.device_id));
                h->num_cus = prop.multiProcessorCount;
            }
            const size_t need = grad_slab_ws_floats(h->cfg, alg, nsteps, B, h->num_cus);
            if (need > h->slab_ws_floats) {
                if (h->slab_ws) HIP_TRY(hipFree(h->slab_ws));
                h->slab_ws = nullptr; h->slab_ws_floats = 0;
                HIP_TRY(hipMalloc((void**)&h->slab_ws, need * sizeof(float)));
                h->slab_ws_floats = need;
            }
            HIP_TRY(grad_slab_launch(h->cfg, h->slab_packed, x, eps, ys, h->w_off.data(), h->b_off.data(), alg, nsteps, t0, t1, tgrid_dev, B, lam,
                                     h->slab_ws, grad, grad_x, h->num_cus, st));
            return CNF_OK;
        }
        std::string msg;
        hipError_t e = layered_grad(&h->layered, h->cfg, h->P_dev, h->w_off.data(), h->b_off.data(), x, eps, ys, alg, nsteps,
                                    t0, t1, tgrid, B, lam, grad, grad_x, st, &msg);
        if (e == hipErrorNotSupported) return fail(CNF_ERR_UNSUPPORTED, w + ": " + msg);
        if (e != hipSuccess) return fail(CNF_ERR_HIP, w + ": " + msg);
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
    const size_t ckpt_z_floats = (size_t)(nsteps + 1) * (size_t)ntiles * 64 * (size_t)ckpt_zr;
    const size_t ckpt_k_floats = (size_t)nsteps * nstages * (size_t)ntiles * 64 * (size_t)ckpt_zr;
    const size_t ckpt_floats = ckpt_z_floats + ckpt_k_floats;
    const cnf_config gc = grad_cfg(h);
    const bool exact = h->cfg.mode == CNF_MODE_EXACT;
    const size_t slab_floats = grad_slab_floats(gc, h->num_cus);
    const size_t state_floats = tgrid ? 2 * (size_t)h->S * (size_t)B : 0;   // ping-pong states of the step-by-step forward pass
    const size_t unit_floats = exact ? (size_t)h->D * (size_t)h->D * (size_t)B : 0;   // the D unit probes of every column
    const size_t need = (ckpt_floats + 4 * (size_t)B + slab_floats + state_floats + unit_floats) * sizeof(float);
    if (need > h->grad_ws_bytes) {
        if (h->grad_ws) HIP_TRY(hipFree(h->grad_ws));
        h->grad_ws = nullptr; h->grad_ws_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->grad_ws, need));
        h->grad_ws_bytes = need;
    }
    float* ckpt = h->grad_ws;
    float* ckpt_k = ckpt + ckpt_z_floats;
    float* logp = ckpt + ckpt_floats;
    float* regs = logp + B;
    float* slab = regs + 3 * (size_t)B;
    const int reg_aug = (!exact && h->cfg.reg_aug && h->cfg.naug > 0) ? 1 : 0;
    if (!tgrid) {
        SolveArgs a{};
        a.x = x; a.eps = eps; a.ys = ys; a.B = B; a.nsteps = nsteps; a.alg = alg; a.t0 = t0; a.t1 = t1;
        a.logp = logp; a.regs = regs; a.nvars = h->cfg.nvars; a.reg_aug = reg_aug; a.ckpt = ckpt; a.ckpt_k = ckpt_k;
        HIP_TRY(mfma_solve(h->plan, h->packed_dev, a, st));
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
            HIP_TRY(mfma_solve(h->plan, h->packed_dev, a, st));
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
    HIP_TRY(grad_launch(gc, h->grad_packed, ckpt, ckpt_k, ckpt_zr, probes, ys, h->w_off.data(), h->b_off.data(), alg, nsteps, t0, t1,
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
    int rc = check_call(h, eps, ys, B, "cnf_loss_grad_adaptive");
    if (rc) return rc;
    if (!x || !grad || !lambdas) return fail(CNF_ERR_INVALID, "cnf_loss_grad_adaptive: null x/grad/lambdas");
    if (t0 == t1) return fail(CNF_ERR_INVALID, "cnf_loss_grad_adaptive: empty time span");
    std::vector<float> grid;
    if (B == 0) {   // nothing to step over: the fixed entry zeroes grad / sums4
        return loss_grad_impl(h, "cnf_loss_grad_adaptive", CNF_ALG_TSIT5, 1, t0, t1, nullptr, x, eps, ys, B, lambdas, grad, grad_x, sums4, stream);
    }
    {
        DeviceGuard g(h->cfg.device_id);
        rc = ensure_adaptive_buf(h, B);
        if (rc) return rc;
        const size_t slot = (size_t)h->S * (size_t)h->ad_B;
        float* u = h->ad_buf + 4 * slot;
        HIP_TRY(assemble_u0(x, h->cfg.nvars, h->S, B, u, (hipStream_t)stream));
        std::vector<double> steps;
        rc = solve_tsit5_impl(h, t0, t1, u, eps, ys, B, abstol, reltol, dt_init, maxiters, u + slot, stats, &steps, stream);
        if (rc) return rc;
        double t = t0;
        grid.push_back(t0);
        for (double d : steps) { t += d; grid.push_back((float)t); }
        grid.back() = t1;
    }
    if (tgrid_out)
        for (size_t i = 0; i < grid.size() && (int64_t)i < grid_cap; ++i) tgrid_out[i] = grid[i];
    return loss_grad_impl(h, "cnf_loss_grad_adaptive", CNF_ALG_TSIT5, (int)grid.size() - 1, 0.f, 0.f, grid.data(), x, eps, ys, B,
                          lambdas, grad, grad_x, sums4, stream);
}

}  // extern "C"
