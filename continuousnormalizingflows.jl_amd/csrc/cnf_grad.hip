// cnf_grad.hip — host side of the register-accumulator parameter gradient (gfx950): instance table, workspace sizes, the launch
// (zeroed slabs, the reverse-sweep kernel of cnf_grad2.hip / cnf_grad2_probes.hip, the slab reduction) and the reduction kernel.
//
// SURVEY.md §8(f) rank 2.  The reference differentiates `loss` through SciMLBase.solve with QuadratureAdjoint + ZygoteVJP
// (src/core/icnf.jl:90-99; src/exts/mlj_ext/core_icnf.jl:42-51).  With a fixed-step solver the exact gradient of the discrete loss is
// reverse mode through the RK steps (discretise-then-optimise): grad = sum_j d(-logp_j)/dp over the batch columns given.
//
// Regularised objective (TrainMode{true}: + l1 |zdot| + l2 |eps^T J| + l3 |z_aug|, icnf.jl:184-251,628-637):
//   kbar += c_E zdot/|zdot|;  gbar = c_n g/|g| - c_l eps with g = W_1[:,0:D]^T delta_1;  dbar_1 = W_1[:,0:D] gbar;
//   lambda_N += l3 z_aug/|z_aug|.
//
// Two launches per gradient: the forward solve kernel (cnf_mfma_kernel.h) with step and stage checkpoints, then the reverse sweep.
// Per wave: one 16-sample tile, steps in reverse; per stage
//   recompute   h_l, act'_l                                   (forward images)
//   pullback    delta_L = c .* act'_L, u_l = W_{l+1}^T delta_{l+1}, delta_l = u_l .* act'_l
//   reverse of  Phi = kbar^T zdot - c_l <delta_1, q>          (c = W_N^T eps)
//     bottom-up: dbar_1 = W_1[:,0:D] gbar; ubar_l = dbar_l .* act'_l; abar''_l = dbar_l .* u_l;
//                dbar_{l+1} = W_{l+1} ubar_l;  Wbar_{l+1} += delta_{l+1} ubar_l^T
//     top-down : hbar_L = W_N^T kbar; abar_l = hbar_l .* act'_l + abar''_l .* act''_l;
//                Wbar_l += abar_l h_{l-1}^T; bbar_l += abar_l; hbar_{l-1} = W_l^T abar_l;  Zbar = W_1[:,0:D]^T abar_1
// Weight cotangents are outer products summed over the tile's 16 samples: MFMAs with the SAMPLE index on K, so each operand tile is
// transposed once through LDS (ds_write_b128 into a padded tile + conflict-free single-dword reads).  At the end each wave deposits
// its accumulator tiles in a slab and grad_reduce_kernel sums the slabs in a fixed order into the Lux-layout gradient: no atomics,
// bit-reproducible.  (History: LDS float atomics - 89 ms at cfg2; wave-private slabs updated by load / MFMA / store - 19 ms; row-block
// ownership with an LDS exchange per matrix, rounds 2-4 - 14.4 ms; every wave its own whole gradient, round 5 - 11.9 ms.)
#include "cnf_grad_dev.h"

namespace cnf {

// Sum the waves' slabs in a fixed order and scatter into the Lux-layout gradient (every parameter is
// written by exactly one thread: no atomics).
template <int HT, int L, int ZR, int CR>
__global__ void __launch_bounds__(1024)
grad_reduce_kernel(const float* __restrict__ slab, int nwaves, GArgs a, float* __restrict__ grad) {
    using SL = GradSlab<HT, L, ZR, CR>;
    // 64 elements per block, 16 slab groups per element (fixed partition, fixed combine order; with 4 groups a thread summed 256
    // slabs one load after the other: 97 us at cfg2 for 51 MB)
    __shared__ float part[16][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float acc0 = 0.f, acc1 = 0.f;
    if (e < SL::TOTAL) {
        int w = grp;
        for (; w + 16 < nwaves; w += 32) {   // two independent chains: two loads in flight per thread
            acc0 += slab[(long long)w * SL::TOTAL + e];
            acc1 += slab[(long long)(w + 16) * SL::TOTAL + e];
        }
        if (w < nwaves) acc0 += slab[(long long)w * SL::TOTAL + e];
    }
    part[grp][el] = acc0 + acc1;
    __syncthreads();
    if (grp != 0 || e >= SL::TOTAL) return;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) sum += (part[4 * q][el] + part[4 * q + 1][el]) + (part[4 * q + 2][el] + part[4 * q + 3][el]);
    const int H = a.H, D = a.D;
    const int ln = (e >> 2) & 63, r = e & 3, n = ln & 15, gg = ln >> 4;
    if (e < SL::WH) {                                   // W_1 image: [mt][input tile][lane][r]
        const int tl = (e - SL::W1) / 256, mt = tl / SL::NT1, it = tl % SL::NT1;
        const int out = 16 * mt + 4 * r + gg;
        const int ncore = D + (a.autonomous ? 0 : 1);   // z and time columns live in input tile 0
        if (out < H && it == 0 && n < ncore) grad[a.w_off[0] + out + H * n] = sum;
        if (out < H && it == 0 && n == 15) grad[a.b_off[0] + out] = sum;       // ones column
        if (out < H && it == 1 && n < a.C) grad[a.w_off[0] + out + H * (ncore + n)] = sum;   // condition columns
    } else if (e < SL::WN) {                            // hidden images
        const int rel = e - SL::WH, l = rel / (HT * HT * 256), tl = (rel / 256) % (HT * HT);
        const int out = 16 * (tl / HT) + 4 * r + gg, in = 16 * (tl % HT) + n;
        if (out < H && in < H) grad[a.w_off[l + 1] + out + H * in] = sum;
    } else if (e < SL::BH) {                            // W_N image: rows = state features
        const int tl = (e - SL::WN) / 256;
        const int out = 16 * (tl / HT) + 4 * r + gg, in = 16 * (tl % HT) + n;
        if (out < D && in < H) grad[a.w_off[L] + out + D * in] = sum;
    } else if (e < SL::BN) {                            // hidden biases: column 0
        const int rel = e - SL::BH, l = rel / (HT * 256), mt = (rel / 256) % HT;
        const int out = 16 * mt + 4 * r + gg;
        if (out < H && n == 0) grad[a.b_off[l + 1] + out] = sum;
    } else {
        const int mt = (e - SL::BN) / 256;
        const int out = 16 * mt + 4 * r + gg;
        if (out < D && n == 0) grad[a.b_off[L] + out] = sum;
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct GradInst {
    int HT, L, ZR, CR, ACT;
    int lds_bytes, slab_total, packed_floats;
    void (*reduce)(const float*, int, GArgs, float*);
};
#define GRAD_INST(HT, L, ZR, CR, ACT)                                                                        \
    GradInst { HT, L, ZR, CR, ACT, GradLds<HT, L, ZR, CR, ACT>::TOTAL * 4, GradSlab<HT, L, ZR, CR>::TOTAL,   \
               MfmaLayout(HT, L, ZR, CR, true, 0).total, &grad_reduce_kernel<HT, L, ZR, CR> }
#define GRAD_HT(HT, CR, ACT)                                                                                 \
    GRAD_INST(HT, 3, 2, CR, ACT), GRAD_INST(HT, 2, 2, CR, ACT), GRAD_INST(HT, 3, 4, CR, ACT), GRAD_INST(HT, 2, 4, CR, ACT)
#define GRAD_SHAPES(CR, ACT) GRAD_HT(1, CR, ACT), GRAD_HT(2, CR, ACT), GRAD_HT(3, CR, ACT), GRAD_HT(4, CR, ACT)
static const GradInst kGrad[] = {GRAD_SHAPES(0, CNF_ACT_TANH), GRAD_SHAPES(0, CNF_ACT_SOFTPLUS),
                                 GRAD_SHAPES(4, CNF_ACT_TANH), GRAD_SHAPES(4, CNF_ACT_SOFTPLUS)};

static const GradInst* grad_find(const cnf_config& c) {
    if (c.mode != CNF_MODE_HUTCH_VJP || c.nprobes < 1 || c.ncond > 16) return nullptr;
    const int N = c.n_layers, L = N - 1;
    if (L < 2 || L > 3 || c.acts[N - 1] != CNF_ACT_IDENTITY) return nullptr;
    const int H = c.widths[1];
    for (int l = 0; l < L; ++l)
        if (c.acts[l] != c.acts[0] || c.widths[l + 1] != H) return nullptr;
    const int D = c.nvars + c.naug, HT = (H + 15) / 16, ZR = (D + 3) / 4;
    if (D + (c.autonomous ? 0 : 1) > 15) return nullptr;   // input tile: 16 columns, the last one is the bias column
    const GradInst* best = nullptr;
    for (const GradInst& g : kGrad)
        if (g.HT == HT && g.L == L && g.ACT == c.acts[0] && g.ZR >= ZR && (g.CR > 0) == (c.ncond > 0) &&
            (!best || g.ZR < best->ZR))
            best = &g;
    return best;
}

bool grad_supported(const cnf_config& c) { return grad_find(c) != nullptr; }
size_t grad_packed_bytes(const cnf_config& c) { return (size_t)grad_find(c)->packed_floats * sizeof(float); }
size_t grad_slab_floats(const cnf_config& c, int num_cus) { return (size_t)num_cus * 4 * grad_find(c)->slab_total; }
void grad_shape(const cnf_config& c, int* HT, int* L, int* ZR, int* CR) {
    const GradInst* g = grad_find(c);
    *HT = g->HT; *L = g->L; *ZR = g->ZR; *CR = g->CR;
}

hipError_t grad_launch(const cnf_config& c, const float* packed_dev, const float* ckpt, const float* ckpt_k,
                       int ckpt_zr, const float* eps, const float* ys,
                       const size_t* w_off, const size_t* b_off, int alg, int nsteps, float t0, float t1, const float* tgrid_dev,
                       float probe_w, long long B, const float lam[3], float* slab, float* grad, float* grad_x, int num_cus, hipStream_t st) {
    const GradInst* gi = grad_find(c);
    if (!gi) return hipErrorNotSupported;
    // > 64 KB of dynamic LDS has to be enabled once per device and kernel
    const int idx = (int)(gi - kGrad);
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    // one probe / several probes (the probe loop rolled around the pullback and its bottom-up reverse): cnf_grad2.hip compiled twice
    if (!ckpt_k) return hipErrorNotSupported;   // the sweep reads the forward kernel's stage checkpoints
    const GradKernel kern = c.nprobes == 1 ? grad2_kernel(gi->HT, gi->L, gi->ZR, gi->CR, gi->ACT) : grad2_probes_kernel(gi->HT, gi->L, gi->ZR, gi->CR, gi->ACT);
    if (!kern) return hipErrorNotSupported;
    static DeviceOnce done_one[sizeof(kGrad) / sizeof(kGrad[0])], done_probes[sizeof(kGrad) / sizeof(kGrad[0])];
    DeviceOnce& done = (c.nprobes > 1 ? done_probes : done_one)[idx];
    if (!done.done(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, gi->lds_bytes);
        if (e != hipSuccess) return e;
        done.set(dev);
    }
    GArgs a{};
    a.packed = packed_dev; a.ckpt = ckpt; a.ckpt_k = ckpt_k; a.ckpt_zr = ckpt_zr; a.eps = eps; a.K = c.nprobes; a.ys = ys; a.C = c.ncond; a.slab = slab; a.grad_x = grad_x; a.B = B;
    a.nsteps = nsteps; a.t0 = t0; a.dt = (t1 - t0) / (float)nsteps; a.tgrid = tgrid_dev; a.probe_w = probe_w;
    a.D = c.nvars + c.naug; a.H = c.widths[1]; a.n_in = c.widths[0]; a.autonomous = c.autonomous; a.nvars = c.nvars;
    a.lam1 = lam[0]; a.lam2 = lam[1]; a.lam3 = lam[2];
    for (int l = 0; l < c.n_layers; ++l) { a.w_off[l] = (int)w_off[l]; a.b_off[l] = (int)b_off[l]; }
    a.T = make_tableau(alg);
    const long long ntiles = (B + 15) / 16;
    const long long want = (ntiles + 3) / 4;
    const int nblocks = (int)(want < num_cus ? want : num_cus);
    const int nwaves = nblocks * 4;
    hipError_t e = zero_async(slab, (size_t)nwaves * gi->slab_total * sizeof(float), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), gi->lds_bytes, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gi->reduce, dim3((gi->slab_total + 63) / 64), dim3(1024), 0, st, slab, nwaves, a, grad);
    return hipGetLastError();
}

}  // namespace cnf
