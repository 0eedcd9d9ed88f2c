// cnf_simt.hip — generic (runtime-shaped) HIP kernels: one thread per sample, f32 VALU.
//
// This is the general path of libcnf_hip.so: any Dense chain (non-uniform widths, up to
// CNF_MAX_LAYERS layers, any activation per layer), every trace mode, K probes, all
// regularisers.  It keeps the reference's *unfused* structure — one launch per dynamics call
// (make_ode_func's closure, src/core/base_icnf.jl:62-78) with the Runge-Kutta stage combination
// folded into the launch prologue — and is the cross-check for the fused MFMA solve kernel
// (cnf_mfma.hip), which is the one the benchmark measures.
//
// Memory: per-sample activations live in a global workspace laid out [row][sample] so a wave's
// 64 lanes touch 64 consecutive floats (coalesced 256-B segments); weights are read through
// wave-uniform addresses (scalar loads).
#include "cnf_internal.h"

namespace cnf {

constexpr int OB = 8;  // outputs accumulated in registers per pass over the inputs

// y[o] = act(b[o] + sum_i W(o,i) x[i]);  optional d[o] = act'(.)
__device__ __forceinline__ void simt_dense_fwd(const float* __restrict__ W,
                                               const float* __restrict__ b, int fin, int fout,
                                               int act, const float* __restrict__ x,
                                               float* __restrict__ y, float* __restrict__ d,
                                               int64_t ld) {
    for (int o0 = 0; o0 < fout; o0 += OB) {
        float acc[OB];
#pragma unroll
        for (int j = 0; j < OB; ++j) acc[j] = (o0 + j < fout) ? b[o0 + j] : 0.f;
        for (int i = 0; i < fin; ++i) {
            const float xi = x[(int64_t)i * ld];
            const float* Wi = W + (size_t)fout * i + o0;
#pragma unroll
            for (int j = 0; j < OB; ++j)
                if (o0 + j < fout) acc[j] = fmaf(Wi[j], xi, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < OB; ++j)
            if (o0 + j < fout) {
                float dd;
                const float h = act_fwd_rt(act, acc[j], dd);
                y[(int64_t)(o0 + j) * ld] = h;
                d[(int64_t)(o0 + j) * ld] = dd;
            }
    }
}

// tangent: y[o] = d[o] * sum_i W(o,i) x[i]
__device__ __forceinline__ void simt_dense_tan(const float* __restrict__ W, int fin, int fout,
                                               const float* __restrict__ x,
                                               const float* __restrict__ d,
                                               float* __restrict__ y, int64_t ld) {
    for (int o0 = 0; o0 < fout; o0 += OB) {
        float acc[OB];
#pragma unroll
        for (int j = 0; j < OB; ++j) acc[j] = 0.f;
        for (int i = 0; i < fin; ++i) {
            const float xi = x[(int64_t)i * ld];
            const float* Wi = W + (size_t)fout * i + o0;
#pragma unroll
            for (int j = 0; j < OB; ++j)
                if (o0 + j < fout) acc[j] = fmaf(Wi[j], xi, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < OB; ++j)
            if (o0 + j < fout) y[(int64_t)(o0 + j) * ld] = acc[j] * d[(int64_t)(o0 + j) * ld];
    }
}

// cotangent: g[i] = dprev[i] * sum_o W(o,i) delta[o]   (dprev == nullptr -> no multiply)
__device__ __forceinline__ void simt_dense_bwd(const float* __restrict__ W, int fin_used, int fout,
                                               const float* __restrict__ delta,
                                               const float* __restrict__ dprev,
                                               float* __restrict__ g, int64_t ld) {
    for (int i0 = 0; i0 < fin_used; i0 += OB) {
        float acc[OB];
#pragma unroll
        for (int j = 0; j < OB; ++j) acc[j] = 0.f;
        for (int o = 0; o < fout; ++o) {
            const float dl = delta[(int64_t)o * ld];
#pragma unroll
            for (int j = 0; j < OB; ++j)
                if (i0 + j < fin_used) acc[j] = fmaf(W[(size_t)fout * (i0 + j) + o], dl, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < OB; ++j)
            if (i0 + j < fin_used) {
                const float m = dprev ? dprev[(int64_t)(i0 + j) * ld] : 1.f;
                g[(int64_t)(i0 + j) * ld] = acc[j] * m;
            }
    }
}

// One dynamics call for every sample of the batch.
//   z = u[0:D] + dt * sum_j coef[j] * k_j[0:D]     (RK stage combination; nprev = 0 -> z = u)
//   du = [zdot; ldot; Edot; ndot]                  (src/core/icnf.jl:517-536 / 561-580 / 297-316)
__global__ void __launch_bounds__(256)
simt_aug_f_kernel(NetDev net, const float* __restrict__ P, StageIn in, float t,
                  const float* __restrict__ eps, const float* __restrict__ ys, int64_t B,
                  float* __restrict__ du, float* __restrict__ ws, int64_t ld) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    const int D = net.D, S = D + 3, N = net.n_layers, maxw = net.maxw;
    float* hA = ws + s;                         // rows [0, maxw)
    float* hB = hA + (int64_t)maxw * ld;        // rows [maxw, 2 maxw)
    float* gA = hB + (int64_t)maxw * ld;
    float* gB = gA + (int64_t)maxw * ld;
    float* dAll = gB + (int64_t)maxw * ld;      // act' of every layer, rows sum(widths[1..N])

    // input rows [z; t; ys]  (src/layers/cond_layer.jl:7-31)
    const float* us = in.u + s * S;
    for (int i = 0; i < D; ++i) {
        float acc = 0.f;
        for (int j = 0; j < in.nprev; ++j) acc = fmaf(in.coef[j], in.k[j][s * S + i], acc);
        hA[(int64_t)i * ld] = fmaf(in.dt, acc, us[i]);
    }
    int r = D;
    if (!net.autonomous) hA[(int64_t)(r++) * ld] = t;
    for (int i = 0; i < net.C; ++i) hA[(int64_t)(r + i) * ld] = ys[s * net.C + i];

    // forward chain
    float* x = hA;
    float* y = hB;
    int doff = 0;
    for (int l = 0; l < N; ++l) {
        simt_dense_fwd(P + net.w_off[l], P + net.b_off[l], net.widths[l], net.widths[l + 1],
                       net.acts[l], x, y, dAll + (int64_t)doff * ld, ld);
        doff += net.widths[l + 1];
        float* tmp = x; x = y; y = tmp;
    }
    const float* zd = x;  // D rows
    float* out = du + s * S;
    float e2 = 0.f;
    for (int i = 0; i < D; ++i) { const float v = zd[(int64_t)i * ld]; out[i] = v; e2 = fmaf(v, v, e2); }

    float ldot = 0.f, ndot = 0.f;
    const int dlast = doff - net.widths[N];  // row offset of the last layer's act'
    if (net.mode == CNF_MODE_HUTCH_JVP) {
        const float invK = 1.f / (float)net.K;
        for (int k = 0; k < net.K; ++k) {
            const float* e = eps + s * ((int64_t)net.K * D) + (int64_t)k * D;
            for (int i = 0; i < net.widths[0]; ++i) gA[(int64_t)i * ld] = i < D ? e[i] : 0.f;
            float* tp = gA; float* tq = gB;
            int off = 0;
            for (int l = 0; l < N; ++l) {
                simt_dense_tan(P + net.w_off[l], net.widths[l], net.widths[l + 1], tp,
                               dAll + (int64_t)off * ld, tq, ld);
                off += net.widths[l + 1];
                float* tmp = tp; tp = tq; tq = tmp;
            }
            float dot = 0.f, n2 = 0.f;
            for (int i = 0; i < D; ++i) {
                const float g = tp[(int64_t)i * ld];
                dot = fmaf(g, e[i], dot);
                n2 = fmaf(g, g, n2);
            }
            ldot -= invK * dot;
            if (net.reg_j) ndot += invK * sqrtf(n2);
        }
    } else {
        // pullbacks: K probes (Hutchinson) or D one-hot seeds (exact trace = sum_i (e_i^T J)_i,
        // the construction of src/core/utils.jl:35-56)
        const bool exact = net.mode == CNF_MODE_EXACT;
        const int nseed = exact ? D : net.K;
        const float invK = exact ? 1.f : 1.f / (float)net.K;
        for (int k = 0; k < nseed; ++k) {
            const float* e = exact ? nullptr : eps + s * ((int64_t)net.K * D) + (int64_t)k * D;
            const float* dN = dAll + (int64_t)dlast * ld;
            for (int i = 0; i < D; ++i) {
                const float seed = exact ? (i == k ? 1.f : 0.f) : e[i];
                gA[(int64_t)i * ld] = seed * dN[(int64_t)i * ld];
            }
            float* dl = gA; float* gp = gB;
            int off = dlast;
            for (int l = N - 1; l >= 0; --l) {
                const int fin_used = l == 0 ? D : net.widths[l];
                const float* dprev = nullptr;
                if (l > 0) { off -= net.widths[l]; dprev = dAll + (int64_t)off * ld; }
                simt_dense_bwd(P + net.w_off[l], fin_used, net.widths[l + 1], dl, dprev, gp, ld);
                float* tmp = dl; dl = gp; gp = tmp;
            }
            if (exact) {
                ldot -= dl[(int64_t)k * ld];
            } else {
                float dot = 0.f, n2 = 0.f;
                for (int i = 0; i < D; ++i) {
                    const float g = dl[(int64_t)i * ld];
                    dot = fmaf(g, e[i], dot);
                    n2 = fmaf(g, g, n2);
                }
                ldot -= invK * dot;
                if (net.reg_j) ndot += invK * sqrtf(n2);
            }
        }
    }
    const bool train = net.mode != CNF_MODE_EXACT;
    out[D] = ldot;
    out[D + 1] = (train && net.reg_z) ? sqrtf(e2) : 0.f;
    out[D + 2] = (train && net.reg_j) ? ndot : 0.f;
}

// u += dt * sum_i b_i k_i, elementwise over the S x B state
__global__ void __launch_bounds__(256)
rk_update_kernel(float* __restrict__ u, StageIn in, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float acc = 0.f;
    for (int j = 0; j < in.nprev; ++j) acc = fmaf(in.coef[j], in.k[j][e], acc);
    u[e] = fmaf(in.dt, acc, in.u[e]);
}

// u0 = vcat(xs, zeros(naug + 3, B))   (src/core/base_icnf.jl:256-266)
__global__ void __launch_bounds__(256)
assemble_u0_kernel(const float* __restrict__ x, int nvars, int S, int64_t B, float* __restrict__ u) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * S) return;
    const int64_t s = e / S;
    const int r = (int)(e - s * S);
    u[e] = r < nvars ? x[s * nvars + r] : 0.f;
}

// inference_sol epilogue (src/core/base_icnf.jl:158-172, 106-122)
__global__ void __launch_bounds__(256)
epilogue_kernel(const float* __restrict__ u, int nvars, int D, int reg_aug, int64_t B,
                float* __restrict__ logp, float* __restrict__ regs) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    const int S = D + 3;
    const float* uc = u + s * S;
    float ss = 0.f, sa = 0.f;
    for (int i = 0; i < D; ++i) ss = fmaf(uc[i], uc[i], ss);
    for (int i = nvars; i < D; ++i) sa = fmaf(uc[i], uc[i], sa);
    logp[s] = (-0.5f * (float)D * kLog2Pi - 0.5f * ss) - uc[D];
    if (regs) {
        regs[s] = uc[D + 1];
        regs[B + s] = uc[D + 2];
        regs[2 * B + s] = reg_aug ? sqrtf(sa) : 0.f;
    }
}

// ---------------------------------------------------------------------------------------
// deterministic loss partial sums (src/core/icnf.jl:628-649): two passes, fixed order.
// ---------------------------------------------------------------------------------------
constexpr int LOSS_BLOCKS = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(256)
loss_partial_kernel(const float* __restrict__ logp, const float* __restrict__ regs, int64_t B,
                    float* __restrict__ partial /* LOSS_BLOCKS x 4 */) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < B;
         s += (int64_t)gridDim.x * blockDim.x) {
        a[0] -= logp[s];
        if (regs) { a[1] += regs[s]; a[2] += regs[B + s]; a[3] += regs[2 * B + s]; }
    }
    __shared__ float sm[4][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float v = wave_sum(a[q]);
        if (lane == 0) sm[w][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        partial[blockIdx.x * 4 + threadIdx.x] =
            (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// mean_out (may be null): (s0 + l1 s1 + l2 s2 + l3 s3) / B in double - the scalar `loss` of an unsharded batch, so that a
// single-process caller needs no further device work after the two reduction kernels
__global__ void __launch_bounds__(256)
loss_final_kernel(const float* __restrict__ partial, float* __restrict__ sums4, float* __restrict__ mean_out, double l1, double l2,
                  double l3, double B) {
    // 256 threads: thread t sums column q = t&3 over blocks t>>2, t>>2 + 64, ...
    const int q = threadIdx.x & 3;
    float v = 0.f;
    for (int b = threadIdx.x >> 2; b < LOSS_BLOCKS; b += 64) v += partial[b * 4 + q];
    __shared__ float sm[256];
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st >= 4; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 4 && sums4) sums4[threadIdx.x] = sm[threadIdx.x];
    if (threadIdx.x == 0 && mean_out) {
        double acc = (double)sm[0];
        acc += (double)sm[1] * l1;
        acc += (double)sm[2] * l2;
        acc += (double)sm[3] * l3;
        mean_out[0] = (float)(acc / B);
    }
}

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
static inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

size_t simt_ws_rows(const NetDev& net) {
    size_t rows = 4 * (size_t)net.maxw;
    for (int l = 1; l <= net.n_layers; ++l) rows += (size_t)net.widths[l];
    return rows;
}

hipError_t simt_aug_f(const NetDev& net, const float* P, const StageIn& in, float t,
                      const float* eps, const float* ys, int64_t B, float* du, float* ws,
                      int64_t ld, hipStream_t st) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(simt_aug_f_kernel, dim3(nblk(B, 256)), dim3(256), 0, st, net, P, in, t, eps,
                       ys, B, du, ws, ld);
    return hipGetLastError();
}

hipError_t rk_update(float* u, const StageIn& in, int64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rk_update_kernel, dim3(nblk(n, 256)), dim3(256), 0, st, u, in, n);
    return hipGetLastError();
}

hipError_t assemble_u0(const float* x, int nvars, int S, int64_t B, float* u, hipStream_t st) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(assemble_u0_kernel, dim3(nblk(B * S, 256)), dim3(256), 0, st, x, nvars, S, B, u);
    return hipGetLastError();
}

hipError_t epilogue(const float* u, int nvars, int D, int reg_aug, int64_t B, float* logp,
                    float* regs, hipStream_t st) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(epilogue_kernel, dim3(nblk(B, 256)), dim3(256), 0, st, u, nvars, D, reg_aug, B,
                       logp, regs);
    return hipGetLastError();
}

// Embedded error estimate of one step (OrdinaryDiffEq: utilde = dt sum_i btilde_i k_i, atmp = utilde ./ (abstol +
// max(|uprev|, |u|) reltol), EEst = sqrt(sum(atmp^2) / length)): squared scaled residuals summed in double,
// fixed partition and combine order (no atomics).
struct ErrIn { const float* k[7]; float coef[7]; int nk; };

// V consecutive elements per thread (V = 4: 16-byte loads when the state size allows), kErrBlocks blocks
template <int V>
__global__ void __launch_bounds__(256)
err_partial_kernel(const float* __restrict__ u, const float* __restrict__ unew, ErrIn in, float dt, float abstol,
                   float reltol, int64_t n, double* __restrict__ partial /* kErrBlocks */) {
    double acc = 0.0;
    for (int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; e < n; e += (int64_t)gridDim.x * blockDim.x * V) {
        float ut[V], a[V], b[V];
#pragma unroll
        for (int i = 0; i < V; ++i) ut[i] = 0.f;
        for (int j = 0; j < in.nk; ++j) {
            if constexpr (V == 4) {
                const float4 k = *reinterpret_cast<const float4*>(in.k[j] + e);
                ut[0] = fmaf(in.coef[j], k.x, ut[0]); ut[1] = fmaf(in.coef[j], k.y, ut[1]);
                ut[2] = fmaf(in.coef[j], k.z, ut[2]); ut[3] = fmaf(in.coef[j], k.w, ut[3]);
            } else {
                ut[0] = fmaf(in.coef[j], in.k[j][e], ut[0]);
            }
        }
        if constexpr (V == 4) {
            const float4 x = *reinterpret_cast<const float4*>(u + e), y = *reinterpret_cast<const float4*>(unew + e);
            a[0] = x.x; a[1] = x.y; a[2] = x.z; a[3] = x.w; b[0] = y.x; b[1] = y.y; b[2] = y.z; b[3] = y.w;
        } else {
            a[0] = u[e]; b[0] = unew[e];
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float r = dt * ut[i] / fmaf(fmaxf(fabsf(a[i]), fabsf(b[i])), reltol, abstol);
            acc += (double)r * (double)r;
        }
    }
    __shared__ double sm[256];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ void __launch_bounds__(256)
err_final_kernel(const double* __restrict__ partial, double* __restrict__ out) {
    __shared__ double sm[256];
    double v = 0.0;
    for (int b = threadIdx.x; b < kErrBlocks; b += 256) v += partial[b];
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

hipError_t embedded_error(const float* u, const float* unew, const float* const* k, const float* btilde, int nk, float dt,
                          float abstol, float reltol, int64_t n, double* partial, double* out, hipStream_t st) {
    ErrIn in{};
    in.nk = nk;
    bool vec = n % 4 == 0 && ((uintptr_t)u % 16 == 0) && ((uintptr_t)unew % 16 == 0);
    for (int j = 0; j < nk; ++j) { in.k[j] = k[j]; in.coef[j] = btilde[j]; vec = vec && ((uintptr_t)k[j] % 16 == 0); }
    if (vec) hipLaunchKernelGGL(err_partial_kernel<4>, dim3(kErrBlocks), dim3(256), 0, st, u, unew, in, dt, abstol, reltol, n, partial);
    else hipLaunchKernelGGL(err_partial_kernel<1>, dim3(kErrBlocks), dim3(256), 0, st, u, unew, in, dt, abstol, reltol, n, partial);
    hipLaunchKernelGGL(err_final_kernel, dim3(1), dim3(256), 0, st, partial, out);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) zero_words_kernel(unsigned* __restrict__ p, size_t nwords) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // 16-byte stores over the aligned body, single words at both ends
    const size_t head = (size_t)((16 - ((uintptr_t)p & 15)) & 15) / 4;
    const size_t h = head < nwords ? head : nwords;
    if (i < h) p[i] = 0u;
    uint4* q = reinterpret_cast<uint4*>(p + h);
    const size_t nq = (nwords - h) / 4;
    for (size_t k = i; k < nq; k += stride) q[k] = uint4{0u, 0u, 0u, 0u};
    const size_t tail0 = h + 4 * nq;
    if (i < nwords - tail0) p[tail0 + i] = 0u;
}

__global__ void __launch_bounds__(256) copy_words_kernel(unsigned* __restrict__ d, const unsigned* __restrict__ s, size_t nwords, int vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {   // both 16-byte aligned: 16-byte body, word tail
        const size_t nq = nwords / 4;
        for (size_t k = i; k < nq; k += stride) reinterpret_cast<uint4*>(d)[k] = reinterpret_cast<const uint4*>(s)[k];
        if (i < nwords - 4 * nq) d[4 * nq + i] = s[4 * nq + i];
    } else {
        for (size_t k = i; k < nwords; k += stride) d[k] = s[k];
    }
}

hipError_t copy_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0 || dst == src) return hipSuccess;
    if ((bytes & 3) || ((uintptr_t)dst & 3) || ((uintptr_t)src & 3)) return hipErrorInvalidValue;
    const size_t nwords = bytes / 4;
    const int vec = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
    size_t nb = ((vec ? nwords / 4 : nwords) + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)nb), dim3(256), 0, st, (unsigned*)dst, (const unsigned*)src, nwords, vec);
    return hipGetLastError();
}

hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) || ((uintptr_t)p & 3)) return hipErrorInvalidValue;
    const size_t nwords = bytes / 4;
    size_t nb = (nwords / 4 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)nb), dim3(256), 0, st, (unsigned*)p, nwords);
    return hipGetLastError();
}

hipError_t loss_sums(const float* logp, const float* regs, int64_t B, float* partial,
                     float* sums4, hipStream_t st) {
    hipLaunchKernelGGL(loss_partial_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, logp, regs, B, partial);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, partial, sums4, (float*)nullptr, 0.0, 0.0, 0.0, 1.0);
    return hipGetLastError();
}

hipError_t loss_mean(const float* logp, const float* regs, int64_t B, float* partial, float* sums4, float* mean_out,
                     const double lam[3], hipStream_t st) {
    hipLaunchKernelGGL(loss_partial_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, logp, regs, B, partial);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, partial, sums4, mean_out, lam[0], lam[1], lam[2], (double)B);
    return hipGetLastError();
}

}  // namespace cnf
