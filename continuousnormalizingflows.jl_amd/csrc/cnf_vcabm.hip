// cnf_vcabm.hip — the elementwise passes of the variable-coefficient Adams PECE step (the reference's default
// `alg = VCABM()`, src/core/icnf.jl:84-89, solved in base_sol, src/core/base_icnf.jl:134-140).
//
// Per element of the S x B state the method keeps the modified divided differences Phi*_j(n-1), j = 0..12 (Hairer,
// Noersett, Wanner I, III.5).  One attempt is   predict -> f(p, t+dt) -> correct(+ three error sums);   an accepted
// step adds   f(u_new, t+dt) -> optional error sum of order k+1.   The f evaluations are the dynamics kernels of the
// handle; the three passes here are HBM-bound streams over (k + 3) .. (2k + 4) vectors of S x B floats each, every
// vector touched once per pass, the whole difference chain of one element kept in registers.
#include "cnf_internal.h"

namespace cnf {

namespace {

constexpr int VC_BLOCKS = 1024;

__device__ __forceinline__ void block_sum3(double a, double b, double c, double* __restrict__ partial) {
    __shared__ double sm[3][256];
    sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = b; sm[2][threadIdx.x] = c;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) {
            sm[0][threadIdx.x] += sm[0][threadIdx.x + st];
            sm[1][threadIdx.x] += sm[1][threadIdx.x + st];
            sm[2][threadIdx.x] += sm[2][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[threadIdx.x * VC_BLOCKS + blockIdx.x] = sm[threadIdx.x][0];
}

// V consecutive elements per thread (V = 4: 16-byte loads and stores when S x B is a multiple of 4, else V = 1)
template <int V> struct Pack { float v[V]; };
template <int V> __device__ __forceinline__ Pack<V> ld(const float* __restrict__ p, int64_t e) {
    Pack<V> r;
    if constexpr (V == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p + e);
        r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
    } else {
        r.v[0] = p[e];
    }
    return r;
}
template <int V> __device__ __forceinline__ void st(float* __restrict__ p, int64_t e, const Pack<V>& r) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(p + e) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    else p[e] = r.v[0];
}

// Phi_0(n) = f_n, Phi_j(n) = Phi_{j-1}(n) - Phi*_{j-1}(n-1), Phi*_j(n) = beta_j Phi_j(n) for j < m;
// p = u + dt sum_{j<k} g_j Phi*_j(n)
template <int V>
__global__ void __launch_bounds__(256)
vc_predict_kernel(const float* __restrict__ f, const float* __restrict__ u, VcCoef c, int64_t n, float* __restrict__ p) {
    for (int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; e < n; e += (int64_t)gridDim.x * blockDim.x * V) {
        Pack<V> phi = ld<V>(f, e), acc;
        st<V>(c.ps_new, e, phi);
#pragma unroll
        for (int i = 0; i < V; ++i) acc.v[i] = c.g[0] * phi.v[i];
        for (int j = 1; j < c.m; ++j) {
            const Pack<V> old = ld<V>(c.ps_old + (size_t)(j - 1) * c.ld, e);
            Pack<V> s;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                phi.v[i] -= old.v[i];
                s.v[i] = c.beta[j] * phi.v[i];
                if (j < c.k) acc.v[i] = fmaf(c.g[j], s.v[i], acc.v[i]);
            }
            st<V>(c.ps_new + (size_t)j * c.ld, e, s);
        }
        const Pack<V> uu = ld<V>(u, e);
        Pack<V> pp;
#pragma unroll
        for (int i = 0; i < V; ++i) pp.v[i] = fmaf(c.dt, acc.v[i], uu.v[i]);
        st<V>(p, e, pp);
    }
}

// Phi_j(n+1) from d = f(p, t+dt);  u_new = p + dt g_k Phi_k(n+1);  squared scaled error sums of orders k, k-1, k-2
template <int V>
__global__ void __launch_bounds__(256)
vc_correct_kernel(const float* __restrict__ d, const float* __restrict__ p, const float* __restrict__ u, VcCoef c,
                  float abstol, float reltol, int64_t n, float* __restrict__ unew, double* __restrict__ partial) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; e < n; e += (int64_t)gridDim.x * blockDim.x * V) {
        Pack<V> phi = ld<V>(d, e), phim1, phim2;
#pragma unroll
        for (int i = 0; i < V; ++i) phim1.v[i] = phim2.v[i] = 0.f;
        for (int j = 1; j <= c.k; ++j) {
            const Pack<V> ps = ld<V>(c.ps_new + (size_t)(j - 1) * c.ld, e);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                phim2.v[i] = phim1.v[i];
                phim1.v[i] = phi.v[i];
                phi.v[i] -= ps.v[i];
            }
        }
        const Pack<V> pp = ld<V>(p, e), uu = ld<V>(u, e);
        Pack<V> un;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            un.v[i] = fmaf(c.dt * c.g[c.k], phi.v[i], pp.v[i]);
            const float inv = 1.f / fmaf(fmaxf(fabsf(uu.v[i]), fabsf(un.v[i])), reltol, abstol);
            const float r0 = c.e0 * phi.v[i] * inv, r1 = c.e1 * phim1.v[i] * inv, r2 = c.e2 * phim2.v[i] * inv;
            a0 += (double)r0 * (double)r0;
            a1 += (double)r1 * (double)r1;
            a2 += (double)r2 * (double)r2;
        }
        st<V>(unew, e, un);
    }
    block_sum3(a0, a1, a2, partial);
}

// Phi_{k+1}(n+1) from f(u_new, t+dt): squared scaled error sum of order k+1 (coefficient c.e0 = dt gamma*_{k+1})
template <int V>
__global__ void __launch_bounds__(256)
vc_errup_kernel(const float* __restrict__ fnew, const float* __restrict__ u, const float* __restrict__ unew, VcCoef c,
                float abstol, float reltol, int64_t n, double* __restrict__ partial) {
    double a0 = 0.0;
    for (int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; e < n; e += (int64_t)gridDim.x * blockDim.x * V) {
        Pack<V> phi = ld<V>(fnew, e);
        for (int j = 0; j <= c.k; ++j) {
            const Pack<V> ps = ld<V>(c.ps_new + (size_t)j * c.ld, e);
#pragma unroll
            for (int i = 0; i < V; ++i) phi.v[i] -= ps.v[i];
        }
        const Pack<V> uu = ld<V>(u, e), un = ld<V>(unew, e);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float r = c.e0 * phi.v[i] / fmaf(fmaxf(fabsf(uu.v[i]), fabsf(un.v[i])), reltol, abstol);
            a0 += (double)r * (double)r;
        }
    }
    block_sum3(a0, 0.0, 0.0, partial);
}

// sum over the state of ((a - b) / (abstol + reltol |u|))^2 (b may be null): the norms of Hairer's initial-step heuristic
__global__ void __launch_bounds__(256)
vc_norm_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ u, float abstol, float reltol,
               int64_t n, double* __restrict__ partial) {
    double a0 = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float r = (b ? a[e] - b[e] : a[e]) / fmaf(fabsf(u[e]), reltol, abstol);
        a0 += (double)r * (double)r;
    }
    block_sum3(a0, 0.0, 0.0, partial);
}

__global__ void __launch_bounds__(256)
vc_final_kernel(const double* __restrict__ partial, int nout, double* __restrict__ out) {
    __shared__ double sm[256];
    for (int q = 0; q < nout; ++q) {
        double v = 0.0;
        for (int b = threadIdx.x; b < VC_BLOCKS; b += 256) v += partial[q * VC_BLOCKS + b];
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int st = 128; st >= 1; st >>= 1) {
            if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[q] = sm[0];
        __syncthreads();
    }
}

}  // namespace

size_t vcabm_partial_doubles() { return 3 * (size_t)VC_BLOCKS; }

hipError_t vcabm_predict(const float* f, const float* u, const VcCoef& c, int64_t n, float* p, hipStream_t st) {
    if (n % 4 == 0) hipLaunchKernelGGL(vc_predict_kernel<4>, dim3(VC_BLOCKS), dim3(256), 0, st, f, u, c, n, p);
    else hipLaunchKernelGGL(vc_predict_kernel<1>, dim3(VC_BLOCKS), dim3(256), 0, st, f, u, c, n, p);
    return hipGetLastError();
}

hipError_t vcabm_correct(const float* d, const float* p, const float* u, const VcCoef& c, float abstol, float reltol,
                         int64_t n, float* unew, double* partial, double* err3, hipStream_t st) {
    if (n % 4 == 0) hipLaunchKernelGGL(vc_correct_kernel<4>, dim3(VC_BLOCKS), dim3(256), 0, st, d, p, u, c, abstol, reltol, n, unew, partial);
    else hipLaunchKernelGGL(vc_correct_kernel<1>, dim3(VC_BLOCKS), dim3(256), 0, st, d, p, u, c, abstol, reltol, n, unew, partial);
    hipLaunchKernelGGL(vc_final_kernel, dim3(1), dim3(256), 0, st, partial, 3, err3);
    return hipGetLastError();
}

hipError_t vcabm_errup(const float* fnew, const float* u, const float* unew, const VcCoef& c, float abstol, float reltol,
                       int64_t n, double* partial, double* err1, hipStream_t st) {
    if (n % 4 == 0) hipLaunchKernelGGL(vc_errup_kernel<4>, dim3(VC_BLOCKS), dim3(256), 0, st, fnew, u, unew, c, abstol, reltol, n, partial);
    else hipLaunchKernelGGL(vc_errup_kernel<1>, dim3(VC_BLOCKS), dim3(256), 0, st, fnew, u, unew, c, abstol, reltol, n, partial);
    hipLaunchKernelGGL(vc_final_kernel, dim3(1), dim3(256), 0, st, partial, 1, err1);
    return hipGetLastError();
}

hipError_t vcabm_scaled_sumsq(const float* a, const float* b, const float* u, float abstol, float reltol, int64_t n,
                              double* partial, double* out1, hipStream_t st) {
    hipLaunchKernelGGL(vc_norm_kernel, dim3(VC_BLOCKS), dim3(256), 0, st, a, b, u, abstol, reltol, n, partial);
    hipLaunchKernelGGL(vc_final_kernel, dim3(1), dim3(256), 0, st, partial, 1, out1);
    return hipGetLastError();
}

}  // namespace cnf
