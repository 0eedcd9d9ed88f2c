// cnf_grad_dev.h — what the register-accumulator gradient shares between its host side (cnf_grad.hip) and its reverse-sweep kernel
// (cnf_grad2.hip, compiled for one probe and - cnf_grad2_probes.hip - for several): argument block, padded transpose tiles,
// LDS / slab layouts.  See cnf_grad.hip for the method, cnf_grad2.hip for the kernel.
#pragma once
#include "cnf_mfma_kernel.h"

namespace cnf {

struct GArgs {
    const float* packed;   // operand image: f32, no tanh pre-scale, forward + transposed
    const float* ckpt;     // [nsteps+1][ntiles][64][ckpt_zr]  (ckpt_zr = state k-steps of the forward instance)
    int ckpt_zr;
    const float* ckpt_k;   // stage derivatives [step * ns + stage][ntiles][64][ckpt_zr] or null (re-sweep)
    const float* eps;      // (K D) x B: probe k occupies rows k D .. k D + D - 1
    int K;                 // Hutchinson probes
    const float* ys;       // C x B or null
    int C;
    float* slab;           // [waves][GradSlab::TOTAL] floats, zeroed by the host
    float* grad_x;         // optional: dL/dx (nvars x B) = rows 0..nvars-1 of the costate at t0, or null
    long long B;
    int nsteps;
    float t0, dt;
    float probe_w;         // weight of each probe's trace term; 0 = 1 / K (Hutchinson mean), 1 = exact trace from the D unit probes
    const float* tgrid;    // optional (device): nsteps + 1 step times of a non-uniform grid; null = t0 + n dt
    int D, H, n_in, autonomous, nvars;
    float lam1, lam2, lam3;   // weights of Edot, ndot, Adot in the objective (0 = term off)
    int w_off[4], b_off[4];   // Lux offsets of the L+1 <= 4 Dense layers
    Tableau T;
};

// Cross-wave exchange of accumulator-layout tiles through LDS.  A tile is stored with 8 dwords of
// padding per 16-lane group (TS = 280 floats), which makes both transposed fragment reads below
// bank-conflict-free (bank = 8 (lane group) + 4 (k lane group) + register, distinct over a 32-lane half).
constexpr int TS = 280;
__device__ __forceinline__ void tile_store(float* __restrict__ slot, int lane, f32x4 t) {
    *reinterpret_cast<f32x4*>(slot + lane * 4 + (lane >> 4) * 8) = t;
}
template <int MT>
__device__ __forceinline__ void tiles_store(float* __restrict__ dst, int lane, const f32x4 (&t)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) tile_store(dst + mt * TS, lane, t[mt]);
}
// A side: lane (i = lane&15, g = lane>>4) gets, for k-step s, feature rowmap(mt, i) of sample 4s+g
__device__ __forceinline__ void read_frag_A(const float* __restrict__ tile, int lane, float (&f)[4]) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = tile[(i >> 2) * 72 + 16 * s + 4 * g + (i & 3)];
}
// B side: lane (j = lane&15, g) gets, for k-step s, feature 16 nt + j (natural order) of sample 4s+g
__device__ __forceinline__ void read_frag_B(const float* __restrict__ tile, int lane, float (&f)[4]) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = tile[(j & 3) * 72 + 16 * s + 4 * g + (j >> 2)];
}
__device__ __forceinline__ f32x4 outer4(const float (&a)[4], const float (&b)[4], f32x4 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma4(a[s], b[s], acc);
    return acc;
}

// dense-layout D-vector (register s, lane group g <-> feature 4s+g) as one accumulator-layout tile
template <int ZR>
__device__ __forceinline__ f32x4 dense_tile(const float (&v)[ZR]) {
    static_assert(ZR <= 4, "gradient kernel: D <= 16");
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < ZR; ++s) t[s] = v[s];
    return t;
}

template <int MT>
__device__ __forceinline__ void zero_tiles(f32x4 (&t)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) t[m] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int HT, int L, int ZR, int CR, int ACT>
struct GradLds {   // float offsets inside dynamic LDS: operand image, then per-wave transpose scratch
    static constexpr MfmaLayout LAY = MfmaLayout(HT, L, ZR, CR, true, 0);
    static constexpr int DT = (ZR + 3) / 4;
    static constexpr int SCR = (LAY.total + 3) / 4 * 4;
    // exchange region per wave: operand tiles published for the other waves (padded tiles of TS floats);
    // the largest exchange is the hidden-matrix one [A1 | B1 | A2 | B2] x HT tiles
    static constexpr int XCH = SCR;
    static constexpr int XCH_TILES = 4 * HT > 2 * HT + 3 ? 4 * HT : 2 * HT + 3;
    static constexpr int XCH_W = XCH_TILES * TS;
    static constexpr int TAB = XCH + 4 * XCH_W;     // cnf_grad2.hip: the Runge-Kutta tableau by stage, 6 x 16 floats
    static constexpr int TOTAL = TAB + 6 * 16;
};
template <int HT, int L, int ZR, int CR>
struct GradSlab {  // float offsets inside one wave's slab; every image is [mt][nt][lane][4] (accumulator layout)
    static constexpr int DT = (ZR + 3) / 4;
    static constexpr int NT1 = CR > 0 ? 2 : 1;                     // input tiles: [z; t; ...; 1@15] and [y (<= 16)]
    static constexpr int W1 = 0;                                   // [HT][NT1]: H x 16 NT1 input columns
    static constexpr int WH = W1 + HT * NT1 * 256;                       // (L-1) x [HT][HT]
    static constexpr int WN = WH + (L - 1) * HT * HT * 256;        // [DT][HT]
    static constexpr int BH = WN + DT * HT * 256;                  // (L-1) x [HT][1]: column 0 = bias of hidden layer l+1
    static constexpr int BN = BH + (L - 1) * HT * 256;             // [DT][1]: column 0 = bias of the last layer
    static constexpr int TOTAL = BN + DT * 256;
};

// forward chain: h_l, act'_l for every hidden layer
template <int HT, int L, int ZR, int CR, int ACT>
__device__ __forceinline__ void grad_forward(const float* __restrict__ smem, int lane, float t, bool autonomous,
                                             const float (&z)[ZR], const float (&y)[CR > 0 ? CR : 1],
                                             f32x4 (&h)[L][HT], f32x4 (&d)[L][HT]) {
    constexpr MfmaLayout LAY(HT, L, ZR, CR, true, 0);
    const int g = lane >> 4;
    f32x4 acc[HT];
    load_cvec<HT>(smem + LAY.v_b1, g, acc);
    if (!autonomous) {
        f32x4 wt[HT];
        load_cvec<HT>(smem + LAY.v_w1t, g, wt);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) acc[mt] += wt[mt] * t;
    }
    gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{z}, acc);
    if constexpr (CR > 0) gemm_tiles<HT, CR>(smem + LAY.f1y, lane, RegIn<CR>{y}, acc);
#pragma unroll
    for (int l = 0; l < L; ++l) {
        if (l > 0) {
            load_cvec<HT>(smem + LAY.v_bh + (l - 1) * MfmaLayout::vecC(HT), g, acc);
            gemm_tiles<HT, 4 * HT>(smem + LAY.fh + (l - 1) * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{h[l - 1]}, acc);
        }
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float dd;
                h[l][mt][r] = act_fwd<ACT>(acc[mt][r], dd);
                d[l][mt][r] = dd;
            }
    }
}

// the reverse-sweep kernel (cnf_grad2.hip): every wave keeps the whole gradient of its own sample tiles; null = no instance
typedef void (*GradKernel)(GArgs);
GradKernel grad2_kernel(int HT, int L, int ZR, int CR, int ACT);
GradKernel grad2_probes_kernel(int HT, int L, int ZR, int CR, int ACT);   // the same for several probes (cnf_grad2_probes.hip)

}  // namespace cnf
