// cnf_mfma_dev.h — device-side helpers shared by the MFMA kernel files.
#pragma once
#include "cnf_internal.h"
#include "cnf_mfma_layout.h"

namespace cnf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct KArgs {
    const float* packed;
    const float* x;     // nvars x B, or null
    const float* u0;    // S x B, or null
    const float* eps;   // (K D) x B
    const float* ys;    // C x B
    float* u_out;       // S x B or null
    float* logp;        // B or null
    float* regs;        // 3B or null
    long long B;
    int nsteps;         // 0: single dynamics call at t0, du -> u_out
    float t0, dt;
    int nvars, D, C, reg_z, reg_j, reg_aug, autonomous;
    int K;              // live Hutchinson probes (<= the instance's capacity KP)
    int exact;          // tangent engine only: seeds are the D unit vectors, ldot = -tr J
    int prio_mode;      // 0 none, 1 waves 0..3 high, 2 waves 4.. high (SIMD partners = w, w+4)
    int* queue;         // dynamic tile queue (zeroed before the launch) or null = static stride
    float* ckpt;        // optional: z at the start of every step + final, [step][tile][lane][ZR] (gradient)
    float* kfull;       // optional: all S rows of every stage derivative of ONE step, [stage][B][S] (adaptive attempts)
    float* ckpt_k;      // optional: stage derivatives zdot_i, [step * ns + stage][tile][lane][ZR] (gradient)
    Tableau T;
    float acol[6][5];   // acol[st][i] = T.a[st + 1 + i][st] (0 beyond the last stage): what stage st contributes to the stages after it
    float* ckpt_g;      // optional (cooperative checkpointing solve): g = eps^T J of every stage, laid out like ckpt_k (gradient of |eps^T J|)
    int q_off;          // extended cooperative kernel, exact trace of a two-hidden-layer flow: float offset of the Q image in `packed` (0: none)
    const float* tgrid; // extended cooperative kernel: nsteps + 1 step times on the device (non-uniform grid) or null
    int KH;             // extended cooperative kernel: 16-row tiles the widest hidden layer really fills (<= the instance's HT; 0: all)
    float* rk;          // cnf_coop_d2.hip, 20 .. 24 hidden tiles: per-workgroup ring of the Runge-Kutta running sums (plan-owned) or null
    int use_q;          // mfma_solve2p_kernel: <eps^T J, eps> as a dot with q = W_1[:,0:D] eps where |eps^T J| is not asked for (the plan's
                        // instance hoists q: PRE = 2); 0: through the last product, as the PRE <= 1 instances do
    int ck_tiles;       // the dealt forward kernel's 64-sample form (cnf_coop_d.hip) only: 1 = the checkpoint rows (ckpt, ckpt_k, ckpt_g) in TILE layout
                        // [row][tile][16-row group][lane][4] - one coalesced 1 KB access per group for the wave that owns the tile - instead of
                        // [row][tile][lane][ZR] (64 bytes between lanes at the default architecture's 16 state registers: every 16-byte access
                        // of a wave touches 32 cache lines instead of 8, and the rows are the sweep owners' whole dense phase); same bytes per tile
};

// Device-side step controller (mfma_adaptive_kernel): the whole adaptive Tsit5 solve of a batch that fits the chip's wave
// slots in ONE launch.  Every wave owns one tile of 16 samples; the error norm of an attempt is a grid-wide sum (per-workgroup
// partials in `slots`, an arrival counter, every workgroup re-sums the partials in the same order), after which every wave runs
// the same PI controller on the same number and takes the same accept / reject decision.
struct AArgs {
    float abstol, reltol, t1, dt_init;
    int maxiters, dts_cap;
    float acol[6][6];   // acol[st][i] = A7[st + 1 + i][st], A7 = Tsit5's stage matrix with the FSAL row 6 = b (0 beyond it)
    float b[6];         // weights of the 5th-order solution
    float bt[7];        // b - bhat: the embedded error estimate
    float c[7];
    double* slots;      // grid_sum3: [2 (round parity)][workgroups][6] tagged words
    unsigned epoch;     // 1 .. 65535, a different one for every launch on this scratch buffer: the upper half of a slot's tag and
                        // the value of the abort flag, so that nothing has to be cleared between launches (the host zeroes the
                        // buffer when it is allocated and when the epoch wraps)
    unsigned* counter;  // (unused since the tagged slots; keeps the status words where they were)
    float* dts;         // accepted steps, dts_cap entries
    int* stats;         // naccept, nreject, nf, status (0 ok, 1 non-finite error estimate, 2 maxiters, 3 no initial step, 4 grid sum
                        // timed out), max order, [5] = abort flag raised by the first workgroup whose wait timed out
    int* orders;        // VCABM: order of every accepted step, dts_cap entries
    int ckpt_cap;       // mfma_adaptive_kernel with KArgs::ckpt / ckpt_k: accepted steps the checkpoint arrays hold (step n: z_n in slot n of
                        // ckpt, its six stage derivatives in slots 6 n .. 6 n + 5 of ckpt_k; z at t1 in slot naccept) - the forward half of
                        // the frozen-grid gradient, written by the solve that finds the grid (host_rec[6] = 1: complete)
    int* host_rec;      // pinned host memory (or null): the eight status words, then the first kHostRec accepted steps as {step size
                        // bits, order} pairs - written by the kernel itself, so the host reads its answer after one stream
                        // synchronisation instead of two or three small device-to-host copies (15 - 40 us each at these sizes)
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// sum over the 4 lane groups (lanes l, l^16, l^32, l^48) that hold one sample.  gfx950's row / half swaps do each exchange
// in one VALU instruction: v_permlane16_swap(a, b) exchanges the odd 16-lane rows of a with the even rows of b, so with
// a = b = v the two results are [r0 r0 r2 r2] and [r1 r1 r3 r3] and their sum is v + v(lane ^ 16); v_permlane32_swap does the
// same with the 32-lane halves.  (__shfl_xor compiles to address arithmetic + ds_bpermute_b32 + an LDS round trip per stage:
// ~7 VALU instructions and ~130 cycles of latency for each of the two stages.)
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float group_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned w = __float_as_uint(s);
    const u32x2_t q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}


enum { ENG_VJP = 0, ENG_TAN = 1 };

// C vector (bias, time column): the f32x4 of lane group g in tile mt
template <int MT>
__device__ __forceinline__ void load_cvec(const float* __restrict__ vec, int g, f32x4 (&out)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) out[mt] = *reinterpret_cast<const f32x4*>(vec + (mt * 4 + g) * 4);
}

// cooperative wide-layer kernel (cnf_coop.hip)
bool coop_supported(int HT, int L, int ZR, int CR, int ACT, int engine, int KP, int* ZR_inst, int* HT_inst);
hipError_t coop_launch(int HT, int L, int ZR, int ACT, const KArgs& a, int num_cus, hipStream_t st);
// the cooperative kernel extended to conditions, several probes and the exact trace as unit probes (cnf_coop_x.hip)
bool coopx_supported(int HT, int L, int ZR, int CR, int ACT, int* HT_inst, int* ZR_inst, int* CR_inst);
hipError_t coopx_launch(int HT, int L, int ZR, int CR, int ACT, const KArgs& a, int num_cus, hipStream_t st);
// the same, restricted to the instance compiled for exactly MfmaLayout(HT, L, ZR, CR, true): what a plan of another kernel
// family must use when it runs this kernel on its own packed image (hipErrorNotSupported otherwise)
bool coopx_exact_supported(int HT, int L, int ZR, int CR, int ACT);
hipError_t coopx_launch_exact(int HT, int L, int ZR, int CR, int ACT, const KArgs& a, int num_cus, hipStream_t st);
// the cooperative kernel with its tiles dealt exactly over four owner waves (cnf_coop_d.hip): one-probe VJP solves of extended-kernel
// plans, on the plan's own image (layout MfmaLayout(HT_lay, L, ZR_lay, 0, true)); H / D = the configuration's widest hidden layer / state rows
bool coopd_supported(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int exact, int C);   // exact: the TestMode form (two hidden layers, Q product)
int coopd_supertile(int H, int D, int L, int ACT, int exact);   // 64 or 32 samples per super-tile (the checkpoint arrays' tile count)
hipError_t coopd_launch(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int CR_lay, const KArgs& a, int num_cus, hipStream_t st);
size_t coopd_rk_floats(int H, int D, int L, int ACT, int exact, int num_cus);   // floats of the KArgs::rk ring the serving instance needs (0: none)
// hand-scheduled form of the per-wave solve kernel for one-probe VJP flows without conditions (cnf_mfma2.hip)
bool solve2_supported(int HT, int L, int ZR, int ACT);
hipError_t solve2_launch(int HT, int L, int ZR, int ACT, int nthreads, const KArgs& a, int num_cus, hipStream_t st);
// ... and its two-waves-per-tile form for small batches (mfma_solve2p_kernel)
bool solve2p_supported(int HT, int L, int ZR, int ACT, long long ntiles, int num_cus);
hipError_t solve2p_launch(int HT, int L, int ZR, int ACT, const KArgs& a, hipStream_t st);
// the same kernel with ONE sample tile per workgroup and the images in LDS: the tile-split form for small batches
bool coop_split_supported(int HT, int L, int ZR, int ACT);
hipError_t coop_split_launch(int HT, int L, int ZR, int ACT, const KArgs& a, hipStream_t st);

}  // namespace cnf
