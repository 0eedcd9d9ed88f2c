// cnf_common.h — shared host/device definitions for libcnf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "cnf.h"

namespace cnf {

// ---------------------------------------------------------------------------------------
// Fixed-step explicit Runge-Kutta tableaux, rounded to float (OrdinaryDiffEq converts its
// tableau to T = Float32).  Values: SURVEY.md §8 A4 (Tsitouras 2011); classic RK4.
// ---------------------------------------------------------------------------------------
struct Tableau {
    int ns;          // stage evaluations per step
    float c[6];
    float a[6][6];   // a[i][j], j < i
    float b[6];
};

inline Tableau make_tableau(int alg) {
    Tableau T{};
    if (alg == CNF_ALG_RK4) {
        T.ns = 4;
        const float c[4] = {0.f, 0.5f, 0.5f, 1.f};
        const float b[4] = {1.f / 6.f, 1.f / 3.f, 1.f / 3.f, 1.f / 6.f};
        for (int i = 0; i < 4; ++i) { T.c[i] = c[i]; T.b[i] = b[i]; }
        T.a[1][0] = 0.5f;
        T.a[2][1] = 0.5f;
        T.a[3][2] = 1.f;
    } else {
        T.ns = 6;
        const float c[6] = {0.f, 0.161f, 0.327f, 0.9f, 0.9800255409045097f, 1.f};
        const float b[6] = {0.09646076681806523f, 0.01f, 0.4798896504144996f,
                            1.379008574103742f, -3.290069515436081f, 2.324710524099774f};
        for (int i = 0; i < 6; ++i) { T.c[i] = c[i]; T.b[i] = b[i]; }
        T.a[1][0] = 0.161f;
        T.a[2][0] = -0.008480655492356989f; T.a[2][1] = 0.335480655492357f;
        T.a[3][0] = 2.8971530571054935f;    T.a[3][1] = -6.359448489975075f;
        T.a[3][2] = 4.3622954328695815f;
        T.a[4][0] = 5.325864828439257f;     T.a[4][1] = -11.748883564062828f;
        T.a[4][2] = 7.4955393428898365f;    T.a[4][3] = -0.09249506636175525f;
        T.a[5][0] = 5.86145544294642f;      T.a[5][1] = -12.92096931784711f;
        T.a[5][2] = 8.159367898576159f;     T.a[5][3] = -0.071584973281401f;
        T.a[5][4] = -0.028269050394068383f;
    }
    return T;
}

// ---------------------------------------------------------------------------------------
// Activations.  act_fwd returns h = act(a) and writes d = act'(a).
//   tanh     : h = 2/(1+exp(-2a)) - 1;  d = 1 - h^2
//   softplus : NNlib.softplus(a) = log1p(exp(-|a|)) + relu(a);  d = sigmoid(a)
// Built from v_exp_f32 / v_log_f32 / v_rcp_f32 (about 1 ulp each); absolute error of h and d
// is <= 2e-7, checked against the fp64 oracle in tests/test_parity_gpu.py.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// v_sqrt_f32 alone (1 ulp; a denormal argument gives 0): sqrtf expands to a scaled, Newton-corrected sequence of ~12 VALU
// instructions, and the RNODE regularisers take K + 1 square roots per dynamics call
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// internal: tanh whose pre-activation arrives already multiplied by -2 log2(e) (the factor is folded
// into the forward weight images and biases by mfma_pack), so exp(-2a) is a bare v_exp_f32
constexpr int CNF_ACT_TANH_PRESCALED = 3;
constexpr float kTanhPrescale = -2.8853900817779268f;

template <int ACT>
__device__ __forceinline__ float act_fwd(float a, float& d) {
    if constexpr (ACT == CNF_ACT_TANH_PRESCALED) {
        const float e = __builtin_amdgcn_exp2f(a);
        const float r = fast_rcp(1.f + e);
        const float h = fmaf(2.f, r, -1.f);
        d = fmaf(-h, h, 1.f);
        return h;
    } else if constexpr (ACT == CNF_ACT_TANH) {
        // tanh(a) = 2 sigmoid(2a) - 1: one v_mul, v_exp, v_add, v_rcp, two v_fma.  Saturates
        // correctly (e -> inf gives r = 0, h = -1; e -> 0 gives h = 1); |error| <= 2e-7.
        const float e = __builtin_amdgcn_exp2f(a * -2.8853900817779268f);   // exp(-2a)
        const float r = fast_rcp(1.f + e);
        const float h = fmaf(2.f, r, -1.f);
        d = fmaf(-h, h, 1.f);
        return h;
    } else if constexpr (ACT == CNF_ACT_SOFTPLUS) {
        const float e = fast_exp(-fabsf(a));          // in (0,1]
        const float r = fast_rcp(1.f + e);
        d = a >= 0.f ? r : e * r;                     // sigmoid(a)
        return __logf(1.f + e) + fmaxf(a, 0.f);
    } else {
        d = 1.f;
        return a;
    }
}

__device__ __forceinline__ float act_fwd_rt(int act, float a, float& d) {
    if (act == CNF_ACT_TANH) return act_fwd<CNF_ACT_TANH>(a, d);
    if (act == CNF_ACT_SOFTPLUS) return act_fwd<CNF_ACT_SOFTPLUS>(a, d);
    d = 1.f;
    return a;
}

constexpr float kLog2Pi = 1.8378770664093453f;

// "dynamic LDS above 64 KB enabled for this kernel" flag, one bit per device; host threads may race on
// the first launch (the attribute call is idempotent, the flag updates are atomic)
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
    bool done(int dev) const { return (mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull; }
    void set(int dev) { mask.fetch_or(1ull << (dev & 63), std::memory_order_release); }
};

}  // namespace cnf
