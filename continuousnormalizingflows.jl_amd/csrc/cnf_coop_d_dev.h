// cnf_coop_d_dev.h - device helpers shared by the dealt cooperative kernels (cnf_coop_d.hip: 64-sample super-tiles, every wave an
// owner; cnf_coop_d2.hip: 32-sample super-tiles for 16 .. 24 hidden tiles): the run-time view of the packed image, fragment loads
// with vector tile offsets, explicit residency in the accumulation registers, the activation forms.
#pragma once
#include "cnf_coop_dev.h"

namespace cnf {

// run-time view of the plan's packed image (float offsets of MfmaLayout) and the real tile counts of the configuration
struct DImg {
    int f1z, fh, fN, bN, bh, b1, v_b1, v_w1t, v_bh, v_bN;
    int KPZ;          // k-group pitch of the state-column images (f1z, bN)
    int HTP;          // k-group pitch of the H-column images (fh, bh, fN, b1) = the layout's hidden tiles
    int imgH, vecH;   // floats per hidden image / hidden C vector
    int b;            // left-over hidden tiles: HT_real = 4 A + b
    int KGH, remH;    // real hidden k-groups (= HT_real) and k-steps of the last one (1 .. 4)
    int KGZ, remZ;    // real state k-groups and k-steps of the last one
    int xalias;       // the partial tiles alias the exchange buffer (LDS is short): one more barrier per D-row product
    int ckzr;         // floats per lane of the checkpoint arrays (the plan's ZR: what the reverse sweep strides by)
    int ck_ls, ck_qs; // checkpoint rows: floats between lanes / between a lane's 16-row groups inside a tile of 64 ckzr floats - (ckzr, 4): [tile][lane][ckzr];
                      // (4, 256): [tile][group][lane][4] (KArgs::ck_tiles, written by the dealt forward kernel for the second form's sweeps)
    int f1y, KPC, remC;   // conditioned flows (C <= 16: one k-group): the condition columns' image of layer 1, its k-group pitch, real k-steps (0: none)
    int q_off;        // exact-trace instances: float offset of the Q image (two hidden layers: tr J = act'_2^T Q act'_1), else 0
    int cvn;          // floats of the C-vector section [v_b1, end of v_bN) of the image: staged into LDS once per workgroup
};
struct DArgs {
    KArgs k;
    DImg g;
};

struct DRs {
    __amdgpu_buffer_rsrc_t r;
    unsigned lane16;
};
__device__ __forceinline__ f32x4 dload(const DRs& R, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(R.r, (int)R.lane16, (int)byte_off, 0));
}
// fragment load with the tile's offset in a VECTOR register (lane slot + tile row offset, made opaque so that it stays one) and
// the image + k-group offset in ONE scalar: the k-loops then carry a single scalar add per k-group.  With every fragment's
// offset in its own scalar the scalar file overflows (193 scalar spills) and each load in the k-loops was preceded by a
// v_readlane (a VALU instruction between MFMAs) + s_add + s_nop.
__device__ __forceinline__ f32x4 dloadv(const DRs& R, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(R.r, (int)voff, (int)soff, 0));
}
template <int A>
struct TileOff { unsigned S[A]; unsigned Rr[3]; };
template <int A>
__device__ __forceinline__ TileOff<A> tile_offsets(const DRs& R, int KP, int mtS0, int mtR0, int mtRmax) {
    TileOff<A> t;
#pragma unroll
    for (int m = 0; m < A; ++m) { t.S[m] = R.lane16 + (unsigned)((mtS0 + m) * KP) * 1024u; asm volatile("" : "+v"(t.S[m])); }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int mt = mtR0 + r < mtRmax ? mtR0 + r : mtRmax;   // clamped, not guarded: no control flow around a load
        t.Rr[r] = R.lane16 + (unsigned)(mt * KP) * 1024u;
        asm volatile("" : "+v"(t.Rr[r]));
    }
    return t;
}

// Explicit residency in the accumulation registers.  With one wave per SIMD a wave has 256 architectural + 256 accumulation
// registers; every VALU / MFMA operand of this kernel must be architectural (-amdgpu-mfma-vgpr-form), and what the allocator
// spills it spills by its own cost model - in the first builds an LDS ADDRESS used inside the k-loops went to scratch, and since
// vmcnt retires in order its reload drained every outstanding fragment prefetch (s_waitcnt vmcnt(0) per k-group: +17 % run
// time).  act' of every hidden layer (written once, read once per evaluation) and the Runge-Kutta running sums (touched once per
// stage) are therefore parked by hand: a value constrained to class "a" costs one v_accvgpr_write and one v_accvgpr_read.
__device__ __forceinline__ float park(float x) {
    float a;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(x));
    return a;
}
__device__ __forceinline__ float unpark(float a) {
    float x;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a));
    return x;
}
// An MFMA's result may be read by a VALU instruction only 11 wait states after an 8-pass v_mfma_f32_16x16x4_f32 (MI300 / MI350 ISA,
// "manually inserted wait states"); the compiler's hazard recogniser provides them for instructions it selected, but it does not
// look inside inline asm - so wherever accumulator tiles go STRAIGHT from a product into park(), this goes in between.
// The wait states are tied to the tiles themselves ("+v": the parked values are the asm's outputs), so neither the products'
// MFMAs can sink below them nor the parks rise above them.
__device__ __forceinline__ void mfma_results_fence(f32x4& t0, f32x4& t1, f32x4& t2, f32x4& t3) {
    asm volatile("s_nop 7\n\ts_nop 3" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
}
__device__ __forceinline__ f32x4 park4(const f32x4& v) { return f32x4{park(v[0]), park(v[1]), park(v[2]), park(v[3])}; }
__device__ __forceinline__ f32x4 unpark4(const f32x4& v) { return f32x4{unpark(v[0]), unpark(v[1]), unpark(v[2]), unpark(v[3])}; }

// the last hidden layer: the activation alone now, its derivative FROM the activation after the D-row product has consumed it
// (softplus: sigmoid(a) = 1 - exp(-softplus(a)); tanh: 1 - tanh^2) - nothing of that layer is kept across the product, and the
// rebuild costs what computing it the first time would have (one transcendental).  |error| <= 1.2e-7 absolute, as everything else.
template <int ACT>
__device__ __forceinline__ f32x4 act_only(const f32x4& a) {
    if constexpr (ACT == CNF_ACT_SOFTPLUS) {
        constexpr float kNegLog2e = -1.4426950408889634f, kLn2 = 0.6931471805599453f;
        float e[4], lg[4], mx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(__builtin_fabsf(a[i]) * kNegLog2e);
        const f32x2 one = {1.f, 1.f};
        const f32x2 s0 = f32x2{e[0], e[1]} + one, s1 = f32x2{e[2], e[3]} + one;
        const float sv[4] = {s0[0], s0[1], s1[0], s1[1]};
#pragma unroll
        for (int i = 0; i < 4; ++i) { lg[i] = __builtin_amdgcn_logf(sv[i]); mx[i] = __builtin_fmaxf(a[i], 0.f); }
        const f32x2 ln2 = {kLn2, kLn2};
        const f32x2 h0 = __builtin_elementwise_fma(f32x2{lg[0], lg[1]}, ln2, f32x2{mx[0], mx[1]});
        const f32x2 h1 = __builtin_elementwise_fma(f32x2{lg[2], lg[3]}, ln2, f32x2{mx[2], mx[3]});
        return f32x4{h0[0], h0[1], h1[0], h1[1]};
    } else {
        f32x4 h, d;
        act_tile<ACT>(a, h, d);
        return h;
    }
}
template <int ACT>
__device__ __forceinline__ f32x4 dact_from_h(const f32x4& h) {
    if constexpr (ACT == CNF_ACT_SOFTPLUS) {
        constexpr float kNegLog2e = -1.4426950408889634f;
        f32x4 d;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = 1.f - __builtin_amdgcn_exp2f(h[i] * kNegLog2e);
        return d;
    } else {
        const f32x2 h0 = {h[0], h[1]}, h1 = {h[2], h[3]}, one = {1.f, 1.f};
        const f32x2 d0 = __builtin_elementwise_fma(-h0, h0, one), d1 = __builtin_elementwise_fma(-h1, h1, one);
        return f32x4{d0[0], d0[1], d1[0], d1[1]};
    }
}
template <int ACT>
__device__ __forceinline__ void act_pair(const f32x4& a, f32x4& h, f32x4& d) { act_tile<ACT>(a, h, d); }

void dimg_fill(DImg& G, int H, int D, int L, int HT_lay, int ZR_lay, int CR_lay, int A_inst, int C);   // cnf_coop_d.hip
// host side of the 32-sample form (cnf_coop_d2.hip): (A, ZR) instances for 16 .. 24 hidden tiles
bool coopd2_supported(int HT_real, int L, int KZ, int ACT, int C, int exact = 0);
hipError_t coopd2_launch(int HT_real, int L, int KZ, int ACT, DArgs& a, int num_cus, hipStream_t st);
size_t coopd2_rk_floats(int HT_real, int KZ, int ACT, int num_cus, int exact = 0);

}  // namespace cnf
