// cnf_coop_dev.h — device helpers of the workgroup-cooperative kernels (cnf_coop.hip: fused solve for wide hidden layers and
// its tile-split form; cnf_coop_grad.hip: the reverse sweep): weight fragments from the packed image (L2 through buffer loads,
// or LDS), activation fragments from the LDS exchange image, and the product loop over them.
#pragma once
#include "cnf_mfma_kernel.h"   // act_tile, tiles_mul, tile_fma: shared with the per-wave kernel

// A 128-bit buffer store with its offset in a scalar register reads its data registers AFTER it has issued; a VALU write to one of
// them needs wait states in between (GCN / CDNA "VMEM store more than 8 bytes followed by a write of the VGPRs holding the write
// data").  This compiler does not provide them on gfx950: in the 8-tile instances of cnf_coop_grad.hip the instruction behind the
// last operand store zeroed an accumulator that shared the store's fourth data register, and - rarely, under memory back-pressure
// (two workgroups per CU) - the store wrote that zero: 3 000 of 12.6 M entries of X_1, the layer-1 cotangent off by 1e-3
// (found through profiles/nv15_three_way.py, the array comparison of -DCNF_CG_COMPARE_BUILD and the ISA).  The stored vector is an
// input of the asm, so its registers stay allocated across the wait states.
#define CNF_STORE_DATA_HAZARD(v) asm volatile("s_nop 1" ::"v"(v))

namespace cnf {

// A fragments of k-group kg for M-tiles mt0 .. mt0 + M - 1 (global image, 16 B per lane, coalesced); A already points at the lane
// (buffer loads: the image offset of the fragment is wave-uniform and rides in an SGPR, the lane's 16-byte slot in a VGPR
// that never changes - no address VALU in the loop; f32 MFMAs and VALU instructions share the issue slot)
// `wl` != nullptr: the packed image has been staged into LDS (the tile-split small-batch form, NT = 1: with one sample tile a
// k-group is 4 MFMAs per M-tile, far too little to hide an L2 round trip per fragment) and fragments are ds_read_b128.
struct AImg { __amdgpu_buffer_rsrc_t r; unsigned off; unsigned lane16; const float* wl; };
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int M>
__device__ __forceinline__ void coop_load_a(const AImg& A, int mt0, int KG, int kg, f32x4 (&a)[M]) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const unsigned so = A.off + (unsigned)(((mt0 + m) * KG + kg) * 1024);
        if (A.wl) a[m] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(A.wl) + so + A.lane16);
        else a[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(A.r, (int)A.lane16, (int)so, 0));
    }
}
// B fragments of k-group kg for sample tiles nt0 .. nt0 + NQ - 1 (LDS exchange image, conflict-free ds_read_b128)
// (NT = sample tiles of the super-tile = tiles per k-group in the image)
template <int NQ, int NT>
__device__ __forceinline__ void coop_load_b(const f32x4* __restrict__ bimg, int nt0, int kg, int lane, f32x4 (&b)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) b[q] = bimg[(kg * NT + nt0 + q) * 64 + lane];
}

template <int M, int NQ>
__device__ __forceinline__ void coop_frag_mfma(const f32x4 (&a)[M], const f32x4 (&b)[NQ], f32x4 (&acc)[M][NQ]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[m][q] = mfma4(a[m][j], b[q][j], acc[m][q]);
}

// acc[m][q] += A(global image; M-tile mt0+m) * B(LDS image; sample tile nt0+q), over KG k-groups.  `a0` arrives holding the A
// fragments of k-group 0 (requested by the caller one phase earlier).  Two fragment sets ping-pong (k-loop unrolled by 2): the
// loads of k-group kg+1 - A from L2, B from LDS - are issued before the 16 M NQ MFMAs of k-group kg.
template <int M, int NQ, int NT>
__device__ __forceinline__ void coop_gemm(const AImg& A, int mt0, int KG,
                                          const f32x4* __restrict__ bimg, int nt0, int lane,
                                          f32x4 (&a0)[M], f32x4 (&acc)[M][NQ]) {
    f32x4 a1[M], b0[NQ], b1[NQ];
    coop_load_b<NQ, NT>(bimg, nt0, 0, lane, b0);
#pragma clang loop unroll(disable)
    for (int kg = 0; kg < KG; kg += 2) {
        const bool has1 = kg + 1 < KG, has2 = kg + 2 < KG;   // wave-uniform
        if (has1) { coop_load_a<M>(A, mt0, KG, kg + 1, a1); coop_load_b<NQ, NT>(bimg, nt0, kg + 1, lane, b1); }
        coop_frag_mfma<M, NQ>(a0, b0, acc);
        if (has1) {
            if (has2) { coop_load_a<M>(A, mt0, KG, kg + 2, a0); coop_load_b<NQ, NT>(bimg, nt0, kg + 2, lane, b0); }
            coop_frag_mfma<M, NQ>(a1, b1, acc);
        }
    }
}

// The same product with a RUN-TIME k-group count KE (even, <= the image's row pitch KP): a hidden width that fills fewer tiles
// than the instance has skips the zero ones (136 hidden units in a 12-tile instance: 10 of 12 k-groups).  Branch-free: the
// prefetch index is clamped instead of guarded - a load under an `if` is a control-flow join at which the compiler's
// wait-count insertion waits for everything outstanding.
template <int M, int NQ, int NT>
__device__ __forceinline__ void coop_gemm_rt(const AImg& A, int mt0, int KP, int KE,
                                             const f32x4* __restrict__ bimg, int nt0, int lane,
                                             f32x4 (&a0)[M], f32x4 (&acc)[M][NQ]) {
    f32x4 a1[M], b0[NQ], b1[NQ];
    coop_load_b<NQ, NT>(bimg, nt0, 0, lane, b0);
#pragma clang loop unroll(disable)
    for (int kg = 0; kg < KE; kg += 2) {
        const int k2 = kg + 2 < KE ? kg + 2 : KE - 1;
        coop_load_a<M>(A, mt0, KP, kg + 1, a1); coop_load_b<NQ, NT>(bimg, nt0, kg + 1, lane, b1);
        coop_frag_mfma<M, NQ>(a0, b0, acc);
        coop_load_a<M>(A, mt0, KP, k2, a0); coop_load_b<NQ, NT>(bimg, nt0, k2, lane, b0);
        coop_frag_mfma<M, NQ>(a1, b1, acc);
    }
}

template <int MT>
__device__ __forceinline__ void gload_cvec(const float* __restrict__ vec, int mt0, int g, f32x4 (&out)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) out[m] = *reinterpret_cast<const f32x4*>(vec + ((mt0 + m) * 4 + g) * 4);
}

}  // namespace cnf
