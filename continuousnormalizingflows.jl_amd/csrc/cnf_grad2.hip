// cnf_grad2.hip — the register-accumulator gradient kernel, barrier-free form (round 5; gfx950).
//
// Same mathematics, operand image, checkpoints, slab layout and reduce kernel as cnf_grad.hip (read that header first; SURVEY.md
// §8(f) rank 2: the reference differentiates `loss` through the solve, src/core/icnf.jl:90-99, src/exts/mlj_ext/core_icnf.jl:42-51).
// What changes is who accumulates what.  cnf_grad.hip partitions every weight cotangent over the workgroup's four waves by row
// block, so every outer product is an exchange: 16 tiles published per wave per matrix, two barriers, every B tile read back by all
// four waves.  Measured with s_memtime stamps (profiles/r5/r5a_cfg2_grad_phase_trace.txt, one stage of cfg2, 41.0 k cycles of which
// 28.8 k are MFMA issue): the four publish phases cost 5.0 k cycles in which no MFMA issues, the eight barriers 2.5 k (wave 0 carries
// the last-layer bias products; LDS contention skews the others), the two small exchanges run at half rate behind LDS latency.
//
// Here EVERY WAVE KEEPS THE WHOLE GRADIENT of its own sample tiles in registers (cfg2: 170 of the 512 a wave owns at one wave per
// SIMD) and nothing crosses waves until the slabs are summed:
//   * no barrier after the image is staged; waves drift, so their LDS bursts no longer collide;
//   * an operand tile is transposed (sample index from the N lanes onto MFMA K) through WAVE-PRIVATE scratch: ds_write_b128 into a
//     padded tile, two conflict-free ds_read2_b32 per fragment; LDS executes a wave's accesses in order, so no fence is needed;
//   * every fragment is read once: 32 fragment reads per hidden matrix per stage instead of 80, 8 tile stores instead of 16;
//   * the second term of a hidden cotangent, delta_{l+1} ubar_l^T, is taken in the bottom-up pass the moment ubar_l exists, the first
//     term, abar_l h_{l-1}^T, in the top-down pass: each phase then has TWO INDEPENDENT MFMA chains - the next chain product (weight
//     fragments from the LDS image) and a cotangent product (operand fragments already in registers) - for the scheduler to interleave;
//   * bias cotangents are row sums of the A fragments that are in registers anyway (four v_add per row tile) instead of outer
//     products with a ones column (64 of 916 MFMAs per stage at cfg2).
// MFMAs per stage per tile at cfg2: 864 (was 900 + 16 on wave 0).
#include "cnf_grad_dev.h"
#include "cnf_sched_dev.h"

// compiled twice: cnf_grad2.hip (one probe) and cnf_grad2_probes.hip (-DG2_MULTI=true: several probes)
#ifndef G2_FIND
#define G2_FIND grad2_kernel
#endif

#ifdef G2_TRACE
#define G2_T(k) do { asm volatile("" ::: "memory"); tr[k] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#else
#define G2_T(k)
#endif

namespace cnf {

namespace {

// Outer-product accumulation of cotangent tiles, K = the tile's 16 samples.  The accumulators are pinned to the ACCUMULATION
// registers ("+a") and multiplied there: with 170 of them live for the whole launch next to ~300 other live values, compiler-managed
// accumulators were copied between the two register halves around every product (660 v_accvgpr_* per stage in the first build, VGPR-
// or AGPR-form MFMAs alike).  One statement = 4 k-steps x up to 4 independent accumulators (a v_mfma_f32_16x16x4_f32 issues every
// 32 cycles and its result is ready for the next accumulate after 40: dependent issues are >= 2 apart wherever a shape has two
// tiles).  Wait states (hipcc pads nothing inside asm): the operands come from ds_read (counted by hipcc) or, at worst, from a
// compiler copy just before the statement - `s_nop 1` opens it; the accumulators are only ever read by the next accumulate (0
// states) until the deposit at the end of the kernel.
#define G2_MF(acc, a, b) "v_mfma_f32_16x16x4_f32 %" #acc ", %" #a ", %" #b ", %" #acc "\n\t"
// row form: W[nt] += sum_s fa[s] fb[nt][s] - one A fragment (row tile), NT B fragments
__device__ __forceinline__ void cot_row(const float (&fa)[4], const float (&fb)[1][4], f32x4 (&W)[1]) {
    asm volatile("s_nop 1\n\t" G2_MF(0, 1, 5) G2_MF(0, 2, 6) G2_MF(0, 3, 7) G2_MF(0, 4, 8)
        : "+a"(W[0])
        : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[0][3]));
}
__device__ __forceinline__ void cot_row(const float (&fa)[4], const float (&fb)[2][4], f32x4 (&W)[2]) {
    asm volatile("s_nop 1\n\t" G2_MF(0, 2, 6) G2_MF(1, 2, 10) G2_MF(0, 3, 7) G2_MF(1, 3, 11) G2_MF(0, 4, 8) G2_MF(1, 4, 12) G2_MF(0, 5, 9) G2_MF(1, 5, 13)
        : "+a"(W[0]), "+a"(W[1])
        : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[0][3]), "v"(fb[1][0]), "v"(fb[1][1]), "v"(fb[1][2]), "v"(fb[1][3]));
}
__device__ __forceinline__ void cot_row(const float (&fa)[4], const float (&fb)[3][4], f32x4 (&W)[3]) {
    asm volatile("s_nop 1\n\t" G2_MF(0, 3, 7) G2_MF(1, 3, 11) G2_MF(2, 3, 15) G2_MF(0, 4, 8) G2_MF(1, 4, 12) G2_MF(2, 4, 16) G2_MF(0, 5, 9) G2_MF(1, 5, 13) G2_MF(2, 5, 17) G2_MF(0, 6, 10) G2_MF(1, 6, 14) G2_MF(2, 6, 18)
        : "+a"(W[0]), "+a"(W[1]), "+a"(W[2])
        : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[0][3]), "v"(fb[1][0]), "v"(fb[1][1]), "v"(fb[1][2]), "v"(fb[1][3]), "v"(fb[2][0]), "v"(fb[2][1]), "v"(fb[2][2]), "v"(fb[2][3]));
}
__device__ __forceinline__ void cot_row(const float (&fa)[4], const float (&fb)[4][4], f32x4 (&W)[4]) {
    asm volatile("s_nop 1\n\t" G2_MF(0, 4, 8) G2_MF(1, 4, 12) G2_MF(2, 4, 16) G2_MF(3, 4, 20) G2_MF(0, 5, 9) G2_MF(1, 5, 13) G2_MF(2, 5, 17) G2_MF(3, 5, 21) G2_MF(0, 6, 10) G2_MF(1, 6, 14) G2_MF(2, 6, 18) G2_MF(3, 6, 22) G2_MF(0, 7, 11) G2_MF(1, 7, 15) G2_MF(2, 7, 19) G2_MF(3, 7, 23)
        : "+a"(W[0]), "+a"(W[1]), "+a"(W[2]), "+a"(W[3])
        : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[0][3]), "v"(fb[1][0]), "v"(fb[1][1]), "v"(fb[1][2]), "v"(fb[1][3]), "v"(fb[2][0]), "v"(fb[2][1]), "v"(fb[2][2]), "v"(fb[2][3]), "v"(fb[3][0]), "v"(fb[3][1]), "v"(fb[3][2]), "v"(fb[3][3]));
}
// column form: W[mt] += sum_s fa[mt][s] fb[s] - MT A fragments, one B fragment (column tile)
__device__ __forceinline__ void cot_col(const float (&fa)[1][4], const float (&fb)[4], f32x4 (&W)[1]) {
    asm volatile("s_nop 1\n\t" G2_MF(0, 5, 1) G2_MF(0, 6, 2) G2_MF(0, 7, 3) G2_MF(0, 8, 4)
        : "+a"(W[0])
        : "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[0][3]));
}
__device__ __forceinline__ void cot_col(const float (&fa)[2][4], const float (&fb)[4], f32x4 (&W)[2]) {
    asm volatile("s_nop 1\n\t" G2_MF(0, 6, 2) G2_MF(1, 10, 2) G2_MF(0, 7, 3) G2_MF(1, 11, 3) G2_MF(0, 8, 4) G2_MF(1, 12, 4) G2_MF(0, 9, 5) G2_MF(1, 13, 5)
        : "+a"(W[0]), "+a"(W[1])
        : "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[0][3]), "v"(fa[1][0]), "v"(fa[1][1]), "v"(fa[1][2]), "v"(fa[1][3]));
}
__device__ __forceinline__ void cot_col(const float (&fa)[3][4], const float (&fb)[4], f32x4 (&W)[3]) {
    asm volatile("s_nop 1\n\t" G2_MF(0, 7, 3) G2_MF(1, 11, 3) G2_MF(2, 15, 3) G2_MF(0, 8, 4) G2_MF(1, 12, 4) G2_MF(2, 16, 4) G2_MF(0, 9, 5) G2_MF(1, 13, 5) G2_MF(2, 17, 5) G2_MF(0, 10, 6) G2_MF(1, 14, 6) G2_MF(2, 18, 6)
        : "+a"(W[0]), "+a"(W[1]), "+a"(W[2])
        : "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[0][3]), "v"(fa[1][0]), "v"(fa[1][1]), "v"(fa[1][2]), "v"(fa[1][3]), "v"(fa[2][0]), "v"(fa[2][1]), "v"(fa[2][2]), "v"(fa[2][3]));
}
__device__ __forceinline__ void cot_col(const float (&fa)[4][4], const float (&fb)[4], f32x4 (&W)[4]) {
    asm volatile("s_nop 1\n\t" G2_MF(0, 8, 4) G2_MF(1, 12, 4) G2_MF(2, 16, 4) G2_MF(3, 20, 4) G2_MF(0, 9, 5) G2_MF(1, 13, 5) G2_MF(2, 17, 5) G2_MF(3, 21, 5) G2_MF(0, 10, 6) G2_MF(1, 14, 6) G2_MF(2, 18, 6) G2_MF(3, 22, 6) G2_MF(0, 11, 7) G2_MF(1, 15, 7) G2_MF(2, 19, 7) G2_MF(3, 23, 7)
        : "+a"(W[0]), "+a"(W[1]), "+a"(W[2]), "+a"(W[3])
        : "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[0][3]), "v"(fa[1][0]), "v"(fa[1][1]), "v"(fa[1][2]), "v"(fa[1][3]), "v"(fa[2][0]), "v"(fa[2][1]), "v"(fa[2][2]), "v"(fa[2][3]), "v"(fa[3][0]), "v"(fa[3][1]), "v"(fa[3][2]), "v"(fa[3][3]));
}
template <int MT, int NT>
__device__ __forceinline__ void cot_block(const float (&fa)[MT][4], const float (&fb)[NT][4], f32x4 (&W)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cot_row(fa[mt], fb, W[mt]);
}

// The transposed fragment reads of cnf_grad_dev.h as four single-dword reads from ONE lane base (a register per access pattern,
// computed once) with the tile's offset in the 16-bit immediate.  Left to the load/store optimiser they become two ds_read2_b32 whose
// 8-bit dword offsets cannot reach the next tile, i.e. one v_add per fragment - a VALU instruction in the middle of an MFMA run costs
// its issue plus a ~9-cycle round trip on gfx950, an LDS instruction in the MFMA's shadow nothing.  (volatile: not merged.)
// tanh of one accumulator tile WITHOUT its derivative (act_tile's asm always computes both; this kernel re-derives act' = 1 - h^2
// where it is used): 2 / (1 + exp(-2 a)) - 1 with bare v_exp / v_rcp and the affine steps on register pairs
__device__ __forceinline__ void tanh_tile_h(const f32x4& a, f32x4& h) {
    const f32x2 x0 = f32x2{a[0], a[1]} * kTanhPrescale, x1 = f32x2{a[2], a[3]} * kTanhPrescale;
    f32x2 e0 = {__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1])};
    f32x2 e1 = {__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1])};
    pk_add1(e0, e1);
    const f32x2 r0 = {fast_rcp(e0[0]), fast_rcp(e0[1])}, r1 = {fast_rcp(e1[0]), fast_rcp(e1[1])};
    f32x2 h0, h1;
    asm volatile("s_nop 0\n\t"
                 "v_pk_fma_f32 %0, %2, 2.0, -1.0 op_sel_hi:[1,0,0]\n\t"
                 "v_pk_fma_f32 %1, %3, 2.0, -1.0 op_sel_hi:[1,0,0]"
                 : "=&v"(h0), "=&v"(h1) : "v"(r0), "v"(r1));
    h = f32x4{h0[0], h0[1], h1[0], h1[1]};
}
// one - h .* h on register pairs, the negation as an operand modifier (the vector expression compiled to a v_xor per pair)
__device__ __forceinline__ f32x4 one_minus_sq(const f32x4& h, float one) {
    const f32x2 lo = {h[0], h[1]}, hi = {h[2], h[3]}, o = {one, one};
    f32x2 d0, d1;
    asm("v_pk_fma_f32 %0, %2, %2, %4 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %3, %3, %4 neg_lo:[1,0,0] neg_hi:[1,0,0]"
        : "=&v"(d0), "=&v"(d1) : "v"(lo), "v"(hi), "v"(o));
    return f32x4{d0[0], d0[1], d1[0], d1[1]};
}
typedef __attribute__((address_space(3))) float LdsF;
__device__ __forceinline__ void read_frag_A1(const float* tile, int lane, float (&f)[4]) {
    const int i = lane & 15, g = lane >> 4;
    const volatile LdsF* p = (const volatile LdsF*)(tile + (i >> 2) * 72 + 4 * g + (i & 3));
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = p[16 * s];
}
__device__ __forceinline__ void read_frag_B1(const float* tile, int lane, float (&f)[4]) {
    const int j = lane & 15, g = lane >> 4;
    const volatile LdsF* p = (const volatile LdsF*)(tile + (j & 3) * 72 + 4 * g + (j >> 2));
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = p[16 * s];
}
template <int MT>
__device__ __forceinline__ void frags_A(const float* tiles, int lane, float (&f)[MT][4]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) read_frag_A(tiles + mt * TS, lane, f[mt]);
}
template <int NT>
__device__ __forceinline__ void frags_B(const float* tiles, int lane, float (&f)[NT][4]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) read_frag_B(tiles + nt * TS, lane, f[nt]);
}


// what stage st reads of the Runge-Kutta tableau: b_st, c_st, row st of a (its stage state) and column st (the adjoint of the later stages)
struct StageCoef {
    float b, c, arow[5], acol[5];
};
__device__ __forceinline__ StageCoef stage_coef(const float* tab, int st) {   // tab: the LDS table [6][16] (three uniform ds_read_b128)
    const f32x4* t = reinterpret_cast<const f32x4*>(tab + st * 16);
    const f32x4 t0 = t[0], t1 = t[1], t2 = t[2];
    StageCoef s;
    s.b = t0[0]; s.c = t0[1];
    s.arow[0] = t0[2]; s.arow[1] = t0[3]; s.arow[2] = t1[0]; s.arow[3] = t1[1]; s.arow[4] = t1[2];
    s.acol[0] = t1[3]; s.acol[1] = t2[0]; s.acol[2] = t2[1]; s.acol[3] = t2[2]; s.acol[4] = t2[3];
    return s;
}

}  // namespace

// MULTI: several Hutchinson probes (a.K of them; the probes share the forward chain and the top-down
// pass, the pullback and its bottom-up reverse run once per probe and a2_l = sum_k dbar_l^k .* u_l^k).  The probe loop is rolled:
// eps_k is read from global memory one probe ahead, c_k = W_N^T eps_k is multiplied per probe (HT x ZR MFMAs of ~450 per probe).
template <int HT, int L, int ZR, int CR, int ACT, bool MULTI>
__global__ void __launch_bounds__(256)
mfma_grad2_kernel(GArgs a) {
    using G = GradLds<HT, L, ZR, CR, ACT>;
    using SL = GradSlab<HT, L, ZR, CR>;
    constexpr MfmaLayout LAY(HT, L, ZR, CR, true, 0);
    constexpr int DT = G::DT;
    constexpr int NH = L - 1;   // hidden (H x H) matrices
    static_assert(DT == 1, "gradient kernel: D <= 16");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.packed);
        f32x4* dst = reinterpret_cast<f32x4*>(smem);
        for (int i = threadIdx.x; i < LAY.total / 4; i += 256) dst[i] = src[i];
    }
    // The tableau by stage - [b_st, c_st, a[st][0..4], a[1..5][st]] - in LDS: read per stage at a run-time index, and as SCALAR loads
    // (kernel-argument segment) those shared the out-of-order lgkmcnt counter with every LDS read of the stage's prologue: each
    // fragment wait became lgkmcnt(0) and the prologue's twelve LDS reads ran one round trip after the other.
    if (threadIdx.x < 6 * 16) {
        const int st = threadIdx.x >> 4, k = threadIdx.x & 15;
        float v = 0.f;
        if (k == 0) v = a.T.b[st];
        else if (k == 1) v = a.T.c[st];
        else if (k < 7) v = a.T.a[st][k - 2];
        else if (k < 12) v = a.T.a[k - 6][st];
        smem[G::TAB + threadIdx.x] = v;
    }
    __syncthreads();   // the only barrier of the kernel
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* slab = a.slab + ((long long)blockIdx.x * 4 + wave) * SL::TOTAL;
    float* scr = smem + G::XCH + wave * G::XCH_W;   // wave-private transpose scratch: XCH_TILES padded tiles
    const long long ntiles = (a.B + 15) / 16;
    const int D = a.D, K = MULTI ? a.K : 1;
    const float invK = MULTI ? (a.probe_w > 0.f ? a.probe_w : 1.f / (float)K) : 1.f;
    const bool autonomous = a.autonomous;
    const float dt0 = a.dt;
    const int ns = a.T.ns;

    // the whole gradient of this wave's sample tiles, accumulator layout ([out tile][in tile], GradSlab's images)
    f32x4 Wh[NH > 0 ? NH : 1][HT][HT], W1in[HT], W1y[CR > 0 ? HT : 1], WNacc[1][HT];
    float bh[NH][HT], bN[ZR];   // bias partial sums: A-fragment lanes (row i, samples == g mod 4) / dense layout (per sample lane)
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) { zero_tiles<HT>(Wh[l][mt]); bh[l][mt] = 0.f; }
    zero_tiles<HT>(W1in);
    zero_tiles<(CR > 0 ? HT : 1)>(W1y);
    zero_tiles<HT>(WNacc[0]);
#pragma unroll
    for (int s = 0; s < ZR; ++s) bN[s] = 0.f;

    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const long long smp = tile * 16 + n;
        const bool valid = smp < a.B;
        const long long sc = valid ? smp : a.B - 1;
        float eps[ZR], lam[ZR];
#pragma unroll
        for (int s = 0; s < ZR; ++s) {
            const int f = 4 * s + g;
            eps[s] = f < D ? a.eps[sc * K * D + f] : 0.f;   // probe 0
            // dL/dz_N = z_N  (L = sum_j -logp_j, -log N(z) = |z|^2/2 + const); zero for padding columns
            lam[s] = valid ? a.ckpt[(((long long)a.nsteps * ntiles + tile) * 64 + lane) * a.ckpt_zr + s] : 0.f;
        }
        if (a.lam3 != 0.f) {   // + l3 |z_aug|_2 at the final time (src/core/base_icnf.jl:106-122)
            float sa = 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) sa = fmaf(lam[s], lam[s], sa); }
            sa = group_sum(sa);
            const float inv = sa > 0.f ? a.lam3 * rsqrtf(sa) : 0.f;
#pragma unroll
            for (int s = 0; s < ZR; ++s) { const int f = 4 * s + g; if (f >= a.nvars && f < D) lam[s] = fmaf(inv, lam[s], lam[s]); }
        }
        float y[CR > 0 ? CR : 1];
        y[0] = 0.f;
        if constexpr (CR > 0) {
#pragma unroll
            for (int s = 0; s < CR; ++s) { const int f = 4 * s + g; y[s] = f < a.C ? a.ys[sc * a.C + f] : 0.f; }
        }
        // c = W_N^T eps: constant over the solve.  (q = W_1[:,0:D] eps is too, but 16 more registers live across every stage spill:
        // dbar_1 = W_1[:,0:D] gbar is multiplied per stage instead - HT x ZR MFMAs of ~870)
        f32x4 cvec[HT];
        zero_tiles<HT>(cvec);
        if constexpr (!MULTI) gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps}, cvec);
        // constant over the solve too: the A fragment of eps (Wbar_N += eps cbar^T) and the B fragment of the conditions (Wbar_1's y columns)
        float fe[1][4], fy[1][4];
        {
            fe[0][0] = fe[0][1] = fe[0][2] = fe[0][3] = 0.f;
            if constexpr (!MULTI) {
                f32x4 et = dense_tile<ZR>(eps);
                tile_store(scr, lane, et);
                frags_A<1>(scr, lane, fe);
            }
            fy[0][0] = fy[0][1] = fy[0][2] = fy[0][3] = 0.f;
            if constexpr (CR > 0) {
                f32x4 yt = dense_tile<(CR > 0 ? CR : 1)>(y);
                tile_store(scr + TS, lane, yt);
                frags_B<1>(scr + TS, lane, fy);
            }
        }

        // step checkpoints (z_n and the stage derivatives kz_i, z rows, written by the forward kernel): requested one STEP ahead -
        // a step opened with an exposed HBM round trip otherwise
        float zn[ZR], kz[6][ZR];
        auto load_ckpt = [&](int step, float (&z)[ZR], float (&k)[6][ZR]) {
#pragma unroll
            for (int s = 0; s < ZR; ++s) z[s] = a.ckpt[(((long long)step * ntiles + tile) * 64 + lane) * a.ckpt_zr + s];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int s = 0; s < ZR; ++s)
                    k[j][s] = j < ns ? a.ckpt_k[((((long long)step * ns + j) * ntiles + tile) * 64 + lane) * a.ckpt_zr + s] : 0.f;
        };
        load_ckpt(a.nsteps - 1, zn, kz);
#pragma clang loop unroll(disable)
        for (int step = a.nsteps - 1; step >= 0; --step) {
            float tn = a.t0 + (float)step * dt0, dt = dt0;
            if (a.tgrid) { tn = a.tgrid[step]; dt = a.tgrid[step + 1] - tn; }
            float zn_nx[ZR], kz_nx[6][ZR];
            load_ckpt(step > 0 ? step - 1 : 0, zn_nx, kz_nx);
            float Zb[6][ZR];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int s = 0; s < ZR; ++s) Zb[j][s] = 0.f;
            // the tableau entries of a stage are read at a run-time index: requested one stage ahead
            StageCoef sc_cur = stage_coef(smem + G::TAB, ns - 1);
#pragma clang loop unroll(disable)
            for (int st = ns - 1; st >= 0; --st) {
#ifdef G2_TRACE
                unsigned long long tr[16];
#endif
                G2_T(0);
                const StageCoef sc_nxt = stage_coef(smem + G::TAB, st > 0 ? st - 1 : 0);
                float zs[ZR], kbar[ZR];
                const float bi = sc_cur.b;
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    float acc = 0.f, kb = bi * lam[s];
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc = fmaf(sc_cur.arow[j], kz[j][s], acc);
#pragma unroll
                    for (int j = 1; j < 6; ++j) kb = fmaf(sc_cur.acol[j - 1], Zb[j][s], kb);   // a[j][st] != 0 only for j > st
                    zs[s] = fmaf(dt, acc, zn[s]);
                    kbar[s] = dt * kb;
                }
                const float cl = valid ? dt * bi : 0.f;   // cotangent of ldot: dL/d(dlogp) = +1
                const float cE = cl * a.lam1, cn = cl * a.lam2;   // cotangents of Edot, ndot
                const bool regz = a.lam1 != 0.f, regj = a.lam2 != 0.f;   // wave-uniform
                const float tt = tn + sc_cur.c * dt;
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                const float* sm = smem + opaque;
                float* sc0 = scr + opaque;
                float* const sE = sc0 + (NH + 1) * HT * TS;   // the "early" slots behind delta_{l+1} (NH groups of HT tiles) and ubar (HT)
                auto IMG_F = [&](int l) { return sm + LAY.fh + l * MfmaLayout::imgA(HT, HT); };   // W_{l+2}
                auto IMG_B = [&](int l) { return sm + LAY.bh + l * MfmaLayout::imgA(HT, HT); };   // W_{l+2}^T
                const float* const no_img = nullptr;

                // ---- prologue: layer-1 fragments and bias requested
                f32x4 nf[HT], nz[1], acc[HT];
                afrag<HT, (ZR < 4 ? ZR : 4)>(sm + LAY.f1z, lane, LAY.KGZ, 0, nf);
                // C vectors: one lane base a stage, at the vectors' own region (the image is larger than a 16-bit immediate reaches)
                const float* smg = sm + 4 * g + LAY.v_b1;
                asm volatile("" : "+v"(smg));
                load_cvec_g<HT>(smg, 0, acc);
                f32x4 wt[HT];
                if (!autonomous) load_cvec_g<HT>(smg, LAY.v_w1t - LAY.v_b1, wt);
                if (!autonomous) {
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) acc[mt] += wt[mt] * tt;
                }
                G2_FENCE();
                // ---- (1) recompute the forward chain: h_l, act'_l
                // (tanh: act' = 1 - h^2 is re-derived from h at each of its three uses - 24 packed FMAs a stage for 48 registers that
                //  otherwise spill around the bottom-up products; softplus keeps its act' = sigmoid)
                // (softplus: act' = sigmoid(a) = 1 - e^-h costs a transcendental per element to re-derive, so it is kept - except in the
                //  widest instances, 4 tiles x 3 layers, where keeping it spills up to 127 registers and one instance does not compile)
                constexpr bool KEEP_D = ACT != CNF_ACT_TANH && !(HT == 4 && L == 3);
                f32x4 h[L][HT], d[KEEP_D ? L : 1][HT];
                // (`one` is an opaque 1.0 of the calling phase: with a literal the three derivations are one common subexpression
                //  and the value is kept alive after all)
                auto dact = [&](int l, int mt, float one) -> f32x4 {
                    if constexpr (KEEP_D) return d[l][mt];
                    else if constexpr (ACT == CNF_ACT_TANH) return one_minus_sq(h[l][mt], one);
                    else {   // softplus: 1 - exp(-h) with the bare v_exp (h >= 0: the argument is <= 0)
                        const f32x4 x = h[l][mt] * (-1.4426950408889634f * one);
                        return f32x4{one - __builtin_amdgcn_exp2f(x[0]), one - __builtin_amdgcn_exp2f(x[1]), one - __builtin_amdgcn_exp2f(x[2]), one - __builtin_amdgcn_exp2f(x[3])};
                    }
                };
                auto opaque_one = [&]() { float o = 1.f; asm volatile("" : "+v"(o)); return o; };
                gemm_pf<HT, ZR, HT>(sm + LAY.f1z, lane, RegIn<ZR>{zs}, nf, acc, no_img, 0, nf);
                if constexpr (CR > 0) gemm_tiles<HT, CR>(sm + LAY.f1y, lane, RegIn<CR>{y}, acc);
                static_for<0, L>([&](auto lc) {
                    constexpr int l = decltype(lc)::value;
                    f32x4 accn[HT];
                    if constexpr (l + 1 < L) { afrag<HT>(IMG_F(l), lane, HT, 0, nf); load_cvec_g<HT>(smg, LAY.v_bh - LAY.v_b1 + l * MfmaLayout::vecC(HT), accn); }
                    else if constexpr (MULTI) afrag<HT, (ZR < 4 ? ZR : 4)>(sm + LAY.bN, lane, LAY.KGZ, 0, nf);   // c_0 = W_N^T eps_0
                    else afrag<HT>(IMG_B(NH - 1), lane, HT, 0, nf);   // the first pullback product
                    G2_FENCE();
                    if constexpr (HT == 4 && ACT == CNF_ACT_TANH) {
                        f32x4 dd[4];
                        tanh_tiles4<false, false>(acc, h[l], dd);   // all four tiles stage by stage: one wait-state statement per packed step
                    } else {
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) {
                            if constexpr (KEEP_D) act_tile<ACT>(acc[mt], h[l][mt], d[l][mt]);
                            else if constexpr (ACT == CNF_ACT_TANH) tanh_tile_h(acc[mt], h[l][mt]);
                            else { f32x4 dd; act_tile<ACT>(acc[mt], h[l][mt], dd); }
                        }
                    }
                    G2_FENCE();
                    if constexpr (l + 1 < L) {
                        gemm_pf<HT, 4 * HT, HT>(IMG_F(l), lane, TileIn<HT>{h[l]}, nf, accn, no_img, 0, nf);
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) acc[mt] = accn[mt];
                    }
                });
                G2_T(1);
                if (regz) {   // Edot = |zdot|: kbar += c_E zdot / |zdot|
                    f32x4 zacc[DT];
                    load_cvec<DT>(sm + LAY.v_bN, g, zacc);
                    gemm_tiles<DT, 4 * HT>(sm + LAY.fN, lane, TileIn<HT>{h[L - 1]}, zacc);
                    float e2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) e2 = fmaf(zacc[s >> 2][s & 3], zacc[s >> 2][s & 3], e2);
                    e2 = group_sum(e2);
                    const float inv = e2 > 0.f ? cE * rsqrtf(e2) : 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) kbar[s] = fmaf(inv, zacc[s >> 2][s & 3], kbar[s]);
                }
                // ---- (2) first-order pullback.  LDS traffic in the products' shadows: delta_{l+1} (l >= 1) goes to scratch group l - 1 while
                //      it is being multiplied (the A operand of Wbar_{l+1}'s second term); the first product also carries h_L and kbar
                //      out and their fragments back, and Wbar_N += kbar h_L^T rides behind it
                float fhL[HT][4], fk[1][4];
                f32x4 db[HT], a2[L][HT];   // a2_l = (sum over the probes of) dbar_l .* u_l  (multiplies act''_l later)
                float gbar[ZR];
                float* const sU = sc0 + NH * HT * TS;   // ubar_l, then cbar (several probes: eps_k and gbar_k pass through its first tile)
                const f32x4 kt = dense_tile<ZR>(kbar);
                if constexpr (MULTI) {
#pragma unroll
                    for (int l = 0; l < L; ++l) zero_tiles<HT>(a2[l]);
                }
                float epsk[ZR];   // the running probe (one probe: eps)
#pragma unroll
                for (int s = 0; s < ZR; ++s) epsk[s] = eps[s];
#pragma clang loop unroll(disable)
                for (int k = 0; k < K; ++k) {
                f32x4 u[NH][HT], dl[HT], ck[HT];
                float eps_nx[ZR];   // several probes: the next one, requested now
                float fek[1][4];
                if constexpr (MULTI) {
                    const int kn = k + 1 < K ? k + 1 : 0;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) {
                        const int f = 4 * s + g;
                        eps_nx[s] = f < D ? a.eps[(sc * K + kn) * D + f] : 0.f;
                    }
                    // c_k = W_N^T eps_k; eps_k out and its A fragment back in the product's shadow (Wbar_N += eps_k cbar_k^T)
                    const f32x4 et = dense_tile<ZR>(epsk);
                    G2_FENCE();
                    zero_tiles<HT>(ck);
                    gemm_pf<HT, ZR, HT>(sm + LAY.bN, lane, RegIn<ZR>{epsk}, nf, ck, IMG_B(NH - 1), HT, nf,
                                        [&](auto qc) {
                                            constexpr int q = decltype(qc)::value;
                                            if constexpr (q == 0) tile_store(sU, lane, et);
                                            if constexpr (q == 1) read_frag_A1(sU, lane, fek[0]);
                                        });
                    G2_FENCE();
                } else {
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) ck[mt] = cvec[mt];
#pragma unroll
                    for (int s = 0; s < 4; ++s) fek[0][s] = fe[0][s];
                }
                const float one_p = opaque_one();
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) dl[mt] = ck[mt] * dact(L - 1, mt, one_p);
                static_for<0, NH>([&](auto lc) {
                    constexpr int l = L - 1 - decltype(lc)::value;   // L-1 .. 1
                    G2_FENCE();
                    zero_tiles<HT>(u[l - 1]);
                    gemm_pf<HT, 4 * HT, HT, (l > 1 ? 4 : ZR)>(IMG_B(l - 1), lane, TileIn<HT>{dl}, nf, u[l - 1], l > 1 ? IMG_B(l > 1 ? l - 2 : 0) : sm + LAY.f1z, l > 1 ? HT : LAY.KGZ, nf,
                                            [&](auto qc) {
                                                constexpr int q = decltype(qc)::value;
                                                if constexpr (q < HT) tile_store(sc0 + ((l - 1) * HT + q) * TS, lane, dl[q]);
                                                if constexpr (l == L - 1 && !MULTI) {
                                                    if constexpr (q >= HT && q < 2 * HT) tile_store(sE + (q - HT) * TS, lane, h[L - 1][q - HT]);
                                                    if constexpr (q >= 2 * HT && q < 3 * HT) read_frag_B1(sE + (q - 2 * HT) * TS, lane, fhL[q - 2 * HT]);
                                                    if constexpr (q == HT) tile_store(sc0 + NH * HT * TS, lane, kt);   // (ubar's slot: free until the bottom-up pass)
                                                    if constexpr (q == 2 * HT) read_frag_A1(sc0 + NH * HT * TS, lane, fk[0]);
                                                }
                                            });
                    if constexpr (l == L - 1 && !MULTI) {
                        cot_row(fk[0], fhL, WNacc[0]);
#pragma unroll
                        for (int s = 0; s < ZR; ++s) bN[s] += kbar[s];
                    }
                    G2_FENCE();
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) dl[mt] = u[l - 1][mt] * dact(l - 1, mt, one_p);
                });
                G2_T(2);
                // gbar = cotangent of g = eps^T J (dense layout): -c_l eps (+ c_n g/|g|);  dbar_1 = W_1[:,0:D] gbar
                const float clk = cl * invK, cnk = cn * invK;   // (several probes: each carries 1 / K of the trace terms)
#pragma unroll
                for (int s = 0; s < ZR; ++s) gbar[s] = -clk * epsk[s];
                if (regj) {
                    f32x4 gacc[DT];
                    zero_tiles<DT>(gacc);
                    gemm_tiles<DT, 4 * HT>(sm + LAY.b1, lane, TileIn<HT>{dl}, gacc);   // g = W_1[:,0:D]^T delta_1
                    float n2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) n2 = fmaf(gacc[s >> 2][s & 3], gacc[s >> 2][s & 3], n2);
                    n2 = group_sum(n2);
                    const float inv = n2 > 0.f ? cnk * rsqrtf(n2) : 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) gbar[s] = fmaf(inv, gacc[s >> 2][s & 3], gbar[s]);
                }
                // ---- (3) bottom-up through the pullback; Wbar_{l+2} += delta_{l+2} ubar_l^T behind the product that consumes ubar_l.
                //      delta_1 waits in the early slots for the stage's last phase (Wbar_1 += delta_1 [gbar; 0]^T)
                const float one_b = opaque_one();
                tiles_store<HT>(sE, lane, dl);
                float fd[HT][4], fg[1][4];   // several probes: delta_1^k and gbar_k, Wbar_1 += delta_1^k [gbar_k; 0]^T behind the bottom-up pass
                const f32x4 gtk = dense_tile<ZR>(gbar);
                G2_FENCE();
                zero_tiles<HT>(db);
                gemm_pf<HT, ZR, HT>(sm + LAY.f1z, lane, RegIn<ZR>{gbar}, nf, db, IMG_F(0), HT, nf,   // dbar_1 = W_1[:,0:D] gbar
                                    [&](auto qc) {
                                        constexpr int q = decltype(qc)::value;
                                        if constexpr (MULTI && q == 0) tile_store(sU, lane, gtk);
                                        if constexpr (MULTI && q == 1) read_frag_B1(sU, lane, fg[0]);
                                    });
                static_for<0, NH>([&](auto lc) {
                    constexpr int l = decltype(lc)::value;
                    f32x4 ubs[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) {
                        ubs[mt] = db[mt] * dact(l, mt, one_b);
                        if constexpr (MULTI) a2[l][mt] += db[mt] * u[l][mt];
                        else a2[l][mt] = db[mt] * u[l][mt];
                    }
                    float fa[HT][4], fb[HT][4];
                    G2_FENCE();
                    zero_tiles<HT>(db);
                    gemm_pf<HT, 4 * HT, HT, (l + 1 < NH ? 4 : ZR)>(IMG_F(l), lane, TileIn<HT>{ubs}, nf, db, l + 1 < NH ? IMG_F(l + 1 < NH ? l + 1 : 0) : sm + LAY.bN, l + 1 < NH ? HT : LAY.KGZ, nf,
                                            [&](auto qc) {
                                                constexpr int q = decltype(qc)::value;
                                                if constexpr (q >= 1 && q <= HT) tile_store(sU + (q - 1) * TS, lane, ubs[q - 1]);
                                                if constexpr (q > HT && q <= 2 * HT) read_frag_A1(sc0 + (l * HT + q - HT - 1) * TS, lane, fa[q - HT - 1]);
                                                if constexpr (q > 2 * HT && q <= 3 * HT) read_frag_B1(sU + (q - 2 * HT - 1) * TS, lane, fb[q - 2 * HT - 1]);
                                                if constexpr (MULTI && l == NH - 1 && q >= 3 * HT) read_frag_A1(sE + (q - 3 * HT) * TS, lane, fd[q - 3 * HT]);
                                            });
                    cot_block<HT, HT>(fa, fb, Wh[l]);
                    G2_FENCE();
                });
                if constexpr (MULTI) {
                    // cbar_k = dbar_L^k .* act'_L out and its fragments back, behind them Wbar_1 += delta_1^k [gbar_k; 0]^T (covers the round
                    // trip), then Wbar_N += eps_k cbar_k^T
                    f32x4 cb[HT];
                    float fck[HT][4];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) { cb[mt] = db[mt] * dact(L - 1, mt, one_b); a2[L - 1][mt] += db[mt] * ck[mt]; }
                    G2_FENCE();
                    tiles_store<HT>(sU, lane, cb);
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) read_frag_B1(sU + mt * TS, lane, fck[mt]);
                    G2_FENCE();
                    cot_col(fd, fg[0], W1in);
                    cot_row(fek[0], fck, WNacc[0]);
                    G2_FENCE();
#pragma unroll
                    for (int s = 0; s < ZR; ++s) epsk[s] = eps_nx[s];
                }
                }   // probes
                G2_T(3);
                // ---- (4) top-down through the forward chain
                f32x4 hb[HT];
                float fc[HT][4];
                if constexpr (!MULTI) {   // cbar = dbar_L .* act'_L;  hbar_L = W_N^T kbar carries cbar out; Wbar_N += eps cbar^T is taken behind the next product
                    const float one_c = opaque_one();
                    f32x4 cb[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) { cb[mt] = db[mt] * dact(L - 1, mt, one_c); a2[L - 1][mt] = db[mt] * cvec[mt]; }
                    G2_FENCE();
                    zero_tiles<HT>(hb);
                    gemm_pf<HT, ZR, HT>(sm + LAY.bN, lane, RegIn<ZR>{kbar}, nf, hb, IMG_B(NH - 1), HT, nf,
                                        [&](auto qc) {
                                            constexpr int q = decltype(qc)::value;
                                            constexpr int per = (HT + ZR - 1) / ZR;   // cbar's HT tiles over the product's ZR k-steps
#pragma unroll
                                            for (int mt = q * per; mt < (q + 1) * per && mt < HT; ++mt) tile_store(sU + mt * TS, lane, cb[mt]);
                                        });
                } else {   // several probes: hbar_L = W_N^T kbar carries kbar out (Wbar_N += kbar h_L^T is taken behind the next product)
                    G2_FENCE();
                    zero_tiles<HT>(hb);
                    gemm_pf<HT, ZR, HT>(sm + LAY.bN, lane, RegIn<ZR>{kbar}, nf, hb, IMG_B(NH - 1), HT, nf,
                                        [&](auto qc) {
                                            constexpr int q = decltype(qc)::value;
                                            if constexpr (q == 0) tile_store(sU, lane, kt);
                                            if constexpr (q == 1) read_frag_A1(sU, lane, fk[0]);
                                        });
                }
                G2_T(4);
                float Zbar[ZR];
                const float one_t = opaque_one();
                static_for<0, L>([&](auto lc) {
                    constexpr int l = L - 1 - decltype(lc)::value;   // L-1 .. 0
                    f32x4 ab[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) {
                        // abar = hbar .* act' + a2 .* act'';  act'': tanh -> -2 h act', so abar = act' .* (hbar - 2 h .* a2);
                        // softplus -> s (1 - s) with s = act' = sigmoid(a)
                        const f32x4 d1 = dact(l, mt, one_t);
                        if constexpr (ACT == CNF_ACT_TANH) ab[mt] = d1 * __builtin_elementwise_fma(h[l][mt] * a2[l][mt], f32x4{-2.f, -2.f, -2.f, -2.f}, hb[mt]);
                        else ab[mt] = hb[mt] * d1 + a2[l][mt] * (d1 * (1.f - d1));
                    }
                    if constexpr (l > 0) {
                        // Wbar_{l+1} += abar_l h_{l-1}^T;  bbar_{l+1} += row sums of abar_l;  hbar_{l-1} = W_{l+1}^T abar_l
                        if constexpr (l == L - 1) G2_T(10);
                        float fa[HT][4], fb[HT][4];
                        auto operands = [&](auto qc) {
                            constexpr int q = decltype(qc)::value;
                            if constexpr (q < HT) tile_store(sc0 + q * TS, lane, ab[q]);
                            else if constexpr (q < 2 * HT) tile_store(sc0 + q * TS, lane, h[l - 1][q - HT]);
                            else if constexpr (q < 3 * HT) read_frag_A1(sc0 + (q - 2 * HT) * TS, lane, fa[q - 2 * HT]);
                            else read_frag_B1(sc0 + (q - 2 * HT) * TS, lane, fb[q - 3 * HT]);
                            // cbar's fragments first: with one hidden matrix its slot is the one h_{l-1} is about to take
                            if constexpr (!MULTI && l == L - 1 && q < HT) read_frag_B1(sU + q * TS, lane, fc[q]);
                            // several probes: h_L out to the early slots and its fragments back (delta_1^k's reads are long issued)
                            if constexpr (MULTI && l == L - 1 && q < HT) tile_store(sE + q * TS, lane, h[L - 1][q]);
                            if constexpr (MULTI && l == L - 1 && q >= 2 * HT && q < 3 * HT) read_frag_B1(sE + (q - 2 * HT) * TS, lane, fhL[q - 2 * HT]);
                        };
                        G2_FENCE();
                        zero_tiles<HT>(hb);
                        if constexpr (l > 1) gemm_pf<HT, 4 * HT, HT>(IMG_B(l - 1), lane, TileIn<HT>{ab}, nf, hb, IMG_B(l > 1 ? l - 2 : 0), HT, nf, operands);
                        else gemm_pf<HT, 4 * HT, 1>(IMG_B(l - 1), lane, TileIn<HT>{ab}, nf, hb, sm + LAY.b1, HT, nz, operands);
                        if constexpr (l == L - 1) G2_T(11);
                        cot_block<HT, HT>(fa, fb, Wh[l - 1]);
                        if constexpr (l == L - 1 && !MULTI) cot_row(fe[0], fc, WNacc[0]);   // Wbar_N += eps cbar^T
                        if constexpr (l == L - 1 && MULTI) {                               // Wbar_N += kbar h_L^T;  bbar_N += kbar
                            cot_row(fk[0], fhL, WNacc[0]);
#pragma unroll
                            for (int s = 0; s < ZR; ++s) bN[s] += kbar[s];
                        }
                        G2_FENCE();
                        if constexpr (l == L - 1) G2_T(12);
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) bh[l - 1][mt] += (fa[mt][0] + fa[mt][1]) + (fa[mt][2] + fa[mt][3]);
                    } else {
                        // Wbar_1 += abar_1 [z; t; 1]^T + delta_1 [gbar; 0]^T (+ abar_1 y^T);  Zbar = W_1[:,0:D]^T abar_1
                        // input pseudo tile [z (D rows); t; 0 ...; 1 at feature 15]: feature j <-> (register j>>2, lane group j&3)
                        f32x4 in_tile;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = 4 * r + g;
                            const float zv = r < ZR ? zs[r < ZR ? r : 0] : 0.f;
                            const float tv = (!autonomous && f == D) ? tt : (f == 15 ? 1.f : 0.f);
                            in_tile[r] = f < D ? zv : tv;
                        }
                        const f32x4 gt = dense_tile<ZR>(gbar);
                        // free slots by now: [HT, 2 HT) of the top-down pair; with one hidden tile ubar's slot (three hidden layers) or the
                        // early slot behind delta_1 (two)
                        constexpr int GT_SLOT = HT >= 2 ? HT + 1 : (NH == 2 ? 2 : 3);
                        float fa[HT][4], fd[HT][4], fi[1][4], fg[1][4];
                        G2_FENCE();
                        f32x4 zb[DT];
                        zero_tiles<DT>(zb);
                        // 4 HT k-steps of one MFMA each: abar_1 out (HT), in_tile and gbar out, then the fragments of abar_1, delta_1 (early
                        // slots), in_tile and gbar
                        gemm_pf<DT, 4 * HT, 1>(sm + LAY.b1, lane, TileIn<HT>{ab}, nz, zb, no_img, 0, nz,
                                               [&](auto qc) {
                                                   constexpr int q = decltype(qc)::value;
                                                   if constexpr (q < HT) tile_store(sc0 + q * TS, lane, ab[q]);
                                                   if constexpr (q == HT) { tile_store(sc0 + HT * TS, lane, in_tile); tile_store(sc0 + GT_SLOT * TS, lane, gt); }
                                                   if constexpr (!MULTI && q > HT && q <= 2 * HT) read_frag_A1(sE + (q - HT - 1) * TS, lane, fd[q - HT - 1]);
                                                   if constexpr (q > 2 * HT && q <= 3 * HT) read_frag_A1(sc0 + (q - 2 * HT - 1) * TS, lane, fa[q - 2 * HT - 1]);
                                                   if constexpr (q == (HT >= 2 ? 3 * HT + 1 : 3 * HT)) { read_frag_B1(sc0 + HT * TS, lane, fi[0]); read_frag_B1(sc0 + GT_SLOT * TS, lane, fg[0]); }
                                               });
                        if constexpr (!MULTI) cot_col(fd, fg[0], W1in);
                        cot_col(fa, fi[0], W1in);
                        if constexpr (CR > 0) cot_col(fa, fy[0], W1y);
                        G2_FENCE();
#pragma unroll
                        for (int s = 0; s < ZR; ++s) Zbar[s] = zb[s >> 2][s & 3];
                    }
                    G2_T(5 + (L - 1 - l));
                });
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int s = 0; s < ZR; ++s) Zb[j][s] = (j == st) ? Zbar[s] : Zb[j][s];
                sc_cur = sc_nxt;
#ifdef G2_TRACE
                G2_T(5 + L);
                if (blockIdx.x == 3 && step == 5 && st == 1 && lane == 0 && tile == (long long)blockIdx.x * 4 + wave) {
                    printf("w%d: fwd %d pull %d up %d WN %d", wave, (int)(tr[1] - tr[0]), (int)(tr[2] - tr[1]), (int)(tr[3] - tr[2]), (int)(tr[4] - tr[3]));
                    for (int k = 5; k <= 5 + L; ++k) printf(" %d", (int)(tr[k] - tr[k - 1]));
                    printf(" | total %d | top: ew %d prod %d cot %d\n", (int)(tr[5 + L] - tr[0]), (int)(tr[10] - tr[4]), (int)(tr[11] - tr[10]), (int)(tr[12] - tr[11]));
                }
#endif
            }
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                float acc = lam[s];
#pragma unroll
                for (int j = 0; j < 6; ++j) acc += Zb[j][s];
                lam[s] = acc;
                zn[s] = zn_nx[s];
#pragma unroll
                for (int j = 0; j < 6; ++j) kz[j][s] = kz_nx[j][s];
            }
        }
        if (a.grad_x && valid) {   // costate at t0 = dL/dz_0; its first nvars rows are dL/dx
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                const int f = 4 * s + g;
                if (f < a.nvars) a.grad_x[smp * a.nvars + f] = lam[s];
            }
        }
    }
    asm volatile("s_nop 15" ::: "memory");   // the last accumulates (asm: unknown to hipcc's hazard recogniser) before their results are read
    // every wave deposits its whole gradient in its own (zeroed) slab; grad_reduce_kernel sums the slabs in a fixed order
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < HT; ++nt)
                *reinterpret_cast<f32x4*>(slab + SL::WH + l * HT * HT * 256 + ((mt * HT + nt) * 64 + lane) * 4) = Wh[l][mt][nt];
            // bias of hidden layer l + 2: the fragment lanes (i, g) hold partial row sums of row rowmap(mt, i); its image is column 0
            // of a [mt] tile: lane (0, gg), register r <-> row 16 mt + 4 r + gg = rowmap(mt, 4 gg + r)
            const float bsum = group_sum(bh[l][mt]);
            if (g == 0) slab[SL::BH + l * HT * 256 + (mt * 64 + 16 * (n >> 2)) * 4 + (n & 3)] = bsum;
        }
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) {
        *reinterpret_cast<f32x4*>(slab + SL::W1 + ((mt * SL::NT1 + 0) * 64 + lane) * 4) = W1in[mt];
        if constexpr (CR > 0) *reinterpret_cast<f32x4*>(slab + SL::W1 + ((mt * SL::NT1 + 1) * 64 + lane) * 4) = W1y[mt];
    }
#pragma unroll
    for (int nt = 0; nt < HT; ++nt) *reinterpret_cast<f32x4*>(slab + SL::WN + (nt * 64 + lane) * 4) = WNacc[0][nt];
#pragma unroll
    for (int s = 0; s < ZR; ++s) {   // last-layer bias: dense layout (row 4 s + g, sample lane n): sum over the 16 sample lanes
        float v = bN[s];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (n == 0) slab[SL::BN + (16 * g) * 4 + s] = v;
    }
}

// ---------------------------------------------------------------------------------------
// host side: instance table (the launch itself is grad_launch's, cnf_grad.hip)
// ---------------------------------------------------------------------------------------
struct Grad2Inst {
    int HT, L, ZR, CR, ACT;
    GradKernel kern;
};
#ifndef G2_MULTI
#define G2_MULTI false
#endif
#define G2_INST(HT, L, ZR, CR, ACT) Grad2Inst { HT, L, ZR, CR, ACT, &mfma_grad2_kernel<HT, L, ZR, CR, ACT, G2_MULTI> }
// the shapes of cnf_grad.hip's table (kGrad): 1 .. 4 hidden tiles, 2 / 3 hidden layers, D <= 8 / 16, with and without conditions
#define G2_HT(HT, CR, ACT) G2_INST(HT, 3, 2, CR, ACT), G2_INST(HT, 2, 2, CR, ACT), G2_INST(HT, 3, 4, CR, ACT), G2_INST(HT, 2, 4, CR, ACT)
#define G2_SHAPES(CR, ACT) G2_HT(1, CR, ACT), G2_HT(2, CR, ACT), G2_HT(3, CR, ACT), G2_HT(4, CR, ACT)
#ifdef G2_ONLY
static const Grad2Inst kGrad2[] = {G2_ONLY};
#else
static const Grad2Inst kGrad2[] = {G2_SHAPES(0, CNF_ACT_TANH), G2_SHAPES(0, CNF_ACT_SOFTPLUS), G2_SHAPES(4, CNF_ACT_TANH), G2_SHAPES(4, CNF_ACT_SOFTPLUS)};
#endif

GradKernel G2_FIND(int HT, int L, int ZR, int CR, int ACT) {
    for (const Grad2Inst& g : kGrad2)
        if (g.HT == HT && g.L == L && g.ZR == ZR && g.CR == CR && g.ACT == ACT) return g.kern;
    return nullptr;
}

}  // namespace cnf
