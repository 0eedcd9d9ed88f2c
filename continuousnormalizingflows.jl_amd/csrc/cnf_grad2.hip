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
    asm("s_nop 1\n\t" G2_MF(0, 1, 5) G2_MF(0, 2, 6) G2_MF(0, 3, 7) G2_MF(0, 4, 8)
        : "+a"(W[0])
        : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[0][3]));
}
__device__ __forceinline__ void cot_row(const float (&fa)[4], const float (&fb)[2][4], f32x4 (&W)[2]) {
    asm("s_nop 1\n\t" G2_MF(0, 2, 6) G2_MF(1, 2, 10) G2_MF(0, 3, 7) G2_MF(1, 3, 11) G2_MF(0, 4, 8) G2_MF(1, 4, 12) G2_MF(0, 5, 9) G2_MF(1, 5, 13)
        : "+a"(W[0]), "+a"(W[1])
        : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[0][3]), "v"(fb[1][0]), "v"(fb[1][1]), "v"(fb[1][2]), "v"(fb[1][3]));
}
__device__ __forceinline__ void cot_row(const float (&fa)[4], const float (&fb)[3][4], f32x4 (&W)[3]) {
    asm("s_nop 1\n\t" G2_MF(0, 3, 7) G2_MF(1, 3, 11) G2_MF(2, 3, 15) G2_MF(0, 4, 8) G2_MF(1, 4, 12) G2_MF(2, 4, 16) G2_MF(0, 5, 9) G2_MF(1, 5, 13) G2_MF(2, 5, 17) G2_MF(0, 6, 10) G2_MF(1, 6, 14) G2_MF(2, 6, 18)
        : "+a"(W[0]), "+a"(W[1]), "+a"(W[2])
        : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[0][3]), "v"(fb[1][0]), "v"(fb[1][1]), "v"(fb[1][2]), "v"(fb[1][3]), "v"(fb[2][0]), "v"(fb[2][1]), "v"(fb[2][2]), "v"(fb[2][3]));
}
__device__ __forceinline__ void cot_row(const float (&fa)[4], const float (&fb)[4][4], f32x4 (&W)[4]) {
    asm("s_nop 1\n\t" G2_MF(0, 4, 8) G2_MF(1, 4, 12) G2_MF(2, 4, 16) G2_MF(3, 4, 20) G2_MF(0, 5, 9) G2_MF(1, 5, 13) G2_MF(2, 5, 17) G2_MF(3, 5, 21) G2_MF(0, 6, 10) G2_MF(1, 6, 14) G2_MF(2, 6, 18) G2_MF(3, 6, 22) G2_MF(0, 7, 11) G2_MF(1, 7, 15) G2_MF(2, 7, 19) G2_MF(3, 7, 23)
        : "+a"(W[0]), "+a"(W[1]), "+a"(W[2]), "+a"(W[3])
        : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[0][3]), "v"(fb[1][0]), "v"(fb[1][1]), "v"(fb[1][2]), "v"(fb[1][3]), "v"(fb[2][0]), "v"(fb[2][1]), "v"(fb[2][2]), "v"(fb[2][3]), "v"(fb[3][0]), "v"(fb[3][1]), "v"(fb[3][2]), "v"(fb[3][3]));
}
// column form: W[mt] += sum_s fa[mt][s] fb[s] - MT A fragments, one B fragment (column tile)
__device__ __forceinline__ void cot_col(const float (&fa)[1][4], const float (&fb)[4], f32x4 (&W)[1]) {
    asm("s_nop 1\n\t" G2_MF(0, 5, 1) G2_MF(0, 6, 2) G2_MF(0, 7, 3) G2_MF(0, 8, 4)
        : "+a"(W[0])
        : "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[0][3]));
}
__device__ __forceinline__ void cot_col(const float (&fa)[2][4], const float (&fb)[4], f32x4 (&W)[2]) {
    asm("s_nop 1\n\t" G2_MF(0, 6, 2) G2_MF(1, 10, 2) G2_MF(0, 7, 3) G2_MF(1, 11, 3) G2_MF(0, 8, 4) G2_MF(1, 12, 4) G2_MF(0, 9, 5) G2_MF(1, 13, 5)
        : "+a"(W[0]), "+a"(W[1])
        : "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[0][3]), "v"(fa[1][0]), "v"(fa[1][1]), "v"(fa[1][2]), "v"(fa[1][3]));
}
__device__ __forceinline__ void cot_col(const float (&fa)[3][4], const float (&fb)[4], f32x4 (&W)[3]) {
    asm("s_nop 1\n\t" G2_MF(0, 7, 3) G2_MF(1, 11, 3) G2_MF(2, 15, 3) G2_MF(0, 8, 4) G2_MF(1, 12, 4) G2_MF(2, 16, 4) G2_MF(0, 9, 5) G2_MF(1, 13, 5) G2_MF(2, 17, 5) G2_MF(0, 10, 6) G2_MF(1, 14, 6) G2_MF(2, 18, 6)
        : "+a"(W[0]), "+a"(W[1]), "+a"(W[2])
        : "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[0][3]), "v"(fa[1][0]), "v"(fa[1][1]), "v"(fa[1][2]), "v"(fa[1][3]), "v"(fa[2][0]), "v"(fa[2][1]), "v"(fa[2][2]), "v"(fa[2][3]));
}
__device__ __forceinline__ void cot_col(const float (&fa)[4][4], const float (&fb)[4], f32x4 (&W)[4]) {
    asm("s_nop 1\n\t" G2_MF(0, 8, 4) G2_MF(1, 12, 4) G2_MF(2, 16, 4) G2_MF(3, 20, 4) G2_MF(0, 9, 5) G2_MF(1, 13, 5) G2_MF(2, 17, 5) G2_MF(3, 21, 5) G2_MF(0, 10, 6) G2_MF(1, 14, 6) G2_MF(2, 18, 6) G2_MF(3, 22, 6) G2_MF(0, 11, 7) G2_MF(1, 15, 7) G2_MF(2, 19, 7) G2_MF(3, 23, 7)
        : "+a"(W[0]), "+a"(W[1]), "+a"(W[2]), "+a"(W[3])
        : "v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]), "v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[0][3]), "v"(fa[1][0]), "v"(fa[1][1]), "v"(fa[1][2]), "v"(fa[1][3]), "v"(fa[2][0]), "v"(fa[2][1]), "v"(fa[2][2]), "v"(fa[2][3]), "v"(fa[3][0]), "v"(fa[3][1]), "v"(fa[3][2]), "v"(fa[3][3]));
}
template <int MT, int NT>
__device__ __forceinline__ void cot_block(const float (&fa)[MT][4], const float (&fb)[NT][4], f32x4 (&W)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cot_row(fa[mt], fb, W[mt]);
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

}  // namespace

template <int HT, int L, int ZR, int CR, int ACT>
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
    __syncthreads();   // the only barrier of the kernel
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* slab = a.slab + ((long long)blockIdx.x * 4 + wave) * SL::TOTAL;
    float* scr = smem + G::XCH + wave * G::XCH_W;   // wave-private transpose scratch: XCH_TILES padded tiles
    const long long ntiles = (a.B + 15) / 16;
    const int D = a.D;
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
            eps[s] = f < D ? a.eps[sc * D + f] : 0.f;
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
        f32x4 cvec[HT], qvec[HT];   // c = W_N^T eps, q = W_1[:,0:D] eps: constant over the solve
        zero_tiles<HT>(cvec);
        zero_tiles<HT>(qvec);
        gemm_tiles<HT, ZR>(smem + LAY.bN, lane, RegIn<ZR>{eps}, cvec);
        gemm_tiles<HT, ZR>(smem + LAY.f1z, lane, RegIn<ZR>{eps}, qvec);

#pragma clang loop unroll(disable)
        for (int step = a.nsteps - 1; step >= 0; --step) {
            float tn = a.t0 + (float)step * dt0, dt = dt0;
            if (a.tgrid) { tn = a.tgrid[step]; dt = a.tgrid[step + 1] - tn; }
            float zn[ZR];
#pragma unroll
            for (int s = 0; s < ZR; ++s) zn[s] = a.ckpt[(((long long)step * ntiles + tile) * 64 + lane) * a.ckpt_zr + s];
            // stage derivatives kz_i (z rows), checkpointed by the forward kernel
            float kz[6][ZR];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int s = 0; s < ZR; ++s) kz[j][s] = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < ns) {
#pragma unroll
                    for (int s = 0; s < ZR; ++s)
                        kz[j][s] = a.ckpt_k[((((long long)step * ns + j) * ntiles + tile) * 64 + lane) * a.ckpt_zr + s];
                }
            float Zb[6][ZR];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int s = 0; s < ZR; ++s) Zb[j][s] = 0.f;
#pragma clang loop unroll(disable)
            for (int st = ns - 1; st >= 0; --st) {
#ifdef G2_TRACE
                unsigned long long tr[16];
#endif
                G2_T(0);
                float zs[ZR], kbar[ZR];
                const float bi = a.T.b[st];
#pragma unroll
                for (int s = 0; s < ZR; ++s) {
                    float acc = 0.f, kb = bi * lam[s];
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc = fmaf(a.T.a[st][j], kz[j][s], acc);
#pragma unroll
                    for (int j = 1; j < 6; ++j) kb = fmaf(a.T.a[j][st], Zb[j][s], kb);   // a[j][st] != 0 only for j > st
                    zs[s] = fmaf(dt, acc, zn[s]);
                    kbar[s] = dt * kb;
                }
                const float cl = valid ? dt * bi : 0.f;   // cotangent of ldot: dL/d(dlogp) = +1
                const float cE = cl * a.lam1, cn = cl * a.lam2;   // cotangents of Edot, ndot
                const bool regz = a.lam1 != 0.f, regj = a.lam2 != 0.f;   // wave-uniform
                const float tt = tn + a.T.c[st] * dt;
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                const float* sm = smem + opaque;
                float* sc0 = scr + opaque;

                // (1) recompute the forward chain
                f32x4 h[L][HT], d[L][HT];
                grad_forward<HT, L, ZR, CR, ACT>(sm, lane, tt, autonomous, zs, y, h, d);
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
                // (2) first-order pullback; delta_{l+1} (l >= 1) goes to scratch slot group l - 1 as it appears: the A operand of
                //     Wbar_{l+1}'s second term
                f32x4 u[NH > 0 ? NH : 1][HT], dl[HT];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) dl[mt] = cvec[mt] * d[L - 1][mt];
#pragma unroll
                for (int l = L - 1; l >= 1; --l) {
                    tiles_store<HT>(sc0 + (l - 1) * HT * TS, lane, dl);
                    zero_tiles<HT>(u[l - 1]);
                    gemm_tiles<HT, 4 * HT>(sm + LAY.bh + (l - 1) * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{dl}, u[l - 1]);
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) dl[mt] = u[l - 1][mt] * d[l - 1][mt];
                }
                G2_T(2);
                // gbar = cotangent of g = eps^T J (dense layout): -c_l eps (+ c_n g/|g|);  dbar_1 = W_1[:,0:D] gbar
                float gbar[ZR];
#pragma unroll
                for (int s = 0; s < ZR; ++s) gbar[s] = -cl * eps[s];
                f32x4 db[HT], a2[L][HT];   // a2_l = dbar_l .* u_l  (multiplies act''_l later)
                if (regj) {
                    f32x4 gacc[DT];
                    zero_tiles<DT>(gacc);
                    gemm_tiles<DT, 4 * HT>(sm + LAY.b1, lane, TileIn<HT>{dl}, gacc);   // g = W_1[:,0:D]^T delta_1
                    float n2 = 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) n2 = fmaf(gacc[s >> 2][s & 3], gacc[s >> 2][s & 3], n2);
                    n2 = group_sum(n2);
                    const float inv = n2 > 0.f ? cn * rsqrtf(n2) : 0.f;
#pragma unroll
                    for (int s = 0; s < ZR; ++s) gbar[s] = fmaf(inv, gacc[s >> 2][s & 3], gbar[s]);
                    zero_tiles<HT>(db);
                    gemm_tiles<HT, ZR>(sm + LAY.f1z, lane, RegIn<ZR>{gbar}, db);
                } else {
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) db[mt] = qvec[mt] * (-cl);                // = W_1[:,0:D] (-c_l eps)
                }
                // (3) bottom-up through the pullback; Wbar_{l+2} += delta_{l+1} ubar_l^T beside the product that consumes ubar_l
                f32x4 dl0[HT];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) dl0[mt] = dl[mt];
#pragma unroll
                for (int l = 0; l < NH; ++l) {
                    f32x4 ubs[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) { ubs[mt] = db[mt] * d[l][mt]; a2[l][mt] = db[mt] * u[l][mt]; }
                    float* sb = sc0 + NH * HT * TS;
                    tiles_store<HT>(sb, lane, ubs);
                    float fa[HT][4], fb[HT][4];
                    frags_A<HT>(sc0 + l * HT * TS, lane, fa);
                    frags_B<HT>(sb, lane, fb);
                    zero_tiles<HT>(db);
                    gemm_tiles<HT, 4 * HT>(sm + LAY.fh + l * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{ubs}, db);   // W_{l+2} ubar_l
                    cot_block<HT, HT>(fa, fb, Wh[l]);
                }
                G2_T(3);
                f32x4 cb[HT];   // cbar = dbar_L .* act'_L: Wbar_N[i, f] += eps_i cbar_f
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) { cb[mt] = db[mt] * d[L - 1][mt]; a2[L - 1][mt] = db[mt] * cvec[mt]; }
                // (4) top-down through the forward chain
                {   // Wbar_N += eps cbar^T + kbar h_L^T;  bbar_N += kbar
                    f32x4 et[1], kt[1];
                    et[0] = dense_tile<ZR>(eps);
                    kt[0] = dense_tile<ZR>(kbar);
                    tile_store(sc0 + 0 * TS, lane, et[0]);
                    tile_store(sc0 + 1 * TS, lane, kt[0]);
                    tiles_store<HT>(sc0 + 2 * TS, lane, cb);
                    tiles_store<HT>(sc0 + (2 + HT) * TS, lane, h[L - 1]);
                    float fe[1][4], fk[1][4], fc[HT][4], fh[HT][4];
                    frags_A<1>(sc0 + 0 * TS, lane, fe);
                    frags_A<1>(sc0 + 1 * TS, lane, fk);
                    frags_B<HT>(sc0 + 2 * TS, lane, fc);
                    frags_B<HT>(sc0 + (2 + HT) * TS, lane, fh);
                    cot_row(fe[0], fc, WNacc[0]);
                    cot_row(fk[0], fh, WNacc[0]);
#pragma unroll
                    for (int s = 0; s < ZR; ++s) bN[s] += kbar[s];
                }
                f32x4 hb[HT];
                zero_tiles<HT>(hb);
                gemm_tiles<HT, ZR>(sm + LAY.bN, lane, RegIn<ZR>{kbar}, hb);   // W_N^T kbar
                G2_T(4);
                float Zbar[ZR];
#pragma unroll
                for (int l = L - 1; l >= 0; --l) {
                    f32x4 ab[HT];
#pragma unroll
                    for (int mt = 0; mt < HT; ++mt) {
                        // act'': tanh -> -2 h (1 - h^2);  softplus -> s (1 - s) with s = act' = sigmoid(a)
                        const f32x4 d2 = ACT == CNF_ACT_TANH ? h[l][mt] * d[l][mt] * -2.f : d[l][mt] * (1.f - d[l][mt]);
                        ab[mt] = hb[mt] * d[l][mt] + a2[l][mt] * d2;
                    }
                    if (l > 0) {
                        // Wbar_{l+1} += abar_l h_{l-1}^T;  bbar_{l+1} += row sums of abar_l;  hbar_{l-1} = W_{l+1}^T abar_l
                        tiles_store<HT>(sc0, lane, ab);
                        tiles_store<HT>(sc0 + HT * TS, lane, h[l - 1]);
                        float fa[HT][4], fb[HT][4];
                        frags_A<HT>(sc0, lane, fa);
                        frags_B<HT>(sc0 + HT * TS, lane, fb);
                        zero_tiles<HT>(hb);
                        gemm_tiles<HT, 4 * HT>(sm + LAY.bh + (l - 1) * MfmaLayout::imgA(HT, HT), lane, TileIn<HT>{ab}, hb);
#pragma unroll
                        for (int mt = 0; mt < HT; ++mt) bh[l - 1][mt] += (fa[mt][0] + fa[mt][1]) + (fa[mt][2] + fa[mt][3]);
                        cot_block<HT, HT>(fa, fb, Wh[l - 1]);
                    } else {
                        // input pseudo tile [z (D rows); t; ...; 1 at feature 15]: feature j <-> (register j>>2, lane group j&3)
                        f32x4 in_tile;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = r < ZR ? zs[r < ZR ? r : 0] : 0.f;
                            if (!autonomous && 4 * r + g == D) v = tt;
                            if (4 * r + g > D || (autonomous && 4 * r + g == D)) v = 0.f;
                            if (4 * r + g == 15) v = 1.f;
                            in_tile[r] = v;
                        }
                        // Wbar_1 += abar_1 [z; t; 1]^T + delta_1 [gbar; 0]^T (+ abar_1 y^T)
                        f32x4 gt[1];
                        gt[0] = dense_tile<ZR>(gbar);
                        tiles_store<HT>(sc0, lane, ab);
                        tiles_store<HT>(sc0 + HT * TS, lane, dl0);
                        tile_store(sc0 + (2 * HT + 0) * TS, lane, in_tile);
                        tile_store(sc0 + (2 * HT + 1) * TS, lane, gt[0]);
                        float fa[HT][4], fd[HT][4], fi[1][4], fg[1][4];
                        frags_A<HT>(sc0, lane, fa);
                        frags_A<HT>(sc0 + HT * TS, lane, fd);
                        frags_B<1>(sc0 + (2 * HT + 0) * TS, lane, fi);
                        frags_B<1>(sc0 + (2 * HT + 1) * TS, lane, fg);
                        f32x4 zb[DT];
                        zero_tiles<DT>(zb);
                        gemm_tiles<DT, 4 * HT>(sm + LAY.b1, lane, TileIn<HT>{ab}, zb);   // W_1[:,0:D]^T abar_1
                        cot_col(fa, fi[0], W1in);
                        cot_col(fd, fg[0], W1in);
                        if constexpr (CR > 0) {
                            f32x4 yt[1];
                            yt[0] = dense_tile<(CR > 0 ? CR : 1)>(y);
                            tile_store(sc0 + (2 * HT + 2) * TS, lane, yt[0]);
                            float fy[1][4];
                            frags_B<1>(sc0 + (2 * HT + 2) * TS, lane, fy);
                            cot_col(fa, fy[0], W1y);
                        }
#pragma unroll
                        for (int s = 0; s < ZR; ++s) Zbar[s] = zb[s >> 2][s & 3];
                    }
                    G2_T(5 + (L - 1 - l));
                }
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int s = 0; s < ZR; ++s) Zb[j][s] = (j == st) ? Zbar[s] : Zb[j][s];
#ifdef G2_TRACE
                G2_T(5 + L);
                if (blockIdx.x == 3 && step == 5 && st == 1 && lane == 0 && tile == (long long)blockIdx.x * 4 + wave) {
                    printf("w%d: fwd %d pull %d up %d WN %d", wave, (int)(tr[1] - tr[0]), (int)(tr[2] - tr[1]), (int)(tr[3] - tr[2]), (int)(tr[4] - tr[3]));
                    for (int k = 5; k <= 5 + L; ++k) printf(" %d", (int)(tr[k] - tr[k - 1]));
                    printf(" | total %d\n", (int)(tr[5 + L] - tr[0]));
                }
#endif
            }
#pragma unroll
            for (int s = 0; s < ZR; ++s) {
                float acc = lam[s];
#pragma unroll
                for (int j = 0; j < 6; ++j) acc += Zb[j][s];
                lam[s] = acc;
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
#define G2_INST(HT, L, ZR, CR, ACT) Grad2Inst { HT, L, ZR, CR, ACT, &mfma_grad2_kernel<HT, L, ZR, CR, ACT> }
static const Grad2Inst kGrad2[] = {
    G2_INST(4, 3, 2, 0, CNF_ACT_TANH),
};

GradKernel grad2_kernel(int HT, int L, int ZR, int CR, int ACT) {
    for (const Grad2Inst& g : kGrad2)
        if (g.HT == HT && g.L == L && g.ZR == ZR && g.CR == CR && g.ACT == ACT) return g.kern;
    return nullptr;
}

}  // namespace cnf
