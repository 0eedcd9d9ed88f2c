// cnf_tiles.h - the TILE-NATIVE operand store of the cooperative gradient's second form (round 6) and the
// interfaces of the kernels that read and write it: the checkpointing forward solves (cnf_coop.hip, cnf_coop_d.hip), the
// second-order reverse sweep (cnf_coop_grad3.hip) and the weight-cotangent products over tiles (cnf_wgrad_tiles.hip).
//
// Reference: the parameter gradient of `loss` through the solve (QuadratureAdjoint + ZygoteVJP, src/core/icnf.jl:90-99, driven by
// src/exts/mlj_ext/core_icnf.jl:42-51); the discrete form differentiated here is DESIGN.md section 8.
//
// A TILE is what one wave holds of a [16 features x 16 samples] block after an MFMA product: 64 lanes x 4 floats = 1 KB, lane
// (g = lane >> 4, n = lane & 15), component e  <->  feature 16 mt + 4 e + g of sample n (cnf_mfma_layout.h's row permutation: an
// accumulator tile IS the next product's B operand).  Stored lane-linear, a tile is one coalesced 16-byte-per-lane access for
// whichever wave owns it - so the forward solve, the sweep and the cotangent products may each deal the tiles over their waves
// as they like - and the cotangent product gets its operands (sample index on the MFMA K axis) from a tile with four
// 4-byte reads per k-step through a padded LDS copy (cnf_wgrad_tiles.hip).
//
// Arrays are [column tile][tiles per column tile][64 lanes][4]; a COLUMN TILE is 16 samples of one Runge-Kutta stage:
// ct = stage * ntp + (16-sample tile of the batch), ntp = the checkpoint arrays' padded tile count.
#pragma once
#include <hip/hip_runtime.h>

#include "cnf_coop_grad.h"

namespace cnf {

// ---- the stage store of the forward solve: h_l and delta_l of every hidden layer, every stage of every step ----
// tile (kind k: 0 = h, 1 = delta; layer l; step; stage; sample tile t; feature tile m) sits at float offset
//   ((((k L + l) nsteps + step) ns + stage) ntp + t) HT + m) * 256
struct StageStore {
    int L, nsteps, ns, HT;
    long long ntp;
    __host__ __device__ long long layer_stride() const { return (long long)nsteps * ns * ntp * HT * 256; }   // floats per (kind, layer)
    __host__ __device__ long long total() const { return 2LL * L * layer_stride(); }
    __host__ __device__ long long at(int k, int l, int step, int stage) const {   // float offset of [ntp][HT] tiles
        return (long long)(k * L + l) * layer_stride() + ((long long)step * ns + stage) * ntp * HT * 256;
    }
};

// ---- weight-cotangent products over tiles:  C[M x Nc] += sum over column tiles of  X0 Y0^T + X1 Y1^T  ----
struct WTTerm {
    const float* x;   // row operand: tile (ct, m) at x + (ct * xtm + m) * 256
    const float* y;   // column operand: tile (ct, t) at y + (ct * ytm + t) * 256
    int xtm, ytm;     // tiles per column tile of the two arrays
};
struct WTArgs {
    float* slabs;             // chunk c accumulates into slabs + c * slab_stride: C column-major, ld = M, (Nc + bias) columns
    long long slab_stride;
    WTTerm t[2];
    long long nct, chunk;     // column tiles in total / per chunk
    int M, Nc;
    int nchunks, rblocks, groups;
    int bias;                 // 1: the row sums of term 1's X (the bias cotangent) go to column Nc of C
};
int wgrad_tiles_chunks(int M, int Nc, long long nct, int num_cus, long long* chunk_out);
hipError_t wgrad_tiles(float* slabs, long long slab_stride, long long chunk, int nchunks, int M, int Nc, const WTTerm& t0, const WTTerm& t1,
                       long long nct, int bias, hipStream_t st);

// ---- the second-order reverse sweep (cnf_coop_grad3.hip): one launch per RK step ----
struct CG3Args {
    CGArgs c;                 // packed, eps, ckpt / ckpt_k / ckpt_g, lam, zb, grad_x, B, ntiles_pad, step, nsteps, tn, dt, D, nvars, H, lam1..3, T
    const float* fh[3];       // h_l of this step (stage store): [ns][ntp][HTs] tiles
    const float* fd[3];       // delta_l
    float* sv[3];             // out: vbar_l  (cbar at l = L - 1)
    float* ss[3];             // out: sbar_l
    float* gb;                // out: [gbar] tiles, DT per column tile
    float* zt;                // out: [z; t] tiles, DTZ per column tile
    float* ep;                // out: [eps]
    float* kb;                // out: [kbar]
    int HTs, DTZ, DTs;        // tiles per column tile of the H-row arrays, of gb / zt, of ep / kb
    int ck_tiles;             // 1: the checkpoint rows (c.ckpt, c.ckpt_k, c.ckpt_g) are in tile layout (KArgs::ck_tiles)
};
bool coop_grad3_supported(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int CR_lay);
hipError_t coop_grad3_step_launch(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, const CG3Args& a, int num_cus, hipStream_t st);
// the same step with two workgroups per CU (cnf_coop_grad3w.hip: two hidden layers, at most 256 registers per wave and 80 KB of LDS)
bool coop_grad3w_supported(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int CR_lay);
hipError_t coop_grad3w_step_launch(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, const CG3Args& a, int num_cus, hipStream_t st);

}  // namespace cnf
