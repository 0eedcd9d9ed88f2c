// cnf_coop_grad.h — interface of the cooperative reverse sweep (cnf_coop_grad.hip) and of the checkpointing form of the
// cooperative forward solve (cnf_coop.hip) it pairs with.
#pragma once
#include "cnf_mfma_dev.h"

namespace cnf {

// One RK step of the reverse sweep over all column super-tiles (one launch).  Operand arrays of the deferred weight-cotangent
// products are column-major, 2 * ns * B columns each: columns [i B, (i+1) B) of the first half belong to stage i's first term,
// the same columns of the second half (offset ns * B) to its second term.
struct CGArgs {
    const float* packed;      // the cooperative plan's operand image (forward images carry the tanh pre-scale)
    const float* eps;         // D x B
    const float* ys;          // C x B conditions or null
    int C;
    const float* ckpt;        // z at the start of every step and after the last: [step][tile][lane][ZR] (forward kernel, CK form)
    const float* ckpt_k;      // stage derivatives: [step * ns + stage][tile][lane][ZR]
    const float* ckpt_g;      // g = eps^T J of every stage, same layout (read when lam2 != 0)
    float* lam;               // costate, [tile][lane][ZR]: read (unless this is the last step), written
    float* zb;                // Zbar_j of the running step's stages, [tile][lane][6][ZR] (scratch of the kernel)
    float* grad_x;            // nvars x B or null; written by step 0
    float* scratch;           // per-workgroup scratch, `scratch_stride` floats apart
    long long scratch_stride;
    float* xh[3];             // X_l, l = 1 .. L: [delta_l | sbar_l], H rows, ld = H
    float* yh[3];             // Y_l, l = 1 .. L: [vbar_l; 0 | h_l; 1] (l < L), [cbar; 0 | h_L; 1] (l = L), H + 1 rows, ld = H + 1
    float* y1;                // [gbar; 0; 0 | z; t; 1]: the input side of Wbar_1, ld_y1 = n_in + 1 rows
    float* xN;                // [eps | kbar]: D rows, ld = D
    int ld_y1;
    int ldy;                  // leading dimension of every Y_l (>= H + 1; a multiple of 16 keeps the 16-byte stores line-aligned)
    long long B;
    long long ntiles_pad;     // 16-sample tiles of the checkpoint arrays (the forward kernel's: 4 x ceil(B / 64))
    int step, nsteps;
    float tn, dt;             // this step's start time and length
    int D, nvars, H, autonomous;
    float lam1, lam2, lam3;   // weights of |zdot|, |eps^T J|, |z_aug| in the objective (src/core/icnf.jl:628-637)
    Tableau T;
};

// LDS of one workgroup: two exchange buffers [HT][2 NT column tiles][64 lanes] + [z | gbar], [eps | kbar] and gbar images
constexpr int coop_grad_lds_bytes(int HT, int ZR, int NT, int CR = 0) {   // (+ the condition image)
    return (2 * HT * 2 * NT * 64 + (2 * ((ZR + 3) / 4) * 2 * NT + ((ZR + 3) / 4) * NT + ((CR + 3) / 4) * NT) * 64) * 16;
}
struct CoopGradInst {
    int HT, L, ZR, CR, ACT, NT;
    hipError_t (*fn[2])(const CGArgs&, int, hipStream_t);   // [0] RK4 (4 stages), [1] Tsit5 (6 stages)
};
const CoopGradInst* coop_grad_table_tanh(int* n);       // cnf_coop_grad.hip
const CoopGradInst* coop_grad_table_softplus(int* n);   // cnf_coop_grad_softplus.hip
bool coop_grad_supported(int HT, int L, int ZR, int CR, int ACT);
int coop_grad_scratch_slots(int L);   // one chain's tile set ([HT] tiles) each, per workgroup
int coop_grad_nblocks(long long B, int num_cus, int HT, int ZR, int CR = 0, int NT = 1);
int coop_grad_nt(int HT, int L, int ZR, int CR, int ACT);   // sample tiles per super-tile of the instance that serves the shape   // workgroups of a launch (each owns `scratch_stride` floats of scratch)
hipError_t coop_grad_step_launch(int HT, int L, int ZR, int CR, int ACT, const CGArgs& a, int num_cus, hipStream_t st);
// the dealt form of the sweep (cnf_coop_dgrad.hip): same arguments, for the flows whose forward solve runs on cnf_coop_d.hip's kernels
// (H hidden units, D state rows, L hidden layers; (HT_lay, ZR_lay, CR_lay) = the plan's layout)
bool coopd_grad_supported(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, int CR_lay);
hipError_t coopd_grad_step_launch(int H, int D, int L, int ACT, int HT_lay, int ZR_lay, const CGArgs& a, int num_cus, hipStream_t st);
// the cooperative forward solve with step / stage checkpoints in tile layout (cnf_coop.hip)
bool coop_ckpt_supported(int HT, int L, int ZR, int ACT);
hipError_t coop_launch_ckpt(int HT, int L, int ZR, int ACT, const KArgs& a, int num_cus, hipStream_t st);

}  // namespace cnf
