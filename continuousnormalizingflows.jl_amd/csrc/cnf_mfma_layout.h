// cnf_mfma_layout.h — LDS image of the Dense chain for the fused MFMA solve kernels.
//
// Every product of the MLP and of its pullback is computed as  Out[M x 16 samples] =
// A[M x K] * In[K x 16 samples]  with v_mfma_f32_16x16x4_f32: samples on the N axis (one
// 16-sample tile per wave), features on M, and the *accumulator tile of one layer fed straight
// back as the B operand of the next* — no LDS round trip, no cross-lane movement.
//
// That works because of one row permutation.  The C/D layout of the 16x16 MFMA puts row i of
// the M-tile in lane group g = i>>2 (lanes 16g..16g+15), register r = i&3; the B operand of
// k-step s wants k = 4s + g from lane group g.  If M-tile row i of tile mt is made to compute
// feature  f = 16 mt + 4 (i&3) + (i>>2)  then register r of accumulator tile mt holds, in lane
// group g, feature 4 (4 mt + r) + g — exactly the B operand of k-step s = 4 mt + r in natural k
// order.  So the A images below are stored with permuted rows and natural columns.
//
// "dense layout" of a D-row quantity (z, eps, zdot, eps^T J): register s (s < ZR = ceil(D/4)),
// lane group g holds row 4 s + g of the lane's sample — D = 8 uses 2 registers on all 64 lanes.
//
// An "A image" for a product with MT M-tiles and KG k-groups (4 k-steps each) is
// [mt][kg][lane 0..63][j 0..3] floats: lane l, component j = A[rowmap(mt, l&15)][16 kg + 4 j + (l>>4)]
// read with one conflict-free ds_read_b128 per 4 MFMAs.
// A "C vector" (bias, time column) for MT tiles is [mt][g][r]: value of feature 16 mt + 4 r + g.
#pragma once
#include "cnf.h"

namespace cnf {

struct MfmaLayout {
    int HT, L, ZR, CR;   // hidden tiles (H_pad = 16 HT), hidden layers, k-steps of D and of C
    int arith;           // 0: f32 hidden images; 1: split-bf16 hidden images (cnf_mfma_kernel.h)
    int DT, KGZ, KGC;    // M-tiles of D, k-groups of D and C
    // A images (float offsets)
    int f1z, f1y;        // layer 1, z columns / cond columns          M = H, K = D_pad / C_pad
    int fh;              // hidden layers 2..L, consecutive             M = H, K = H
    int fN;              // last layer                                  M = D, K = H
    int bN;              // W_N^T                                       M = H, K = D_pad
    int bh;              // W_l^T for l = 2..L (stored in forward order) M = H, K = H
    int b1;              // W_1[:, 0:D]^T                               M = D, K = H
    // C vectors
    int v_b1, v_w1t, v_bh, v_bN;
    int qtr;             // tangent-engine layouts with L = 2: image of Q = W_2 .* (W_1[:,0:D] W_3)^T (exact trace = act'_2^T Q act'_1)
    int v_w1c;           // tangent-engine layouts only: column i of W_1[:, 0:D] as a C vector (exact trace: tau_1 = W_1 e_i)
    int v_wNr;           // tangent-engine layouts only: row i of W_N as a C vector, i < 4 ZR (exact trace: J_ii = <W_N[i,:], tau>)
    int total;           // floats of the packed image (global memory)
    int lds_total;       // floats staged into LDS (= total unless fN_global)
    bool fN_global;      // the last-layer image sits behind the LDS part and is read from global memory

    static constexpr int imgA(int MT, int KG) { return MT * KG * 256; }
    // hidden H x H image in split-bf16 form: [split 3][mt HT][chunk HT/2][lane 64][8 bf16] = 16 B per
    // lane per (split, mt, chunk); in float units
    static constexpr int imgH16(int HT) { return 3 * HT * (HT / 2) * 256; }
    constexpr int imgHid() const { return arith ? imgH16(HT) : imgA(HT, HT); }
    static constexpr int vecC(int MT) { return MT * 16; }

    constexpr MfmaLayout(int HT_, int L_, int ZR_, int CR_, bool with_bwd, int arith_ = 0)
        : HT(HT_), L(L_), ZR(ZR_), CR(CR_), arith(arith_), DT((ZR_ + 3) / 4), KGZ((ZR_ + 3) / 4), KGC((CR_ + 3) / 4),
          f1z(0), f1y(0), fh(0), fN(0), bN(0), bh(0), b1(0), v_b1(0), v_w1t(0), v_bh(0), v_bN(0), qtr(-1), v_w1c(-1), v_wNr(-1),
          total(0), lds_total(0), fN_global(false) {
        // Pass 0 lays every image out in LDS.  If the Q image of the two-hidden-layer exact trace does not fit that way
        // (8 hidden tiles with 8 state k-steps), pass 1 moves the last-layer image fN behind the LDS part: the kernel stages
        // [0, lds_total) and reads fN from the packed image in global memory (one D-row product per evaluation).
        for (int pass = 0; pass < 2; ++pass) {
            const bool fN_out = pass == 1;
            int o = 0;
            f1z = o; o += imgA(HT, KGZ);
            f1y = o; o += imgA(HT, KGC);
            fh = o;  o += (L - 1) * (arith_ ? imgH16(HT_) : imgA(HT_, HT_));
            fN = o;  if (!fN_out) o += imgA(DT, HT);
            bN = o;  if (with_bwd) o += imgA(HT, KGZ);
            bh = o;  if (with_bwd) o += (L - 1) * (arith_ ? imgH16(HT_) : imgA(HT_, HT_));
            b1 = o;  if (with_bwd) o += imgA(DT, HT);
            v_b1 = o;  o += vecC(HT);
            v_w1t = o; o += vecC(HT);
            v_bh = o;  o += (L - 1) * vecC(HT);
            v_bN = o;  o += vecC(DT);
            const int o_fixed = o;   // everything that must be in LDS besides the optional sections below
            // two hidden layers (the reference's default architecture): tr J = act'_2^T Q act'_1 needs one H x H product
            qtr = -1;
            if (!with_bwd && L_ == 2 && arith_ == 0 && (o + imgA(HT_, HT_)) * 4 <= 160 * 1024) { qtr = o; o += imgA(HT_, HT_); }
            // otherwise the exact trace pushes D unit tangents: column p of W_1 and row p of W_N in accumulator layout save the
            // first and last product of each (only where they still fit the 160 KB of LDS)
            v_wNr = -1;
            if (!with_bwd && qtr < 0 && (o + 4 * ZR_ * vecC(HT_)) * 4 <= 160 * 1024) { v_wNr = o; o += 4 * ZR_ * vecC(HT_); }
            v_w1c = -1;
            if (!with_bwd && qtr < 0 && (o + 4 * ZR_ * vecC(HT_)) * 4 <= 160 * 1024) { v_w1c = o; o += 4 * ZR_ * vecC(HT_); }
            lds_total = (o + 3) / 4 * 4;
            fN_global = fN_out;
            if (fN_out) { fN = lds_total; o = lds_total + imgA(DT, HT); }
            total = (o + 3) / 4 * 4;
            const bool want_retry = pass == 0 && !with_bwd && L_ == 2 && arith_ == 0 && qtr < 0 &&
                                    (o_fixed - imgA(DT, HT) + imgA(HT_, HT_)) * 4 <= 160 * 1024;
            if (!want_retry) break;
        }
    }
};

// feature computed by M-tile row i of tile mt
constexpr int mfma_rowmap(int mt, int i) { return 16 * mt + 4 * (i & 3) + (i >> 2); }

}  // namespace cnf
