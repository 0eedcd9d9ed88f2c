// cnf_mfma_generic_zr8.hip — zero-padded per-wave solve instances with 8 state k-steps (16 < D <= 32).
//
// The reference's default constructor builds D = 2 n + 1 state rows and hidden width 8 n + 8 for
// nvariables = n (src/core/icnf.jl:62-71): n = 8 .. 15 gives D = 17 .. 31 and H = 72 .. 128 — beyond the
// 4 state k-steps of cnf_mfma_generic.hip, below the cooperative kernel's widths.  These instances keep
// such flows (and any other chain with D <= 32, H <= 128) on the fused path: z, eps and the RK stage
// derivatives use 8 registers per lane each, the D-row products two M-tiles.
#define CNF_WITH_DEVICE_CONTROLLER 1
#include "cnf_mfma_kernel.h"

namespace cnf {

#define VJP_ACT(ACT) ((ACT) == CNF_ACT_TANH ? CNF_ACT_TANH_PRESCALED : (ACT))
#define GEN8(HT, L, ACT, NT)                                   \
    MFMA_INST(HT, L, 8, 0, VJP_ACT(ACT), ENG_VJP, 1, 1, NT),   \
    MFMA_INST(HT, L, 8, 4, VJP_ACT(ACT), ENG_VJP, 1, 1, NT),   \
    MFMA_INST(HT, L, 8, 0, ACT, ENG_TAN, 1, 0, NT),            \
    MFMA_INST(HT, L, 8, 4, ACT, ENG_TAN, 1, 0, NT)
// the reference's default nets (two hidden layers of softplus, nvariables 8..15), unconditioned: with the one-launch adaptive twins
#define GEN8_DEF(HT, L, ACT, NT)                                  \
    MFMA_INST_AD(HT, L, 8, 0, VJP_ACT(ACT), ENG_VJP, 1, 1, NT),   \
    MFMA_INST(HT, L, 8, 4, VJP_ACT(ACT), ENG_VJP, 1, 1, NT),      \
    MFMA_INST_AD(HT, L, 8, 0, ACT, ENG_TAN, 1, 0, NT),            \
    MFMA_INST(HT, L, 8, 4, ACT, ENG_TAN, 1, 0, NT)
#define GEN8_ACT(HT, NT)                                                         \
    GEN8(HT, 2, CNF_ACT_TANH, NT), GEN8(HT, 3, CNF_ACT_TANH, NT),                \
    GEN8_DEF(HT, 2, CNF_ACT_SOFTPLUS, NT), GEN8(HT, 3, CNF_ACT_SOFTPLUS, NT)

// 7 hidden tiles, two hidden layers, tangent engine: H = 104 / 112 (default nets for nvariables = 12, 13) on the 8-tile instance
// would not fit the Q image of the exact-trace shortcut in LDS; on 7 tiles it does
#define GEN8_TAN7(ACT) MFMA_INST(7, 2, 8, 0, ACT, ENG_TAN, 1, 0, 256), MFMA_INST(7, 2, 8, 4, ACT, ENG_TAN, 1, 0, 256)
#define GEN8_TAN7_DEF(ACT) MFMA_INST_AD(7, 2, 8, 0, ACT, ENG_TAN, 1, 0, 256), MFMA_INST(7, 2, 8, 4, ACT, ENG_TAN, 1, 0, 256)

// 5 and 7 hidden tiles for the reference's default nets at nvariables = 8, 9 (H = 72, 80) and 12, 13 (H = 104, 112): on the 6- / 8-tile
// instances their H x H products multiplied 1.4 - 1.8 x / 1.3 - 1.5 x the tiles they have (round 4: nvariables = 12 TrainMode 14.3 ms,
// the same as nvariables = 15)
#define GEN8_DEF57 MFMA_INST_AD(5, 2, 8, 0, CNF_ACT_SOFTPLUS, ENG_VJP, 1, 1, 256), MFMA_INST_AD(5, 2, 8, 0, CNF_ACT_SOFTPLUS, ENG_TAN, 1, 0, 256), \
                   MFMA_INST_AD(7, 2, 8, 0, CNF_ACT_SOFTPLUS, ENG_VJP, 1, 1, 256)

static const Inst kGenericZr8[] = {
    GEN8_TAN7(CNF_ACT_TANH), GEN8_TAN7_DEF(CNF_ACT_SOFTPLUS), GEN8_DEF57,
    GEN8_ACT(2, 512), GEN8_ACT(4, 256), GEN8_ACT(6, 256), GEN8_ACT(8, 256),   // HT = 4 spills at 2 waves/SIMD (8 state k-steps)
};

const Inst* mfma_generic_zr8_insts(int* count) {
    *count = (int)(sizeof(kGenericZr8) / sizeof(kGenericZr8[0]));
    return kGenericZr8;
}

}  // namespace cnf
