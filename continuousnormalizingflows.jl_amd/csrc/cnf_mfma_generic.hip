// cnf_mfma_generic.hip — zero-padded instantiations of the per-wave fused solve kernel.
//
// The specialised table in cnf_mfma.hip matches (D, C) exactly (BASELINE shapes).  These cover
// every other Dense chain with H <= 128 and L in {2,3} (H <= 64 for L in {1,4}), D <= 16 (D <= 32: cnf_mfma_generic_zr8.hip), C = 0 or <= 16,
// tanh or softplus, K = 1 (the unconditioned ones with their device-controlled adaptive twins, MFMA_INST_AD): the state/condition k-steps are padded to 4 (zero rows in the operand
// images, zero registers in the state), which costs at most 2-3 wasted MFMAs per product against
// the 40x the generic SIMT path would cost.  E.g. the reference's default net for nvariables = 2
// (D = 5, n_in = 6, hidden 24, softplus) runs on <HT=2, L=2, ZR=4, CR=0, softplus>.
#define CNF_WITH_DEVICE_CONTROLLER 1
#include "cnf_mfma_kernel.h"

// This file is compiled twice - tanh here, softplus through cnf_mfma_generic_softplus.hip - so that the two halves of the
// table (each instance with its adaptive-Tsit5 and VCABM twins) build in parallel.
#ifndef GEN_ACTIVATION
#define GEN_ACTIVATION CNF_ACT_TANH
#define GEN_TABLE_FN mfma_generic_insts
#endif

namespace cnf {

// VJP instances of tanh nets run the pre-scaled tanh (forward images carry the -2 log2 e factor)
#define VJP_ACT(ACT) ((ACT) == CNF_ACT_TANH ? CNF_ACT_TANH_PRESCALED : (ACT))
#define GEN4(HT, L, ACT, NT)                                   \
    MFMA_INST_AD(HT, L, 4, 0, VJP_ACT(ACT), ENG_VJP, 1, 1, NT),   \
    MFMA_INST(HT, L, 4, 4, VJP_ACT(ACT), ENG_VJP, 1, 1, NT),   \
    MFMA_INST_AD(HT, L, 4, 0, ACT, ENG_TAN, 1, 0, NT),            \
    MFMA_INST(HT, L, 4, 4, ACT, ENG_TAN, 1, 0, NT)
// two hidden layers of softplus are the reference's default nets (conditioned ones included): all four get the twins
#define GEN4_ALL_AD(HT, L, ACT, NT)                               \
    MFMA_INST_AD(HT, L, 4, 0, VJP_ACT(ACT), ENG_VJP, 1, 1, NT),   \
    MFMA_INST_AD(HT, L, 4, 4, VJP_ACT(ACT), ENG_VJP, 1, 1, NT),   \
    MFMA_INST_AD(HT, L, 4, 0, ACT, ENG_TAN, 1, 0, NT),            \
    MFMA_INST_AD(HT, L, 4, 4, ACT, ENG_TAN, 1, 0, NT)
#ifdef GEN_DEFAULT_NETS
#define GEN_ACT(HT, NT) GEN4_ALL_AD(HT, 2, GEN_ACTIVATION, NT), GEN4(HT, 3, GEN_ACTIVATION, NT)
#else
#define GEN_ACT(HT, NT) GEN4(HT, 2, GEN_ACTIVATION, NT), GEN4(HT, 3, GEN_ACTIVATION, NT)
#endif

// 1 and 4 hidden layers (H <= 64)
#define GEN_L14(HT, NT) GEN4(HT, 1, GEN_ACTIVATION, NT), GEN4(HT, 4, GEN_ACTIVATION, NT)

static const Inst kGeneric[] = {
    GEN_ACT(1, 512), GEN_ACT(2, 512), GEN_ACT(3, 512), GEN_ACT(4, 512), GEN_ACT(6, 256), GEN_ACT(8, 256),
    GEN_L14(1, 512), GEN_L14(2, 512), GEN_L14(3, 512), GEN_L14(4, 512),
};

const Inst* GEN_TABLE_FN(int* count) {
    *count = (int)(sizeof(kGeneric) / sizeof(kGeneric[0]));
    return kGeneric;
}

}  // namespace cnf
