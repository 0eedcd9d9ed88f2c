// cnf_mfma_generic_probes.hip — zero-padded per-wave solve instances for K > 1 Hutchinson probes.
//
// The probe vectors of a tile live in registers (eps[KP][ZR]); KP is the instance's capacity and the
// live count K <= KP arrives at run time (KArgs::K): the probe loop of dyn_eval is rolled, so one
// instance serves K = 2 .. KP.  Shapes as in cnf_mfma_generic.hip (VJP engine only, no hoisting:
// c = W_N^T eps_k is recomputed per probe).  BASELINE cfg3 (D = 8, 3x64, K = 4) keeps its exact-shape
// instance in cnf_mfma.hip; these cover the other RNODE-style configurations.
#include "cnf_mfma_kernel.h"

namespace cnf {

#define VJP_ACT(ACT) ((ACT) == CNF_ACT_TANH ? CNF_ACT_TANH_PRESCALED : (ACT))
#define KCAP 8
#define GENP2(HT, L, ACT, NT)                                    \
    MFMA_INST(HT, L, 4, 0, VJP_ACT(ACT), ENG_VJP, KCAP, 0, NT),  \
    MFMA_INST(HT, L, 4, 4, VJP_ACT(ACT), ENG_VJP, KCAP, 0, NT)
#define GENP_L23(HT, NT)                                                         \
    GENP2(HT, 2, CNF_ACT_TANH, NT), GENP2(HT, 3, CNF_ACT_TANH, NT),              \
    GENP2(HT, 2, CNF_ACT_SOFTPLUS, NT), GENP2(HT, 3, CNF_ACT_SOFTPLUS, NT)
#define GENP_L14(HT, NT)                                                         \
    GENP2(HT, 1, CNF_ACT_TANH, NT), GENP2(HT, 4, CNF_ACT_TANH, NT),              \
    GENP2(HT, 1, CNF_ACT_SOFTPLUS, NT), GENP2(HT, 4, CNF_ACT_SOFTPLUS, NT)

static const Inst kGenericProbes[] = {
    GENP_L23(1, 512), GENP_L23(2, 512), GENP_L23(3, 512), GENP_L23(4, 512), GENP_L23(6, 256), GENP_L23(8, 256),
    GENP_L14(1, 512), GENP_L14(2, 512), GENP_L14(3, 512), GENP_L14(4, 512),
};

const Inst* mfma_generic_probe_insts(int* count) {
    *count = (int)(sizeof(kGenericProbes) / sizeof(kGenericProbes[0]));
    return kGenericProbes;
}

}  // namespace cnf
