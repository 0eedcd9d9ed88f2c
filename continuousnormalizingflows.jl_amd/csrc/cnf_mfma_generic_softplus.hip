// cnf_mfma_generic_softplus.hip - the softplus half of the zero-padded instance table (see cnf_mfma_generic.hip).
#define GEN_ACTIVATION CNF_ACT_SOFTPLUS
#define GEN_DEFAULT_NETS 1
#define GEN_TABLE_FN mfma_generic_softplus_insts
#include "cnf_mfma_generic.hip"
