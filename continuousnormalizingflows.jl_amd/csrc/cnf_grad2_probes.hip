// cnf_grad2_probes.hip - the barrier-free gradient kernel of cnf_grad2.hip instantiated for several Hutchinson probes (BASELINE cfg3:
// RNODE, K = 4).  Instantiation-only translation unit.
#define G2_MULTI true
#define G2_FIND grad2_probes_kernel
#include "cnf_grad2.hip"
