// cnf_coop_grad_softplus.hip — the softplus instances of the cooperative reverse sweep (cnf_coop_grad.hip): the reference's
// default architecture (two softplus layers, src/core/icnf.jl:66-71) at the widths the extended cooperative kernel serves.
#define CG_ACT_SOFTPLUS 1
#include "cnf_coop_grad.hip"
