// placeholder until the fused MFMA kernels land
#include "cnf_internal.h"
namespace cnf {
struct MfmaPlan { int dummy; };
MfmaPlan* mfma_plan_create(const cnf_config&) { return nullptr; }
void mfma_plan_destroy(MfmaPlan* p) { delete p; }
size_t mfma_packed_bytes(const MfmaPlan*) { return 0; }
void mfma_pack(const MfmaPlan*, const float*, const size_t*, const size_t*, float*) {}
const char* mfma_plan_name(const MfmaPlan*) { return "none"; }
hipError_t mfma_solve(const MfmaPlan*, const float*, const SolveArgs&, hipStream_t) { return hipErrorNotSupported; }
}
