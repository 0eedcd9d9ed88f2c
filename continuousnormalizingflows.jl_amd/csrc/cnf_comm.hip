// cnf_comm.hip — the one exchange step of the sharded hot path, behind the C ABI: the mean in `loss`
// (src/core/icnf.jl:636) over column shards = an all-reduce of four partial sums and the column count
// (SURVEY.md section 8(e)), plus the two sums the "next" rows need (the parameter gradient, the adaptive
// solvers' error norms).  RCCL over xGMI; one communicator rank per process per GPU, or — for a single
// host process driving several GPUs, which is how a Julia host without MPI would do it — one communicator
// per device from cnf_comm_init_all.
//
// RCCL is resolved with dlopen at first use (a process that already holds librccl.so.1 — torch does — shares
// that copy), so libcnf_hip.so has no link-time dependency on it and every other entry point works without it.
#include <dlfcn.h>

#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "cnf_handle.h"

namespace cnf {
namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_id = nullptr;
    decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclCommInitAll) init_all = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) err_str = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;        // optional: what the communicator itself reports
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    bool ok = false;
};

Rccl load_rccl() {
    Rccl r;
    // a copy already mapped into the process first (RTLD_NOLOAD), so both sides share one RCCL and one HIP runtime
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (r.lib) break;
    }
    if (!r.lib)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
    if (!r.lib) return r;
    r.get_id = (decltype(r.get_id))dlsym(r.lib, "ncclGetUniqueId");
    r.init_rank = (decltype(r.init_rank))dlsym(r.lib, "ncclCommInitRank");
    r.init_all = (decltype(r.init_all))dlsym(r.lib, "ncclCommInitAll");
    r.destroy = (decltype(r.destroy))dlsym(r.lib, "ncclCommDestroy");
    r.all_reduce = (decltype(r.all_reduce))dlsym(r.lib, "ncclAllReduce");
    r.err_str = (decltype(r.err_str))dlsym(r.lib, "ncclGetErrorString");
    r.group_start = (decltype(r.group_start))dlsym(r.lib, "ncclGroupStart");
    r.group_end = (decltype(r.group_end))dlsym(r.lib, "ncclGroupEnd");
    r.comm_count = (decltype(r.comm_count))dlsym(r.lib, "ncclCommCount");
    r.comm_user_rank = (decltype(r.comm_user_rank))dlsym(r.lib, "ncclCommUserRank");
    r.ok = r.get_id && r.init_rank && r.init_all && r.destroy && r.all_reduce && r.err_str && r.group_start && r.group_end;
    return r;
}

Rccl& rccl() {
    static Rccl r = load_rccl();
    return r;
}

// [sum(-logp), sum E, sum n, sum A] (float) + this shard's column count -> five doubles, so that the combination
// over ranks does not depend on the rank count beyond the fp32 rounding of the per-rank sums
__global__ void pack_loss_kernel(const float* __restrict__ sums4, double count, double* __restrict__ out5) {
    const int i = threadIdx.x;
    if (i < 4) out5[i] = (double)sums4[i];
    if (i == 4) out5[4] = count;
}

}  // namespace
}  // namespace cnf

struct cnf_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
};

#define RCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess)                                                                             \
            return ::cnf::api_fail(CNF_ERR_COMM, std::string(#expr) + ": " + ::cnf::rccl().err_str(_r));   \
    } while (0)

static int need_rccl() {
    if (!cnf::rccl().ok) return cnf::api_fail(CNF_ERR_COMM, "librccl.so.1 could not be loaded (dlopen)");
    return CNF_OK;
}

extern "C" {

int cnf_comm_unique_id(void* id_out) {
    if (!id_out) return cnf::api_fail(CNF_ERR_INVALID, "cnf_comm_unique_id: id_out is NULL");
    if (int rc = need_rccl()) return rc;
    static_assert(CNF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "cnf.h must carry RCCL's unique-id size");
    ncclUniqueId id;
    RCCL_TRY(cnf::rccl().get_id(&id));
    std::memcpy(id_out, id.internal, CNF_COMM_ID_BYTES);
    return CNF_OK;
}

int cnf_comm_init(cnf_comm** out, int rank, int nranks, const void* id, int device_id) {
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks)
        return cnf::api_fail(CNF_ERR_INVALID, "cnf_comm_init: bad rank / nranks / id");
    if (int rc = need_rccl()) return rc;
    cnf::DeviceGuard g(device_id);
    if (!g.ok) return cnf::api_fail(CNF_ERR_NO_DEVICE, "cnf_comm_init: hipSetDevice failed");
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, CNF_COMM_ID_BYTES);
    cnf_comm* c = new cnf_comm();
    c->rank = rank; c->nranks = nranks; c->device = device_id;
    ncclResult_t r = cnf::rccl().init_rank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return cnf::api_fail(CNF_ERR_COMM, std::string("ncclCommInitRank: ") + cnf::rccl().err_str(r));
    }
    *out = c;
    return CNF_OK;
}

int cnf_comm_init_all(cnf_comm** out, int ndev, const int* devs) {
    if (!out || ndev < 1 || !devs) return cnf::api_fail(CNF_ERR_INVALID, "cnf_comm_init_all: bad arguments");
    if (int rc = need_rccl()) return rc;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1)
        return cnf::api_fail(CNF_ERR_NO_DEVICE, "cnf_comm_init_all: no HIP device");
    for (int i = 0; i < ndev; ++i) {
        if (devs[i] < 0 || devs[i] >= have)
            return cnf::api_fail(CNF_ERR_INVALID, "cnf_comm_init_all: device index out of range");
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return cnf::api_fail(CNF_ERR_INVALID, "cnf_comm_init_all: a device is listed twice");
    }
    std::vector<ncclComm_t> comms((size_t)ndev);
    RCCL_TRY(cnf::rccl().init_all(comms.data(), ndev, devs));
    for (int i = 0; i < ndev; ++i) {
        cnf_comm* c = new cnf_comm();
        c->comm = comms[(size_t)i]; c->rank = i; c->nranks = ndev; c->device = devs[i];
        out[i] = c;
    }
    return CNF_OK;
}

int cnf_comm_destroy(cnf_comm* c) {
    if (!c) return CNF_OK;
    if (c->comm && cnf::rccl().ok) (void)cnf::rccl().destroy(c->comm);
    delete c;
    return CNF_OK;
}

// rank / size as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount): a host that prints
// cnf_comm_size() has proof of how many ranks RCCL joined, not an echo of the number it passed to cnf_comm_init
int cnf_comm_rank(const cnf_comm* c) {
    if (!c) return CNF_ERR_INVALID;
    int v = c->rank;
    if (c->comm && cnf::rccl().comm_user_rank && cnf::rccl().comm_user_rank(c->comm, &v) != ncclSuccess) return CNF_ERR_COMM;
    return v;
}
int cnf_comm_size(const cnf_comm* c) {
    if (!c) return CNF_ERR_INVALID;
    int v = c->nranks;
    if (c->comm && cnf::rccl().comm_count && cnf::rccl().comm_count(c->comm, &v) != ncclSuccess) return CNF_ERR_COMM;
    return v;
}

int cnf_comm_group_start(void) {
    if (int rc = need_rccl()) return rc;
    RCCL_TRY(cnf::rccl().group_start());
    return CNF_OK;
}
int cnf_comm_group_end(void) {
    if (int rc = need_rccl()) return rc;
    RCCL_TRY(cnf::rccl().group_end());
    return CNF_OK;
}

int cnf_allreduce_sum(cnf_comm* c, void* buf, size_t count, int dtype, void* stream) {
    if (!c || !c->comm || (!buf && count)) return cnf::api_fail(CNF_ERR_INVALID, "cnf_allreduce_sum: bad arguments");
    if (dtype != CNF_DTYPE_F32 && dtype != CNF_DTYPE_F64)
        return cnf::api_fail(CNF_ERR_INVALID, "cnf_allreduce_sum: dtype must be CNF_DTYPE_F32 or CNF_DTYPE_F64");
    if (count == 0) return CNF_OK;
    cnf::DeviceGuard g(c->device);
    RCCL_TRY(cnf::rccl().all_reduce(buf, buf, count, dtype == CNF_DTYPE_F32 ? ncclFloat32 : ncclFloat64, ncclSum, c->comm,
                                    (hipStream_t)stream));
    return CNF_OK;
}

int cnf_allreduce_loss(cnf_comm* c, const float* sums4, int64_t B_local, double* out5, void* stream) {
    if (!c || !c->comm || !sums4 || !out5 || B_local < 0)
        return cnf::api_fail(CNF_ERR_INVALID, "cnf_allreduce_loss: bad arguments");
    cnf::DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cnf::pack_loss_kernel, dim3(1), dim3(64), 0, st, sums4, (double)B_local, out5);
    HIP_TRY(hipGetLastError());
    RCCL_TRY(cnf::rccl().all_reduce(out5, out5, 5, ncclFloat64, ncclSum, c->comm, st));
    return CNF_OK;
}

}  // extern "C"
