// Gradient-bucket all-reduce over RCCL behind the C ABI (rccl_bucket_allreduce_{init,launch,wait}; reference
// train.py:174-179 -- Lightning's strategy="ddp" averages the gradients over NCCL buckets).
//
// The package's default transport is torch.distributed (backend "nccl" = this very RCCL); these entries bind the
// library directly for a host that has no torch.distributed: one communicator per process (= per GPU), in-place average (or sum) of a
// contiguous fp32 bucket on the stream the caller names, and a stream-level wait.  RCCL is resolved at run time (dlopen: the copy the
// process already holds -- PyTorch-ROCm ships one -- or the system one), so the library loads, and everything else works, on hosts
// without it; the RCCL header is optional at build time too (declarations below).  Host code only: no kernels in this file.
// Verification status: exercised on the GPU with a world of ONE rank only (ncclAvg is then the identity); no multi-GPU node has run
// it, so the multi-rank average and the comm-stream ordering are unverified on hardware.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// Build host without the RCCL development headers: the few declarations the binding needs, as RCCL's public header states them
// (nccl.h ABI: a 128-byte unique id, an opaque communicator, result 0 = success, ncclFloat32 = 7, ncclSum = 0, ncclAvg = 4).  The
// library is still resolved at run time; without it the entries answer WJ_ERR_UNSUPPORTED.
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclAvg = 4 } ncclRedOp_t;
#endif
#include "common.h"
#include "../../include/wavjepa_hip.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = -1, world = 0;
    std::mutex mu;
};
Rccl& R() {
    static Rccl r;
    return r;
}

bool resolve(Rccl& r) {
    if (r.all_reduce) return true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) {                    // the copy already in the process first (two RCCL instances would not share state)
        r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.lib) break;
    }
    for (int i = 0; !r.lib && i < 2; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) return false;
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.lib, "ncclCommInitRank");
    r.all_reduce = (decltype(r.all_reduce))dlsym(r.lib, "ncclAllReduce");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.lib, "ncclCommDestroy");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
        r.all_reduce = nullptr;
        return false;
    }
    return true;
}

}  // namespace

// 128 opaque bytes that identify a new communicator: created on ONE rank, carried to the others by the host (any side channel),
// then handed to wj_rccl_bucket_allreduce_init on every rank.
extern "C" int wj_rccl_unique_id(void* out128) {
    if (!out128) return WJ_ERR_ARG;
    Rccl& r = R();
    std::lock_guard<std::mutex> lk(r.mu);
    if (!resolve(r)) return WJ_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r.get_unique_id(&id) != ncclSuccess) return WJ_ERR_LAUNCH;
    memcpy(out128, &id, sizeof(id));
    return WJ_OK;
}

extern "C" int wj_rccl_bucket_allreduce_init(const wj_rccl_init_args* a) {
    if (!a || !a->unique_id || a->world < 1 || a->rank < 0 || a->rank >= a->world) return WJ_ERR_ARG;
    Rccl& r = R();
    std::lock_guard<std::mutex> lk(r.mu);
    if (!resolve(r)) return WJ_ERR_UNSUPPORTED;
    if (r.comm) return WJ_ERR_ARG;                   // one communicator per process; wj_rccl_bucket_allreduce_finalize first
    ncclUniqueId id;
    memcpy(&id, a->unique_id, sizeof(id));
    if (r.comm_init_rank(&r.comm, a->world, id, a->rank) != ncclSuccess) {
        r.comm = nullptr;
        return WJ_ERR_LAUNCH;
    }
    r.rank = a->rank;
    r.world = a->world;
    return WJ_OK;
}

// buf[i] <- average (or sum) over the ranks of buf[i], i < count, in place, enqueued on `stream`
extern "C" int wj_rccl_bucket_allreduce_launch(const wj_rccl_launch_args* a, void* stream) {
    if (!a || !a->buf || a->count <= 0) return WJ_ERR_ARG;
    Rccl& r = R();
    if (!r.comm) return WJ_ERR_UNSUPPORTED;          // not initialised
    const ncclResult_t rc = r.all_reduce(a->buf, a->buf, (size_t)a->count, ncclFloat32, a->average ? ncclAvg : ncclSum, r.comm, (hipStream_t)stream);
    return rc == ncclSuccess ? WJ_OK : WJ_ERR_LAUNCH;
}

// `stream` waits (stream-level, the host does not) for everything enqueued so far on `on_stream` -- the stream the buckets were
// launched on: what the optimiser's stream calls before it reads the averaged gradients
extern "C" int wj_rccl_bucket_allreduce_wait(const wj_rccl_wait_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a) return WJ_ERR_ARG;
    if ((hipStream_t)a->on_stream == (hipStream_t)stream) return WJ_OK;
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return WJ_ERR_LAUNCH;
    hipError_t e = hipEventRecord(ev, (hipStream_t)a->on_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, ev, 0);
    (void)hipEventDestroy(ev);                       // released once the recorded work has completed
    return e == hipSuccess ? WJ_OK : WJ_ERR_LAUNCH;
}

extern "C" int wj_rccl_bucket_allreduce_finalize(void) {
    Rccl& r = R();
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.comm && r.comm_destroy) (void)r.comm_destroy(r.comm);
    r.comm = nullptr;
    r.rank = -1;
    r.world = 0;
    return WJ_OK;
}
