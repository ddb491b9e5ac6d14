// Gradient exchange for hosts that have no torch.distributed: a thin C layer over RCCL (one communicator per process
// and GPU, in-place sum of one bucket of the flat gradient buffer on a stream).  RCCL is resolved at run time
// (dlopen / dlsym): the library carries no link-time dependency on it, and a process that already holds a copy (a
// Python host with PyTorch) keeps using that one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

#include "common.h"

namespace {

struct RcclApi {
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char* (*get_error_string)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        // a copy already in the process first (PyTorch ships its own), then the ROCm installation's
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        if (!h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
        if (!h) { api.why = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
        api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
        api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
        api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(dlsym(h, "ncclAllReduce"));
        api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
        api.get_error_string = reinterpret_cast<decltype(api.get_error_string)>(dlsym(h, "ncclGetErrorString"));
        api.ok = api.get_unique_id && api.comm_init_rank && api.all_reduce && api.comm_destroy;
        if (!api.ok) api.why = "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
    });
    return api;
}

int rccl_fail(const char* what, ncclResult_t r) {
    RcclApi& a = rccl();
    mi_set_error(std::string(what) + ": " + (a.get_error_string ? a.get_error_string(r) : "RCCL error") + " (" + std::to_string((int)r) + ")");
    return -2;
}

}  // namespace

struct mi_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
};

extern "C" {

int mi_comm_unique_id(void* id128) {
    MI_CHECK(id128, "null argument");
    RcclApi& a = rccl();
    if (!a.ok) { mi_set_error(a.why); return -3; }
    static_assert(sizeof(ncclUniqueId) == 128, "the header promises 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = a.get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    __builtin_memcpy(id128, &id, sizeof(id));
    return 0;
}

int mi_comm_init(int nranks, int rank, const void* id128, mi_comm** out) {
    MI_CHECK(out && id128, "null argument");
    MI_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "rank must lie in [0, nranks)");
    RcclApi& a = rccl();
    if (!a.ok) { mi_set_error(a.why); return -3; }
    ncclUniqueId id;
    __builtin_memcpy(&id, id128, sizeof(id));
    mi_comm* c = new mi_comm();
    ncclResult_t r = a.comm_init_rank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return rccl_fail("ncclCommInitRank", r); }
    c->nranks = nranks; c->rank = rank;
    *out = c;
    return 0;
}

int mi_comm_allreduce_bucket(mi_comm* c, float* grads, int64_t offset, int64_t count, void* stream) {
    MI_CHECK(c && c->comm && grads, "null argument");
    MI_CHECK(offset >= 0 && count >= 0, "bad range");
    if (count == 0) return 0;
    RcclApi& a = rccl();
    ncclResult_t r = a.all_reduce(grads + offset, grads + offset, (size_t)count, ncclFloat, ncclSum, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    return 0;
}

int mi_comm_destroy(mi_comm* c) {
    if (!c) return 0;
    int rc = 0;
    if (c->comm) {
        ncclResult_t r = rccl().comm_destroy(c->comm);
        if (r != ncclSuccess) rc = rccl_fail("ncclCommDestroy", r);
    }
    delete c;
    return rc;
}

}  // extern "C"
