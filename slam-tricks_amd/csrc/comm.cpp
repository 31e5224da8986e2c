// comm.cpp -- the cross-rank sum of the sharded solvers as a native RCCL collective (one process
// per GPU, landmark shards, all-reduce of the reduced camera system over xGMI).
//
// The reference's callers are C++ executables (st20-g2o/src/src/test_ceres.cpp:7-19, st17-ceres/src/main.cpp):
// a C++ host creates one stba_comm per process (rank 0 makes the id with stba_comm_unique_id and hands it to
// the other ranks by whatever side channel it has -- MPI, a file, torch.distributed in bench.py), and passes it
// to stba_ba_set_comm / stba_pg_set_comm.  The engines then call ncclAllReduce(buf, buf, count, ncclDouble,
// ncclSum, comm, stream) on THEIR stream: no Python, no host round trip on the data path.
//
// RCCL is bound lazily with dlopen("librccl.so.1"): libstba.so has no link-time dependency on it (a
// single-GPU host needs no RCCL), and a process that already holds a librccl (PyTorch ships one) shares that
// copy instead of loading a second one.
#include <dlfcn.h>

#include <mutex>

#include "common.hpp"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;     // NCCL_UNIQUE_ID_BYTES, rccl.h:40-43
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclDouble = 8 }; // rccl.h:52,448,467

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("STBA_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string why;
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            const char* e = dlerror();            // (one call: dlerror() clears the message it returns)
            if (why.empty() && e) why = e;
        }
        if (!r.handle) { r.error = std::string("librccl not found: ") + (why.empty() ? "?" : why); return; }
        auto sym = [&](const char* s) { void* p = dlsym(r.handle, s); if (!p) r.error = std::string("librccl lacks ") + s; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

int rccl_ready() {
    Rccl& r = rccl();
    if (!r.error.empty() || !r.handle) return stba::fail(STBA_ERR_STATE, r.error.empty() ? "librccl not loaded" : r.error);
    return STBA_OK;
}

int rccl_fail(const char* what, int rc) {
    Rccl& r = rccl();
    return stba::fail(STBA_ERR_CALLBACK, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "error ") + " (" + std::to_string(rc) + ")");
}

}  // namespace

struct stba_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" {

int stba_comm_unique_id(char id[STBA_COMM_ID_BYTES]) {
    if (!id) return stba::fail(STBA_ERR_INVALID_ARGUMENT, "null id");
    STBA_TRY(stba::require_device());
    STBA_TRY(rccl_ready());
    ncclUniqueId u;
    const int rc = rccl().GetUniqueId(&u);
    if (rc != kNcclSuccess) return rccl_fail("ncclGetUniqueId", rc);
    static_assert(sizeof u.internal == STBA_COMM_ID_BYTES, "id size");
    memcpy(id, u.internal, STBA_COMM_ID_BYTES);
    return STBA_OK;
}

int stba_comm_create(stba_comm** out, const char id[STBA_COMM_ID_BYTES], int rank, int world_size, int device) {
    if (!out) return stba::fail(STBA_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (!id || world_size < 1 || rank < 0 || rank >= world_size) return stba::fail(STBA_ERR_INVALID_ARGUMENT, "stba_comm_create: bad id/rank/world");
    STBA_TRY(stba::require_device());
    STBA_TRY(rccl_ready());
    if (device >= 0) STBA_HIP(hipSetDevice(device));
    else STBA_HIP(hipGetDevice(&device));
    ncclUniqueId u;
    memcpy(u.internal, id, STBA_COMM_ID_BYTES);
    stba_comm* c = new stba_comm();
    c->rank = rank; c->world = world_size; c->device = device;
    const int rc = rccl().CommInitRank(&c->comm, world_size, u, rank);
    if (rc != kNcclSuccess) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return STBA_OK;
}

int stba_comm_destroy(stba_comm* c) {
    if (!c) return STBA_OK;
    int rc = kNcclSuccess;
    if (c->comm && rccl().CommDestroy) rc = rccl().CommDestroy(c->comm);
    delete c;
    return rc == kNcclSuccess ? STBA_OK : rccl_fail("ncclCommDestroy", rc);
}

int stba_comm_rank(const stba_comm* c, int* rank, int* world_size) {
    if (!c) return stba::fail(STBA_ERR_INVALID_ARGUMENT, "null communicator");
    if (rank) *rank = c->rank;
    if (world_size) *world_size = c->world;
    return STBA_OK;
}

int stba_comm_allreduce_sum(stba_comm* c, void* buf_dev, size_t count, void* hip_stream) {
    if (!c || !c->comm || (!buf_dev && count)) return stba::fail(STBA_ERR_INVALID_ARGUMENT, "stba_comm_allreduce_sum: bad argument");
    if (count == 0) return STBA_OK;
    const int rc = rccl().AllReduce(buf_dev, buf_dev, count, kNcclDouble, kNcclSum, c->comm, reinterpret_cast<hipStream_t>(hip_stream));
    return rc == kNcclSuccess ? STBA_OK : rccl_fail("ncclAllReduce", rc);
}

// stba_allreduce_fn with user = stba_comm*
int stba_comm_allreduce_hook(void* user, void* buf_dev, size_t count, void* hip_stream) {
    return stba_comm_allreduce_sum(static_cast<stba_comm*>(user), buf_dev, count, hip_stream) == STBA_OK ? 0 : 1;
}

int stba_ba_set_comm(stba_ba* ba, stba_comm* c) {
    if (!c) return stba_ba_set_allreduce(ba, nullptr, nullptr, 0, 1);
    return stba_ba_set_allreduce(ba, &stba_comm_allreduce_hook, c, c->rank, c->world);
}

int stba_pg_set_comm(stba_pg* pg, stba_comm* c) {
    if (!c) return stba_pg_set_allreduce(pg, nullptr, nullptr, 0, 1);
    return stba_pg_set_allreduce(pg, &stba_comm_allreduce_hook, c, c->rank, c->world);
}

}  // extern "C"
