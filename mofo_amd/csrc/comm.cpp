// RCCL communicator behind the C-ABI (mofo_comm_*): what the reference gets from
// torch.distributed's NCCL backend (utils.py:289-294) and DistributedDataParallel's gradient all-reduce
// (run_mae_pretraining.py:225-227), for a caller that binds libmofo_hip.so without torch.distributed.  The Python host
// (mofo_amd/dist.py) keeps using torch.distributed ("nccl" = RCCL) -- process-group plumbing it needs anyway -- so this is the
// stand-alone route: one communicator per rank, SUM all-reduce of a contiguous f32 range on the caller's stream.
// librccl is opened at run time (dlopen): a process that already loaded an RCCL (torch's) gets that one, and libmofo_hip.so
// carries no link-time dependency on it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "../../include/mofo_hip.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;          // NCCL_UNIQUE_ID_BYTES = 128
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7 };                                     // ncclDataType_t
enum { ncclSum = 0 };                                         // ncclRedOp_t

struct Api {
    int (*GetUniqueId)(ncclUniqueId*);
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
    int (*CommDestroy)(ncclComm_t);
    const char* (*GetErrorString)(int);
    bool ok;
};

Api* api() {
    static Api a;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // An RCCL that the process has ALREADY loaded (torch's) is reused: first the global scope, then RTLD_NOLOAD by name; only a
        // process without one maps the system library -- locally (no RTLD_GLOBAL: its symbols must not interpose on another copy).
        void* h = nullptr;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) h = dlopen(nullptr, RTLD_NOW);
        for (int pass = 0; pass < 2 && !h; ++pass)
            for (const char* n : names) {
                h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (h) break;
            }
        if (h) {
            // the unique id is passed by value as 128 bytes and the enums are restated above: refuse a library whose major
            // version is not the 2.x ABI they were taken from
            int (*GetVersion)(int*) = (int (*)(int*))dlsym(h, "ncclGetVersion");
            int ver = 0;
            if (!GetVersion || GetVersion(&ver) != ncclSuccess || ver < 20000 || ver >= 30000) h = nullptr;
        }
        if (h) {
            a.GetUniqueId = (int (*)(ncclUniqueId*))dlsym(h, "ncclGetUniqueId");
            a.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(h, "ncclCommInitRank");
            a.AllReduce = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
            a.CommDestroy = (int (*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
            a.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
            a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy;
        }
    }
    return &a;
}

struct Comm {
    ncclComm_t comm;
    int rank, world;
};

}  // namespace

#define RCCL_OR_FAIL(who)                                                                                   \
    Api* a = api();                                                                                         \
    if (!a->ok) MOFO_FAIL(MOFO_ERUNTIME, "%s: librccl.so could not be opened", who)

extern "C" int mofo_comm_unique_id(void* id128) {
    if (!id128) MOFO_FAIL(MOFO_EINVAL, "mofo_comm_unique_id: null pointer");
    RCCL_OR_FAIL("mofo_comm_unique_id");
    ncclUniqueId id;
    const int rc = a->GetUniqueId(&id);
    if (rc != ncclSuccess) MOFO_FAIL(MOFO_ERUNTIME, "mofo_comm_unique_id: %s", a->GetErrorString ? a->GetErrorString(rc) : "rccl error");
    memcpy(id128, &id, sizeof(id));
    return MOFO_OK;
}

extern "C" int mofo_comm_init(const void* id128, int rank, int world, void** comm_out) {
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) MOFO_FAIL(MOFO_EINVAL, "mofo_comm_init: bad arguments");
    RCCL_OR_FAIL("mofo_comm_init");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm* c = (Comm*)malloc(sizeof(Comm));                   // the handle is the library's own object (owned by the C side)
    if (!c) MOFO_FAIL(MOFO_ERUNTIME, "mofo_comm_init: out of memory");
    const int rc = a->CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) {
        free(c);
        MOFO_FAIL(MOFO_ERUNTIME, "mofo_comm_init: %s", a->GetErrorString ? a->GetErrorString(rc) : "rccl error");
    }
    c->rank = rank;
    c->world = world;
    *comm_out = c;
    return MOFO_OK;
}

extern "C" int mofo_comm_allreduce_f32(void* comm, float* buf, long long n, void* stream) {
    if (!comm || !buf || n <= 0) MOFO_FAIL(MOFO_EINVAL, "mofo_comm_allreduce_f32: bad arguments");
    RCCL_OR_FAIL("mofo_comm_allreduce_f32");
    Comm* c = (Comm*)comm;
    const int rc = a->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
    if (rc != ncclSuccess) MOFO_FAIL(MOFO_ERUNTIME, "mofo_comm_allreduce_f32: %s", a->GetErrorString ? a->GetErrorString(rc) : "rccl error");
    return MOFO_OK;
}

extern "C" int mofo_comm_destroy(void* comm) {
    if (!comm) return MOFO_OK;
    RCCL_OR_FAIL("mofo_comm_destroy");
    Comm* c = (Comm*)comm;
    const int rc = a->CommDestroy(c->comm);
    free(c);
    if (rc != ncclSuccess) MOFO_FAIL(MOFO_ERUNTIME, "mofo_comm_destroy: %s", a->GetErrorString ? a->GetErrorString(rc) : "rccl error");
    return MOFO_OK;
}
