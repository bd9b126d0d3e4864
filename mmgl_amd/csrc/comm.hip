// Gradient-exchange entry points of the C ABI over RCCL (xGMI): the collectives of the data-parallel path as plain C calls.
// replaces: torch.distributed's NCCL process group as used by the reference's DDP wiring --
//   language_modelling/run_generation.py:283 (init_process_group("nccl")), :317-319 (DistributedDataParallel: bucketed gradient
//   all-reduce + constructor broadcast), language_modelling/utils.py:113-118 (meter all-reduce), run_generation.py:608-616 (eval
//   all_gather) -- for a caller that binds libmmgl_hip.so without torch.distributed (option B).
// mmgl_amd's own trainer issues the same collectives through torch.distributed's "nccl" backend, which IS RCCL (one communicator
// per process, shared with the rest of the torch program); these entry points are the same five calls for a host that has none.
//
// RCCL is resolved at run time (dlopen: the copy the process already holds -- PyTorch ships its own librccl.so -- else the
// system's), so the library loads, and every other entry point works, on a machine without it.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <mutex>
#include <string>

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // optional: only to validate a broadcast root
    bool ok = false;
    std::string why;           // dlerror() text captured once, at dlopen time (dlerror() clears itself when read)
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);          // the copy already mapped into the process (torch's), if any
            if (r.h) break;
        }
        if (!r.h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
                r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.h) break;
            }
        if (!r.h) {
            const char* e = dlerror();
            r.why = e ? e : "dlopen failed";
            return;
        }
#define RCCL_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.h, sym)
        RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
        RCCL_SYM(CommInitRank, "ncclCommInitRank");
        RCCL_SYM(CommDestroy, "ncclCommDestroy");
        RCCL_SYM(AllReduce, "ncclAllReduce");
        RCCL_SYM(AllGather, "ncclAllGather");
        RCCL_SYM(Broadcast, "ncclBroadcast");
        RCCL_SYM(GetErrorString, "ncclGetErrorString");
        RCCL_SYM(CommCount, "ncclCommCount");
#undef RCCL_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather && r.Broadcast && r.GetErrorString;
        if (!r.ok) r.why = "symbols missing";
    });
    return r;
}

int comm_dtype(const char* who, int dtype, ncclDataType_t* out) {
    switch (dtype) {
        case MMGL_F32: *out = ncclFloat32; return MMGL_OK;
        case MMGL_BF16: *out = ncclBfloat16; return MMGL_OK;
        case MMGL_COMM_I64: *out = ncclInt64; return MMGL_OK;
        default: MMGL_FAIL(MMGL_ERR_INVALID, "%s: dtype %d (MMGL_F32, MMGL_BF16 or MMGL_COMM_I64)", who, dtype);
    }
}

#define RCCL_REQUIRE(who)                                                                                        \
    Rccl& R = rccl();                                                                                            \
    if (!R.ok) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "%s: librccl.so could not be loaded (%s)", who, R.why.c_str())
#define RCCL_CHECK(who, expr)                                                                                    \
    do {                                                                                                         \
        ncclResult_t rc_ = (expr);                                                                               \
        if (rc_ != ncclSuccess) MMGL_FAIL(MMGL_ERR_HIP, "%s: RCCL: %s", who, R.GetErrorString(rc_));             \
    } while (0)

}  // namespace

extern "C" int mmgl_comm_unique_id(void* out128) {
    MMGL_CHECK_ARG(out128, "mmgl_comm_unique_id: null pointer");
    RCCL_REQUIRE("mmgl_comm_unique_id");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RCCL_CHECK("mmgl_comm_unique_id", R.GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return MMGL_OK;
}

extern "C" int mmgl_comm_init(int rank, int world, const void* unique_id, void** comm) {
    MMGL_CHECK_ARG(unique_id && comm && world > 0 && rank >= 0 && rank < world, "mmgl_comm_init: bad arguments (rank %d of %d)", rank, world);
    RCCL_REQUIRE("mmgl_comm_init");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    RCCL_CHECK("mmgl_comm_init", R.CommInitRank(&c, world, id, rank));       // binds to the caller's current HIP device
    *comm = (void*)c;
    return MMGL_OK;
}

extern "C" int mmgl_allreduce_sum(void* comm, void* buf, size_t count, int dtype, void* stream) {
    MMGL_CHECK_ARG(comm && (buf || !count), "mmgl_allreduce_sum: null pointer");
    RCCL_REQUIRE("mmgl_allreduce_sum");
    ncclDataType_t dt;
    if (int rc = comm_dtype("mmgl_allreduce_sum", dtype, &dt)) return rc;
    if (!count) return MMGL_OK;
    RCCL_CHECK("mmgl_allreduce_sum", R.AllReduce(buf, buf, count, dt, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
    return MMGL_OK;
}

extern "C" int mmgl_allgather(void* comm, const void* in, void* out, size_t count_per_rank, int dtype, void* stream) {
    MMGL_CHECK_ARG(comm && ((in && out) || !count_per_rank), "mmgl_allgather: null pointer");
    RCCL_REQUIRE("mmgl_allgather");
    ncclDataType_t dt;
    if (int rc = comm_dtype("mmgl_allgather", dtype, &dt)) return rc;
    if (!count_per_rank) return MMGL_OK;
    RCCL_CHECK("mmgl_allgather", R.AllGather(in, out, count_per_rank, dt, (ncclComm_t)comm, (hipStream_t)stream));
    return MMGL_OK;
}

extern "C" int mmgl_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream) {
    MMGL_CHECK_ARG(comm && (buf || !count) && root >= 0, "mmgl_broadcast: bad arguments");
    RCCL_REQUIRE("mmgl_broadcast");
    ncclDataType_t dt;
    if (int rc = comm_dtype("mmgl_broadcast", dtype, &dt)) return rc;
    if (R.CommCount) {
        int world = 0;
        RCCL_CHECK("mmgl_broadcast", R.CommCount((ncclComm_t)comm, &world));
        MMGL_CHECK_ARG(root < world, "mmgl_broadcast: root %d of a %d-rank communicator", root, world);
    }
    if (!count) return MMGL_OK;
    RCCL_CHECK("mmgl_broadcast", R.Broadcast(buf, buf, count, dt, root, (ncclComm_t)comm, (hipStream_t)stream));
    return MMGL_OK;
}

extern "C" int mmgl_comm_destroy(void* comm) {
    if (!comm) return MMGL_OK;
    RCCL_REQUIRE("mmgl_comm_destroy");
    RCCL_CHECK("mmgl_comm_destroy", R.CommDestroy((ncclComm_t)comm));
    return MMGL_OK;
}
