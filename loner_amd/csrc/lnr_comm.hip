// RCCL binding of the keyframe-sharded mapping loop (loner_amd/mapping/sharding.py): the collectives of one iteration -
// the front all-gather, the gradient all-reduce / reduce-scatter, the parameter all-gather, the occupancy all-reduce - enqueued
// straight onto the caller's HIP stream.
//
// Why not torch.distributed for these: ProcessGroupNCCL puts every collective on a stream of its own and hands over with events both
// ways; measured on a one-keyframe rank (profiles/r05_host_profile_sharded.txt, r05_trace_sharded_world1.txt): ~30 us of host time per
// enqueue and two cross-queue waits per asynchronous collective = 54-64 us of GPU idle in a 0.38 ms iteration.  Here a collective is one
// ncclXxx call on the stream the kernels around it run on: no hand-over at all when issued in line, one event each way when the caller
// wants it beside other work (the caller's choice: the stream argument).
//
// RCCL is loaded at run time (dlopen of librccl.so.1 - in a PyTorch process that is the copy torch itself loaded): the library has no
// link-time dependency on it, and a single-GPU user never touches it.  The reference has no multi-GPU mapping at all; its only fan-out
// is independent trials (examples/run_loner.py:339-424).
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

#include "lnr_common.h"

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
#define LNR_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym))
        LNR_SYM(GetUniqueId, "ncclGetUniqueId");
        LNR_SYM(CommInitRank, "ncclCommInitRank");
        LNR_SYM(CommDestroy, "ncclCommDestroy");
        LNR_SYM(AllReduce, "ncclAllReduce");
        LNR_SYM(ReduceScatter, "ncclReduceScatter");
        LNR_SYM(AllGather, "ncclAllGather");
        LNR_SYM(Broadcast, "ncclBroadcast");
        LNR_SYM(GetErrorString, "ncclGetErrorString");
#undef LNR_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.ReduceScatter && api.AllGather && api.Broadcast &&
                 api.GetErrorString;
    });
    return api;
}

struct LnrComm {
    ncclComm_t comm;
    int rank, world;
};

int fail(const char* what, ncclResult_t r) {
    lnr_set_error("%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
    return LNR_ERR_LAUNCH;
}

bool dtype_of(int32_t dtype, ncclDataType_t* out) {
    switch (dtype) {
        case LNR_COMM_F32: *out = ncclFloat32; return true;
        case LNR_COMM_BF16: *out = ncclBfloat16; return true;
        case LNR_COMM_I64: *out = ncclInt64; return true;
        case LNR_COMM_I32: *out = ncclInt32; return true;
        case LNR_COMM_U8: *out = ncclUint8; return true;
        default: return false;
    }
}
bool op_of(int32_t op, ncclRedOp_t* out) {
    switch (op) {
        case LNR_COMM_SUM: *out = ncclSum; return true;
        case LNR_COMM_MIN: *out = ncclMin; return true;
        case LNR_COMM_MAX: *out = ncclMax; return true;
        default: return false;
    }
}
}  // namespace

extern "C" int lnr_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int lnr_comm_unique_id(void* id, size_t id_bytes) {
    LNR_REQUIRE(id != nullptr && id_bytes == LNR_COMM_ID_BYTES, "lnr_comm_unique_id: id must be LNR_COMM_ID_BYTES bytes");
    static_assert(LNR_COMM_ID_BYTES == sizeof(ncclUniqueId), "id size");
    LNR_REQUIRE(rccl().ok, "lnr_comm_unique_id: librccl.so.1 could not be loaded");
    ncclUniqueId u;
    const ncclResult_t r = rccl().GetUniqueId(&u);
    if (r != ncclSuccess) return fail("lnr_comm_unique_id", r);
    memcpy(id, &u, sizeof(u));
    return LNR_OK;
}

extern "C" int lnr_comm_init(const void* id, size_t id_bytes, int32_t rank, int32_t world, void** comm_out) {
    LNR_REQUIRE(id != nullptr && id_bytes == LNR_COMM_ID_BYTES && comm_out != nullptr, "lnr_comm_init: bad argument");
    LNR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "lnr_comm_init: rank %d of %d", rank, world);
    LNR_REQUIRE(rccl().ok, "lnr_comm_init: librccl.so.1 could not be loaded");
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    LnrComm* c = new LnrComm{nullptr, rank, world};
    const ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);        // (the current HIP device of the calling thread)
    if (r != ncclSuccess) { delete c; return fail("lnr_comm_init", r); }
    *comm_out = c;
    return LNR_OK;
}

extern "C" int lnr_comm_destroy(void* comm) {
    if (comm == nullptr) return LNR_OK;
    LnrComm* c = static_cast<LnrComm*>(comm);
    const ncclResult_t r = rccl().CommDestroy(c->comm);
    delete c;
    return r == ncclSuccess ? LNR_OK : fail("lnr_comm_destroy", r);
}

extern "C" int lnr_comm_all_reduce(void* comm, void* buf, size_t count, int32_t dtype, int32_t op, void* stream) {
    LNR_REQUIRE(comm != nullptr && (buf != nullptr || count == 0), "lnr_comm_all_reduce: null argument");
    ncclDataType_t dt; ncclRedOp_t ro;
    LNR_REQUIRE(dtype_of(dtype, &dt) && op_of(op, &ro), "lnr_comm_all_reduce: unknown dtype %d / op %d", dtype, op);
    if (count == 0) return LNR_OK;
    const ncclResult_t r = rccl().AllReduce(buf, buf, count, dt, ro, static_cast<LnrComm*>(comm)->comm, (hipStream_t)stream);
    return r == ncclSuccess ? LNR_OK : fail("lnr_comm_all_reduce", r);
}

extern "C" int lnr_comm_reduce_scatter(void* comm, const void* send, void* recv, size_t recv_count, int32_t dtype, void* stream) {
    LNR_REQUIRE(comm != nullptr && send != nullptr && recv != nullptr, "lnr_comm_reduce_scatter: null argument");
    ncclDataType_t dt;
    LNR_REQUIRE(dtype_of(dtype, &dt), "lnr_comm_reduce_scatter: unknown dtype %d", dtype);
    if (recv_count == 0) return LNR_OK;
    const ncclResult_t r = rccl().ReduceScatter(send, recv, recv_count, dt, ncclSum, static_cast<LnrComm*>(comm)->comm, (hipStream_t)stream);
    return r == ncclSuccess ? LNR_OK : fail("lnr_comm_reduce_scatter", r);
}

extern "C" int lnr_comm_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    LNR_REQUIRE(comm != nullptr && send != nullptr && recv != nullptr, "lnr_comm_all_gather: null argument");
    if (bytes_per_rank == 0) return LNR_OK;
    // (send may be the caller's own slot of recv: RCCL's in-place form)
    const ncclResult_t r = rccl().AllGather(send, recv, bytes_per_rank, ncclUint8, static_cast<LnrComm*>(comm)->comm, (hipStream_t)stream);
    return r == ncclSuccess ? LNR_OK : fail("lnr_comm_all_gather", r);
}

extern "C" int lnr_comm_broadcast(void* comm, void* buf, size_t bytes, int32_t root, void* stream) {
    LNR_REQUIRE(comm != nullptr && (buf != nullptr || bytes == 0), "lnr_comm_broadcast: null argument");
    if (bytes == 0) return LNR_OK;
    const ncclResult_t r = rccl().Broadcast(buf, buf, bytes, ncclUint8, root, static_cast<LnrComm*>(comm)->comm, (hipStream_t)stream);
    return r == ncclSuccess ? LNR_OK : fail("lnr_comm_broadcast", r);
}
