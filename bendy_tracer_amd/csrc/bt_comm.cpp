// bt_comm.cpp -- the multi-GPU exchange step behind the C ABI (include/bendy_hip.h, "Multi-GPU"): one RCCL
// all-gather of the ranks' tile shards over xGMI, then the un-permute kernel.  The reference has no counterpart (its
// only parallelism is rayon tiles inside one process, tracer/mod.rs:190-197); this is the collective BASELINE.json's
// north_star names, callable by a host that has no RCCL binding of its own (one process per GPU).
//
// RCCL is bound at run time (dlopen "librccl.so.1"): a process that never calls bt_comm_* does not need the library,
// and a process that already holds a copy (PyTorch ships its own librccl.so.1, same SONAME) shares that one instead of
// loading a second runtime next to it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/bendy_hip.h"

extern "C" int bt_set_error_internal(int code, const char *msg);      // bt_api.cpp

namespace {

// the slice of rccl.h this file needs (stable C ABI of NCCL 2.x / RCCL)
struct NcclUniqueId { char internal[BT_COMM_ID_BYTES]; };
typedef void *NcclComm;
enum { kNcclSuccess = 0, kNcclFloat = 7 };
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = std::string("cannot load librccl.so.1: ") + dlerror();
            return;
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) r.error = "librccl.so.1 lacks the NCCL 2 entry points";
    });
    return r;
}

int rccl_error(const char *what, int code) {
    Rccl &r = rccl();
    std::string msg = std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(code) : "RCCL error") + " (" +
                      std::to_string(code) + ")";
    return bt_set_error_internal(BT_ERR_DEVICE, msg.c_str());
}

} // namespace

struct bt_comm {
    NcclComm comm = nullptr;
    uint32_t rank = 0, world = 1;
    int device = -1;
};

extern "C" {

int bt_comm_unique_id(void *id_out, size_t cap) {
    if (!id_out || cap < BT_COMM_ID_BYTES) return bt_set_error_internal(BT_ERR_INVALID_ARG, "unique id buffer must hold BT_COMM_ID_BYTES");
    Rccl &r = rccl();
    if (!r.error.empty()) return bt_set_error_internal(BT_ERR_DEVICE, r.error.c_str());
    NcclUniqueId id;
    const int e = r.GetUniqueId(&id);
    if (e != kNcclSuccess) return rccl_error("ncclGetUniqueId", e);
    std::memcpy(id_out, id.internal, BT_COMM_ID_BYTES);
    return BT_COMM_ID_BYTES;
}

bt_comm *bt_comm_init(uint32_t rank, uint32_t world, const void *unique_id, size_t id_bytes) {
    if (world == 0 || rank >= world || !unique_id || id_bytes != BT_COMM_ID_BYTES) {
        bt_set_error_internal(BT_ERR_INVALID_ARG, "bt_comm_init: rank < world and a BT_COMM_ID_BYTES unique id are required");
        return nullptr;
    }
    Rccl &r = rccl();
    if (!r.error.empty()) {
        bt_set_error_internal(BT_ERR_DEVICE, r.error.c_str());
        return nullptr;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        bt_set_error_internal(BT_ERR_DEVICE, "bt_comm_init: no HIP device");
        return nullptr;
    }
    NcclUniqueId id;
    std::memcpy(id.internal, unique_id, BT_COMM_ID_BYTES);
    bt_comm *c = new bt_comm();
    c->rank = rank;
    c->world = world;
    c->device = dev;
    const int e = r.CommInitRank(&c->comm, (int)world, id, (int)rank);      // collective: every rank calls it
    if (e != kNcclSuccess) {
        rccl_error("ncclCommInitRank", e);
        delete c;
        return nullptr;
    }
    return c;
}

void bt_comm_free(bt_comm *comm) {
    if (!comm) return;
    if (comm->comm) (void)rccl().CommDestroy(comm->comm);
    delete comm;
}

int bt_comm_rank(const bt_comm *comm) { return comm ? (int)comm->rank : -1; }
int bt_comm_world(const bt_comm *comm) { return comm ? (int)comm->world : 0; }

int bt_allgather_shards_device(bt_comm *comm, const float *shard_device, float *gathered_device, uint32_t width,
                               uint32_t height, void *stream) {
    if (!comm || !shard_device || !gathered_device || width == 0 || height == 0)
        return bt_set_error_internal(BT_ERR_INVALID_ARG, "bt_allgather_shards_device: null / zero argument");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != comm->device)        // the communicator lives on the device it was created on
        return bt_set_error_internal(BT_ERR_INVALID_ARG, "bt_allgather_shards_device: the current device is not the communicator's (hipSetDevice first)");
    const size_t count = bt_shard_floats(width, height, comm->world);
    const int e = rccl().AllGather(shard_device, gathered_device, count, kNcclFloat, comm->comm, (hipStream_t)stream);
    if (e != kNcclSuccess) return rccl_error("ncclAllGather", e);
    return 0;
}

int bt_exchange_frame_device(bt_comm *comm, const float *shard_device, float *gathered_device, float *rgba_device,
                             uint32_t width, uint32_t height, void *stream) {
    int rc = bt_allgather_shards_device(comm, shard_device, gathered_device, width, height, stream);
    if (rc) return rc;
    return bt_unshard_device(gathered_device, rgba_device, width, height, comm->world, stream);
}

} // extern "C"
