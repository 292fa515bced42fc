// comm.cpp -- RCCL ghost-plane exchange (see comm.h).
#include "comm.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace wv {
namespace {

// Minimal RCCL surface, resolved with dlsym (types per rccl.h; ncclUniqueId is 128 opaque bytes).
struct UniqueId {
    char internal[128];
};
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*GroupFn)(void);
typedef int (*SendFn)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*RecvFn)(void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*ErrStrFn)(int);

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GroupFn group_start = nullptr, group_end = nullptr;
    SendFn send = nullptr;
    RecvFn recv = nullptr;
    ErrStrFn err_str = nullptr;
    std::string load_error;
};

constexpr int kNcclInt8 = 0;  // ncclInt8 / ncclChar

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.load_error = std::string("cannot load librccl: ") + dlerror();
            return;
        }
        r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
        r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
        r.group_start = (GroupFn)dlsym(r.handle, "ncclGroupStart");
        r.group_end = (GroupFn)dlsym(r.handle, "ncclGroupEnd");
        r.send = (SendFn)dlsym(r.handle, "ncclSend");
        r.recv = (RecvFn)dlsym(r.handle, "ncclRecv");
        r.err_str = (ErrStrFn)dlsym(r.handle, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.group_start || !r.group_end ||
            !r.send || !r.recv || !r.err_str)
            r.load_error = "librccl is missing an expected symbol";
    });
    return r;
}

bool nccl_ok(int rc, const char* what, std::string* err) {
    if (rc == 0) return true;
    *err = std::string(what) + ": " + rccl().err_str(rc);
    return false;
}

bool hip_ok(hipError_t rc, const char* what, std::string* err) {
    if (rc == hipSuccess) return true;
    *err = std::string(what) + ": " + hipGetErrorString(rc);
    return false;
}

}  // namespace

bool SlabComm::unique_id(void* bytes128, std::string* err) {
    Rccl& r = rccl();
    if (!r.load_error.empty()) {
        *err = r.load_error;
        return false;
    }
    UniqueId id;
    if (!nccl_ok(r.get_unique_id(&id), "ncclGetUniqueId", err)) return false;
    std::memcpy(bytes128, &id, sizeof(id));
    return true;
}

bool SlabComm::init(const void* id_bytes128, int rank, int nranks, int device, hipStream_t comm_stream,
                    bool has_lo, bool has_hi, std::string* err) {
    Rccl& r = rccl();
    if (!r.load_error.empty()) {
        *err = r.load_error;
        return false;
    }
    if (nranks < 1 || rank < 0 || rank >= nranks) {
        *err = "rank outside [0, nranks)";
        return false;
    }
    // nranks == 1 with both ghosts = loopback: the slab is its own neighbour on both sides
    // (periodic in z).  Exists so that the RCCL send/recv + stream/event choreography can be
    // exercised on a single GPU; the multi-rank chain is open-ended.
    loopback_ = (nranks == 1 && has_lo && has_hi);
    if (!loopback_ && (has_lo != (rank > 0) || has_hi != (rank + 1 < nranks))) {
        *err = "ghost_lo/ghost_hi of the engine do not match its position in the slab chain";
        return false;
    }
    if (!hip_ok(hipSetDevice(device), "hipSetDevice", err)) return false;
    UniqueId id;
    std::memcpy(&id, id_bytes128, sizeof(id));
    if (!nccl_ok(r.comm_init_rank(&comm_, nranks, id, rank), "ncclCommInitRank", err)) return false;
    rank_ = rank;
    nranks_ = nranks;
    has_lo_ = has_lo;
    has_hi_ = has_hi;
    stream_ = comm_stream;
    if (!hip_ok(hipEventCreateWithFlags(&faces_ready_, hipEventDisableTiming), "hipEventCreate", err)) return false;
    if (!hip_ok(hipEventCreateWithFlags(&ghosts_ready_, hipEventDisableTiming), "hipEventCreate", err)) return false;
    return true;
}

SlabComm::~SlabComm() {
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (comm_) (void)rccl().comm_destroy(comm_);
    if (faces_ready_) (void)hipEventDestroy(faces_ready_);
    if (ghosts_ready_) (void)hipEventDestroy(ghosts_ready_);
}

bool SlabComm::wait_ghosts(hipStream_t compute, std::string* err) {
    if (!pending_) return true;
    return hip_ok(hipStreamWaitEvent(compute, ghosts_ready_, 0), "hipStreamWaitEvent", err);
}

bool SlabComm::exchange_faces(hipStream_t compute, hipEvent_t also, void* field, size_t elem_size, int nx, int ny,
                              int nz, std::string* err) {
    Rccl& r = rccl();
    const size_t plane_bytes = (size_t)nx * ny * elem_size;
    char* base = static_cast<char*>(field);
    if (!hip_ok(hipEventRecord(faces_ready_, compute), "hipEventRecord", err)) return false;
    if (!hip_ok(hipStreamWaitEvent(stream_, faces_ready_, 0), "hipStreamWaitEvent", err)) return false;
    if (also && !hip_ok(hipStreamWaitEvent(stream_, also, 0), "hipStreamWaitEvent", err)) return false;
    if (loopback_) {
        // sends and receives to the same peer pair up in issue order
        if (!nccl_ok(r.group_start(), "ncclGroupStart", err)) return false;
        if (!nccl_ok(r.send(base + plane_bytes, plane_bytes, kNcclInt8, 0, comm_, stream_), "ncclSend", err)) return false;
        if (!nccl_ok(r.recv(base + (size_t)(nz - 1) * plane_bytes, plane_bytes, kNcclInt8, 0, comm_, stream_), "ncclRecv", err))
            return false;
        if (!nccl_ok(r.send(base + (size_t)(nz - 2) * plane_bytes, plane_bytes, kNcclInt8, 0, comm_, stream_), "ncclSend", err))
            return false;
        if (!nccl_ok(r.recv(base, plane_bytes, kNcclInt8, 0, comm_, stream_), "ncclRecv", err)) return false;
        if (!nccl_ok(r.group_end(), "ncclGroupEnd", err)) return false;
    } else if (has_lo_ || has_hi_) {
        if (!nccl_ok(r.group_start(), "ncclGroupStart", err)) return false;
        if (has_lo_) {
            // my first owned plane (z=1) -> lower neighbour's top ghost; its top owned plane -> my z=0
            if (!nccl_ok(r.send(base + plane_bytes, plane_bytes, kNcclInt8, rank_ - 1, comm_, stream_), "ncclSend", err))
                return false;
            if (!nccl_ok(r.recv(base, plane_bytes, kNcclInt8, rank_ - 1, comm_, stream_), "ncclRecv", err)) return false;
        }
        if (has_hi_) {
            if (!nccl_ok(r.send(base + (size_t)(nz - 2) * plane_bytes, plane_bytes, kNcclInt8, rank_ + 1, comm_, stream_),
                         "ncclSend", err))
                return false;
            if (!nccl_ok(r.recv(base + (size_t)(nz - 1) * plane_bytes, plane_bytes, kNcclInt8, rank_ + 1, comm_, stream_),
                         "ncclRecv", err))
                return false;
        }
        if (!nccl_ok(r.group_end(), "ncclGroupEnd", err)) return false;
    }
    if (!hip_ok(hipEventRecord(ghosts_ready_, stream_), "hipEventRecord", err)) return false;
    pending_ = true;
    return true;
}

}  // namespace wv
