// comm.cpp -- RCCL ghost-plane exchange (see comm.h).
#include "comm.h"

#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <cstring>
#include <initializer_list>
#include <mutex>
#include <thread>

namespace wv {
namespace {

// Minimal RCCL surface, resolved with dlsym (types per rccl.h; ncclUniqueId is 128 opaque bytes).
struct UniqueId {
    char internal[128];
};
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*CommAbortFn)(void*);
typedef int (*GroupFn)(void);
typedef int (*SendFn)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*RecvFn)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*ErrStrFn)(int);

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    CommAbortFn comm_abort = nullptr;  // optional: the watchdog's way to end kernels in flight
    GroupFn group_start = nullptr, group_end = nullptr;
    SendFn send = nullptr;
    RecvFn recv = nullptr;
    AllReduceFn all_reduce = nullptr;
    ErrStrFn err_str = nullptr;
    std::string load_error;
};

constexpr int kNcclInt8 = 0;    // ncclInt8 / ncclChar
constexpr int kNcclUint64 = 5;  // ncclUint64
constexpr int kNcclSum = 0;     // ncclSum
constexpr int kNcclMin = 3;     // ncclMin

// local transport: whose device-filling launch was enqueued last on each device (SlabComm::bulk_begin / bulk_end)
struct DeviceTurn {
    const void* owner = nullptr;
    hipEvent_t done = nullptr;
};
constexpr int kMaxDevices = 64;
std::mutex g_turn_mutex;
DeviceTurn g_turn[kMaxDevices];

std::string& library_override() {
    static std::string path;
    return path;
}
// the override and "has the loader run" are read and written under one lock: wv_comm_use_library racing with another
// thread's first communicator call either lands before the load or is refused
std::mutex g_library_mutex;
bool g_library_loaded = false;

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::lock_guard<std::mutex> lock(g_library_mutex);
        g_library_loaded = true;
        if (!library_override().empty()) {  // wv_comm_use_library: this file and nothing else
            r.handle = dlopen(library_override().c_str(), RTLD_NOW | RTLD_LOCAL);
        } else {
            const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
            for (const char* n : names) {
                r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (r.handle) break;
            }
        }
        if (!r.handle) {
            r.load_error = std::string("cannot load librccl: ") + dlerror();
            return;
        }
        r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
        r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
        r.comm_abort = (CommAbortFn)dlsym(r.handle, "ncclCommAbort");
        r.group_start = (GroupFn)dlsym(r.handle, "ncclGroupStart");
        r.group_end = (GroupFn)dlsym(r.handle, "ncclGroupEnd");
        r.send = (SendFn)dlsym(r.handle, "ncclSend");
        r.recv = (RecvFn)dlsym(r.handle, "ncclRecv");
        r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
        r.err_str = (ErrStrFn)dlsym(r.handle, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.group_start || !r.group_end ||
            !r.send || !r.recv || !r.all_reduce || !r.err_str)
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

// RCCL has no bitwise-OR reduction.  The error_code word has 5 bits (cl/structs.h:8-15): bit b goes
// to a 12-bit counter at bit 12*b of a 64-bit word, the words are summed over the ranks (no carry
// into the next counter below 4096 ranks), and a non-zero counter is a set bit again.
constexpr int kFlagBits = 5, kFlagField = 12;

__global__ void flag_spread_kernel(const int* flags, uint64_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t v = 0;
    for (int b = 0; b < kFlagBits; ++b)
        if ((flags[i] >> b) & 1) v |= 1ull << (kFlagField * b);
    out[i] = v;
}

// ---- IPC transport: counters in the neighbours' mailboxes ---------------------------------------------------------------------
// (one thread each; on the stream behind the copies whose landing they announce / in front of the work that needs the planes)
struct IpcFlags {
    uint64_t* flag[2];
    uint64_t value[2];
    int side[2];  // 0: the lower neighbour's, 1: the upper one's (what a time-out names)
    int n;
};
// *status != 0: a wait of this rank has timed out -- its ghost planes are stale and so is everything computed from them since.  The
// rank says nothing more to its neighbours (they time out in their turn and end with WV_E_COMM too, instead of stepping on with planes
// that mean nothing), and its own later waits return at once (one time-out per batch, not one per exchange).
__global__ void ipc_post_kernel(IpcFlags f, const int* status) {
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
    __threadfence_system();  // the copies before this kernel in stream order are visible to whoever sees the counter
    for (int i = 0; i < f.n; ++i) __hip_atomic_store(f.flag[i], f.value[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Waits until every counter has reached its value; gives up after `ticks` of the 100 MHz wall clock and says so in *status
// (`code` << 4 * side): the batch then ends with WV_E_COMM instead of a stream that never drains.
__global__ void ipc_wait_kernel(IpcFlags f, long long ticks, int* status, int code) {
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
    const long long t0 = wall_clock64();
    for (int i = 0; i < f.n; ++i) {
        while (__hip_atomic_load(f.flag[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < f.value[i]) {
            if (ticks > 0 && wall_clock64() - t0 > ticks) {
                __hip_atomic_fetch_or(status, code << (4 * f.side[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
}

// what a rank tells its neighbours about itself at set-up (travels through the communicator as bytes)
struct IpcBlob {
    int64_t pid;
    uint64_t raw_field[4], raw_mailbox;  // addresses in the owner's process: used as they are by a neighbour in the SAME process
    hipIpcMemHandle_t field[4], mailbox;
    int32_t has_field[4];
    int32_t nz;
    uint32_t device_uuid_lo;
    uint64_t plane_bytes;
};

__global__ void flag_gather_kernel(const uint64_t* in, int* flags, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int f = 0;
    for (int b = 0; b < kFlagBits; ++b)
        if ((in[i] >> (kFlagField * b)) & ((1ull << kFlagField) - 1)) f |= 1 << b;
    flags[i] = f;
}

}  // namespace

bool SlabComm::use_library(const char* path, std::string* err) {
    std::lock_guard<std::mutex> lock(g_library_mutex);
    if (g_library_loaded) {
        *err = "the collective library is already loaded: wv_comm_use_library must come before the first communicator call";
        return false;
    }
    library_override() = path ? path : "";
    return true;
}

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

bool SlabComm::init_local(int rank, int nranks, int device, hipStream_t comm_stream, bool has_lo, bool has_hi,
                          std::string* err) {
    if (nranks < 1 || rank < 0 || rank >= nranks) {
        *err = "rank outside [0, nranks)";
        return false;
    }
    if (has_lo != (rank > 0) || has_hi != (rank + 1 < nranks)) {
        *err = "ghost_lo/ghost_hi of the engine do not match its position in the slab chain";
        return false;
    }
    if (device < 0 || device >= kMaxDevices) {
        *err = "device ordinal beyond the in-process transport's turn table";
        return false;
    }
    if (!hip_ok(hipSetDevice(device), "hipSetDevice", err)) return false;
    local_ = true;
    device_ = device;
    rank_ = rank;
    nranks_ = nranks;
    has_lo_ = has_lo;
    has_hi_ = has_hi;
    stream_ = comm_stream;
    for (hipEvent_t* e : {&faces_ready_, &ghosts_ready_, &pushed_lo_[0], &pushed_lo_[1], &pushed_lo_[2], &pushed_lo_[3], &pushed_hi_[0],
                          &pushed_hi_[1], &pushed_hi_[2], &pushed_hi_[3], &step_done_[0], &step_done_[1], &bulk_done_})
        if (!hip_ok(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate", err)) return false;
    return true;
}

void SlabComm::link_local(SlabComm* lo, SlabComm* hi) {
    lo_ = lo;
    hi_ = hi;
    for (SlabComm* peer : {lo, hi}) {
        if (!peer || peer->device_ == device_) continue;
        int can = 0, before = 0;
        (void)hipGetDevice(&before);
        if (hipDeviceCanAccessPeer(&can, device_, peer->device_) == hipSuccess && can && hipSetDevice(device_) == hipSuccess) {
            (void)hipDeviceEnablePeerAccess(peer->device_, 0);  // "already enabled" is fine
            (void)hipGetLastError();
        }
        (void)hipSetDevice(before);
    }
}

// one face plane into a neighbour's ghost plane, on this slab's halo stream
static hipError_t push_plane(void* dst, int dst_device, const void* src, int src_device, size_t bytes, hipStream_t stream) {
    if (dst_device == src_device) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    return hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, stream);
}

void SlabComm::set_fields(void* const* fields, int n_fields, size_t plane_bytes, int nz) {
    n_fields_ = n_fields < 4 ? n_fields : 4;
    for (int i = 0; i < n_fields_; ++i) fields_[i] = fields[i];
    plane_bytes_ = plane_bytes;
    nz_ = nz;
}

SlabComm::~SlabComm() {
    if (dead_) return;  // (its streams may never drain, and the watchdog has aborted the communicator: everything is left to the process's end)
    if (stream_) (void)hipStreamSynchronize(stream_);
    // local transport: a neighbour's push into this slab's ghost plane runs on the NEIGHBOUR's halo stream
    for (SlabComm* peer : {lo_, hi_})
        if (peer && peer->stream_) {
            int before = 0;
            (void)hipGetDevice(&before);
            if (peer->device_ != before) (void)hipSetDevice(peer->device_);
            (void)hipStreamSynchronize(peer->stream_);
            if (peer->device_ != before) (void)hipSetDevice(before);
        }
    if (comm_) (void)rccl().comm_destroy(comm_);
    for (int side = 0; side < 2; ++side)
        for (void* p : peer_opened_[side])
            if (p) (void)hipIpcCloseMemHandle(p);
    if (mailbox_) (void)hipFree(mailbox_);
    if (ipc_status_) (void)hipHostFree(ipc_status_);
    if (own_push_) (void)hipEventDestroy(own_push_);
    if (local_) {
        std::lock_guard<std::mutex> lock(g_turn_mutex);
        DeviceTurn& t = g_turn[device_];
        if (t.owner == this) t = DeviceTurn{};
    }
    for (hipEvent_t e : {halo_joined_, bulk_done_, faces_ready_, ghosts_ready_, pushed_lo_[0], pushed_lo_[1], pushed_lo_[2], pushed_lo_[3], pushed_hi_[0], pushed_hi_[1],
                         pushed_hi_[2], pushed_hi_[3], step_done_[0], step_done_[1], reduce_in_, reduce_out_, sync_ev_})
        if (e) (void)hipEventDestroy(e);
    if (spread_) (void)hipFree(spread_);
    if (host_words_) (void)hipHostFree(host_words_);
    // a local chain is torn down as a whole (wv_comm_destroy on every engine); unlink anyway
    if (lo_ && lo_->hi_ == this) lo_->hi_ = nullptr;
    if (hi_ && hi_->lo_ == this) hi_->lo_ = nullptr;
}

bool SlabComm::wait_ghosts(hipStream_t compute, int field, std::string* err) {
    if (dead_) {
        *err = "the communicator was aborted after a time-out (an earlier call says where)";
        return false;
    }
    if (local_) {
        // my ghost planes of THIS buffer are written by the neighbours' pushes into it -- and by no later ones: a
        // neighbour that is already a step (or half a pass) ahead in host order has pushed into another buffer since
        if (field < 0 || field >= 4) {
            *err = "wait_ghosts: no such field buffer";
            return false;
        }
        if (lo_ && lo_->pushed_hi_set_[field] &&
            !hip_ok(hipStreamWaitEvent(compute, lo_->pushed_hi_[field], 0), "hipStreamWaitEvent", err))
            return false;
        if (hi_ && hi_->pushed_lo_set_[field] &&
            !hip_ok(hipStreamWaitEvent(compute, hi_->pushed_lo_[field], 0), "hipStreamWaitEvent", err))
            return false;
        // ... and MY pushes have read my face planes: what follows on the compute stream may write into them -- a source that
        // lies on a slab face gets its sample added to this very field, and a push that read the plane after that would hand
        // the neighbour (which adds the sample to its ghost copy itself) the sample twice.  (Found by tools/extended_fuzz.py,
        // about one chain in 1 500; the RCCL transport's "ghosts ready" event already covers this rank's sends.)
        // (the halo stream runs them in order: the latest one stands for all)
        if (last_own_push_ && !hip_ok(hipStreamWaitEvent(compute, last_own_push_, 0), "hipStreamWaitEvent", err)) return false;
        return true;
    }
    if (ipc_) {
        // the neighbours' planes of THIS buffer: each neighbour takes the same steps as this rank, so its latest exchange of the
        // buffer has the number of this rank's own latest one
        if (field < 0 || field >= 4) {
            *err = "wait_ghosts: no such field buffer";
            return false;
        }
        if (pushes_[field]) {
            const uint64_t* flags[2];
            uint64_t values[2];
            int n = 0;
            for (int side = 0; side < 2; ++side)
                if (side == 0 ? has_lo_ : has_hi_) {
                    flags[n] = mailbox_ + side * 4 + field;
                    values[n++] = pushes_[field];
                }
            if (n && !ipc_wait(compute, n, flags, values, 1, err)) return false;
        }
        // ... and this rank's own copies have read its face planes (see the local transport above)
        if (own_push_set_ && compute != stream_ && !hip_ok(hipStreamWaitEvent(compute, own_push_, 0), "hipStreamWaitEvent", err)) return false;
        return true;
    }
    if (!pending_) return true;
    return hip_ok(hipStreamWaitEvent(compute, ghosts_ready_, 0), "hipStreamWaitEvent", err);
}

bool SlabComm::ipc_wait(hipStream_t stream, int n, const uint64_t* const* flags, const uint64_t* values, int code, std::string* err) {
    IpcFlags f{};
    f.n = n;
    for (int i = 0; i < n; ++i) {
        f.flag[i] = const_cast<uint64_t*>(flags[i]);
        f.value[i] = values[i];
        // (callers list the lower neighbour's counter first when there is one: with one neighbour only, it is whichever that is)
        f.side[i] = (n == 2) ? i : (has_lo_ ? 0 : 1);
    }
    const long long ticks = timeout_s_ > 0 ? (long long)(timeout_s_ * 1e8) : 0;
    hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(1), 0, stream, f, ticks, ipc_status_, code);
    return hip_ok(hipGetLastError(), "ipc_wait_kernel", err);
}

bool SlabComm::ipc_post(hipStream_t stream, int n, uint64_t* const* flags, const uint64_t* values, std::string* err) {
    IpcFlags f{};
    f.n = n;
    for (int i = 0; i < n; ++i) {
        f.flag[i] = flags[i];
        f.value[i] = values[i];
    }
    hipLaunchKernelGGL(ipc_post_kernel, dim3(1), dim3(1), 0, stream, f, ipc_status_);
    return hip_ok(hipGetLastError(), "ipc_post_kernel", err);
}

// Handles of this rank's fields and mailbox to its neighbours, theirs back, through the communicator (grouped send / receive of
// bytes, like the planes of the RCCL transport); then the neighbours' memory is mapped (hipIpcOpenMemHandle) -- or taken as it
// is where the neighbour lives in this very process (ranks as threads: the tests' stand-in; a rank that is its own neighbour).
bool SlabComm::init_ipc(std::string* err) {
    if (local_ || !comm_) {
        *err = "the IPC transport rides on an RCCL communicator (wv_comm_init), not on the in-process one";
        return false;
    }
    Rccl& r = rccl();
    // the mailbox: uncached, so that a counter written from another process / GPU is what the next load sees
    if (hipExtMallocWithFlags((void**)&mailbox_, kMailboxWords * sizeof(uint64_t), hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (!hip_ok(hipExtMallocWithFlags((void**)&mailbox_, kMailboxWords * sizeof(uint64_t), hipDeviceMallocFinegrained),
                    "hipExtMallocWithFlags (mailbox)", err))
            return false;
    }
    if (!hip_ok(hipMemset(mailbox_, 0, kMailboxWords * sizeof(uint64_t)), "hipMemset", err)) return false;
    if (!hip_ok(hipHostMalloc((void**)&ipc_status_, sizeof(int), hipHostMallocDefault), "hipHostMalloc", err)) return false;
    *ipc_status_ = 0;
    if (!hip_ok(hipEventCreateWithFlags(&own_push_, hipEventDisableTiming), "hipEventCreate", err)) return false;
    IpcBlob mine{};
    mine.pid = (int64_t)getpid();
    mine.raw_mailbox = (uint64_t)(uintptr_t)mailbox_;
    mine.nz = nz_;
    mine.plane_bytes = plane_bytes_;
    bool need_handles = !loopback_;
    for (int i = 0; i < 4; ++i) {
        mine.has_field[i] = i < n_fields_ && fields_[i] != nullptr;
        mine.raw_field[i] = (uint64_t)(uintptr_t)fields_[i];
        if (mine.has_field[i] && need_handles && !hip_ok(hipIpcGetMemHandle(&mine.field[i], fields_[i]), "hipIpcGetMemHandle (field)", err))
            return false;
    }
    if (need_handles && !hip_ok(hipIpcGetMemHandle(&mine.mailbox, mailbox_), "hipIpcGetMemHandle (mailbox)", err)) return false;
    IpcBlob theirs[2] = {};
    if (loopback_) {
        theirs[0] = theirs[1] = mine;
    } else {
        IpcBlob* dev = nullptr;  // [0] mine, [1] from rank - 1, [2] from rank + 1
        if (!hip_ok(hipMalloc((void**)&dev, 3 * sizeof(IpcBlob)), "hipMalloc", err)) return false;
        bool ok = hip_ok(hipMemcpy(dev, &mine, sizeof(IpcBlob), hipMemcpyHostToDevice), "hipMemcpy", err);
        ok = ok && nccl_ok(r.group_start(), "ncclGroupStart", err);
        if (ok && has_lo_)
            ok = nccl_ok(r.send(dev, sizeof(IpcBlob), kNcclInt8, rank_ - 1, comm_, stream_), "ncclSend", err) &&
                 nccl_ok(r.recv(dev + 1, sizeof(IpcBlob), kNcclInt8, rank_ - 1, comm_, stream_), "ncclRecv", err);
        if (ok && has_hi_)
            ok = nccl_ok(r.send(dev, sizeof(IpcBlob), kNcclInt8, rank_ + 1, comm_, stream_), "ncclSend", err) &&
                 nccl_ok(r.recv(dev + 2, sizeof(IpcBlob), kNcclInt8, rank_ + 1, comm_, stream_), "ncclRecv", err);
        ok = ok && nccl_ok(r.group_end(), "ncclGroupEnd", err);
        ok = ok && sync(stream_, "the exchange of IPC handles with the neighbouring ranks", err);
        ok = ok && hip_ok(hipMemcpy(theirs, dev + 1, 2 * sizeof(IpcBlob), hipMemcpyDeviceToHost), "hipMemcpy", err);
        (void)hipFree(dev);
        if (!ok) return false;
    }
    for (int side = 0; side < 2; ++side) {
        if (!(side == 0 ? has_lo_ : has_hi_)) continue;
        const IpcBlob& b = theirs[side];
        if (b.plane_bytes != plane_bytes_) {
            *err = "neighbouring slabs disagree about the plane size";
            return false;
        }
        peer_nz_[side] = b.nz;
        const bool same_process = b.pid == mine.pid;
        for (int i = 0; i < 4; ++i) {
            if (!b.has_field[i]) continue;
            if (same_process) {
                peer_field_[side][i] = (char*)(uintptr_t)b.raw_field[i];
            } else {
                void* p = nullptr;
                if (!hip_ok(hipIpcOpenMemHandle(&p, b.field[i], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle (a neighbour's field)", err)) return false;
                peer_opened_[side][i] = p;
                peer_field_[side][i] = static_cast<char*>(p);
            }
        }
        if (same_process) {
            peer_mailbox_[side] = (uint64_t*)(uintptr_t)b.raw_mailbox;
        } else {
            void* p = nullptr;
            if (!hip_ok(hipIpcOpenMemHandle(&p, b.mailbox, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle (a neighbour's mailbox)", err)) return false;
            peer_opened_[side][4] = p;
            peer_mailbox_[side] = static_cast<uint64_t*>(p);
        }
    }
    ipc_ = true;
    return true;
}

bool SlabComm::step_done(hipStream_t compute, std::string* err) {
    if (ipc_) {  // tell the neighbours: their next copy into my ghost planes may come
        ++steps_done_;
        uint64_t* flags[2];
        uint64_t values[2];
        int n = 0;
        for (int side = 0; side < 2; ++side)
            if (side == 0 ? has_lo_ : has_hi_) {
                flags[n] = peer_mailbox_[side] + 8 + (1 - side);  // (I am that neighbour's neighbour on its other side)
                values[n++] = steps_done_;
            }
        return n == 0 || ipc_post(compute, n, flags, values, err);
    }
    if (!local_) return true;  // RCCL: the matching ncclRecv is issued by this rank itself, in stream order
    const int which = (int)(steps_done_ & 1u);
    ++steps_done_;
    return hip_ok(hipEventRecord(step_done_[which], compute), "hipEventRecord", err);
}

bool SlabComm::bulk_begin(hipStream_t compute, std::string* err) {
    if (!local_ || nranks_ < 2) return true;
    std::lock_guard<std::mutex> lock(g_turn_mutex);
    const DeviceTurn& t = g_turn[device_];
    if (t.owner && t.owner != this) return hip_ok(hipStreamWaitEvent(compute, t.done, 0), "hipStreamWaitEvent", err);
    return true;
}

bool SlabComm::bulk_end(hipStream_t compute, std::string* err) {
    if (!local_ || nranks_ < 2) return true;
    if (!hip_ok(hipEventRecord(bulk_done_, compute), "hipEventRecord", err)) return false;
    std::lock_guard<std::mutex> lock(g_turn_mutex);
    DeviceTurn& t = g_turn[device_];
    t.owner = this;
    t.done = bulk_done_;
    return true;
}

bool SlabComm::join_halo(hipStream_t compute, std::string* err) {
    if (!halo_joined_ && !hip_ok(hipEventCreateWithFlags(&halo_joined_, hipEventDisableTiming), "hipEventCreate", err)) return false;
    if (!hip_ok(hipEventRecord(halo_joined_, stream_), "hipEventRecord", err)) return false;
    return hip_ok(hipStreamWaitEvent(compute, halo_joined_, 0), "hipStreamWaitEvent", err);
}

bool SlabComm::exchange_faces(hipStream_t compute, int field, std::string* err, bool on_halo_stream) {
    if (dead_) {
        *err = "the communicator was aborted after a time-out (an earlier call says where)";
        return false;
    }
    if (field < 0 || field >= n_fields_ || !fields_[field] || nz_ < 3) {
        *err = "exchange_faces: no such field buffer";
        return false;
    }
    const size_t plane_bytes = plane_bytes_;
    const int nz = nz_;
    char* base = static_cast<char*>(fields_[field]);
    if (!on_halo_stream) {
        if (!hip_ok(hipEventRecord(faces_ready_, compute), "hipEventRecord", err)) return false;
        if (!hip_ok(hipStreamWaitEvent(stream_, faces_ready_, 0), "hipStreamWaitEvent", err)) return false;
    }
    ++exchanges_;
    planes_sent_ += (loopback_ ? 2u : 0u) + (!loopback_ && has_lo_ ? 1u : 0u) + (!loopback_ && has_hi_ ? 1u : 0u);
    if (local_) {
        // push my face planes into the neighbours' ghost planes of the same buffer.  The neighbour
        // may still be reading that ghost plane (it was part of its `current` one step ago): wait
        // for the end of the step (or two-step pass) BEFORE the one being enqueued -- the event of
        // that parity, not the neighbour's latest: the lower neighbour has this step enqueued already,
        // and waiting for its end would put the chain's slabs one behind the other.  Lockstep driving
        // (wv_run_group) guarantees that every event waited for here has been recorded in host order
        // and is not recorded again before this slab's step is enqueued.
        const bool earlier = steps_done_ > 0;
        const int before = (int)((steps_done_ + 1u) & 1u);
        if (has_lo_ && lo_) {
            if (earlier && lo_->steps_done_ >= steps_done_ &&
                !hip_ok(hipStreamWaitEvent(stream_, lo_->step_done_[before], 0), "hipStreamWaitEvent", err))
                return false;
            if (field >= lo_->n_fields_ || !lo_->fields_[field]) {
                *err = "exchange_faces: the lower neighbour has no such field buffer (slabs of a chain must take the same steps)";
                return false;
            }
            char* dst = static_cast<char*>(lo_->fields_[field]) + (size_t)(lo_->nz_ - 1) * lo_->plane_bytes_;
            if (lo_->plane_bytes_ != plane_bytes) {
                *err = "neighbouring slabs disagree about the plane size";
                return false;
            }
            if (!hip_ok(push_plane(dst, lo_->device_, base + plane_bytes, device_, plane_bytes, stream_), "hipMemcpyAsync", err))
                return false;
            if (!hip_ok(hipEventRecord(pushed_lo_[field], stream_), "hipEventRecord", err)) return false;
            pushed_lo_set_[field] = true;
            last_own_push_ = pushed_lo_[field];
        }
        if (has_hi_ && hi_) {
            if (earlier && hi_->steps_done_ >= steps_done_ &&
                !hip_ok(hipStreamWaitEvent(stream_, hi_->step_done_[before], 0), "hipStreamWaitEvent", err))
                return false;
            if (hi_->plane_bytes_ != plane_bytes) {
                *err = "neighbouring slabs disagree about the plane size";
                return false;
            }
            if (field >= hi_->n_fields_ || !hi_->fields_[field]) {
                *err = "exchange_faces: the upper neighbour has no such field buffer (slabs of a chain must take the same steps)";
                return false;
            }
            char* dst = static_cast<char*>(hi_->fields_[field]);
            if (!hip_ok(push_plane(dst, hi_->device_, base + (size_t)(nz - 2) * plane_bytes, device_, plane_bytes, stream_),
                        "hipMemcpyAsync", err))
                return false;
            if (!hip_ok(hipEventRecord(pushed_hi_[field], stream_), "hipEventRecord", err)) return false;
            pushed_hi_set_[field] = true;
            last_own_push_ = pushed_hi_[field];
        }
        return true;
    }
    if (ipc_) {
        if (field >= 4 || (has_lo_ && !peer_field_[0][field]) || (has_hi_ && !peer_field_[1][field])) {
            *err = "exchange_faces: a neighbour has not shared this field buffer (slabs of a chain must take the same steps)";
            return false;
        }
        const uint64_t k = ++pushes_[field];
        // the neighbours have finished the step (or pass) in which they read the ghost planes these copies overwrite
        if (steps_done_ > 0) {
            const uint64_t* flags[2];
            uint64_t values[2];
            int n = 0;
            for (int side = 0; side < 2; ++side)
                if (side == 0 ? has_lo_ : has_hi_) {
                    flags[n] = mailbox_ + 8 + side;
                    values[n++] = steps_done_;
                }
            if (n && !ipc_wait(stream_, n, flags, values, 2, err)) return false;
        }
        uint64_t* posts[2];
        uint64_t values[2];
        int n = 0;
        if (has_lo_) {  // my first owned plane -> the lower neighbour's top ghost plane
            char* dst = peer_field_[0][field] + (size_t)(peer_nz_[0] - 1) * plane_bytes;
            if (!hip_ok(hipMemcpyAsync(dst, base + plane_bytes, plane_bytes, hipMemcpyDeviceToDevice, stream_), "hipMemcpyAsync (plane to rank - 1)", err))
                return false;
            posts[n] = peer_mailbox_[0] + 1 * 4 + field;  // (I am its upper neighbour: side 1 over there)
            values[n++] = k;
        }
        if (has_hi_) {  // my last owned plane -> the upper neighbour's bottom ghost plane
            char* dst = peer_field_[1][field];
            if (!hip_ok(hipMemcpyAsync(dst, base + (size_t)(nz - 2) * plane_bytes, plane_bytes, hipMemcpyDeviceToDevice, stream_),
                        "hipMemcpyAsync (plane to rank + 1)", err))
                return false;
            posts[n] = peer_mailbox_[1] + 0 * 4 + field;
            values[n++] = k;
        }
        if (n && !ipc_post(stream_, n, posts, values, err)) return false;
        if (!hip_ok(hipEventRecord(own_push_, stream_), "hipEventRecord", err)) return false;
        own_push_set_ = true;
        return true;
    }
    Rccl& r = rccl();
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

// words[i] <- min over the ranks, on the halo stream like every other RCCL call of this communicator; host-synchronous
// (the caller needs the answer to decide what to enqueue next).
bool SlabComm::agree_min(hipStream_t stream, uint64_t* words, int n, std::string* err) {
    if (dead_) {
        *err = "the communicator was aborted after a time-out (an earlier call says where)";
        return false;
    }
    if (local_ || n <= 0) return true;
    if (n > kMaxFlags) {
        *err = "agree_min: too many words";
        return false;
    }
    if (!spread_ && !hip_ok(hipMalloc((void**)&spread_, kMaxFlags * sizeof(uint64_t)), "hipMalloc", err)) return false;
    for (hipEvent_t* e : {&reduce_in_, &reduce_out_})
        if (!*e && !hip_ok(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate", err)) return false;
    if (!host_words_ && !hip_ok(hipHostMalloc((void**)&host_words_, kMaxFlags * sizeof(uint64_t), hipHostMallocDefault), "hipHostMalloc", err))
        return false;
    // (`spread_` is also or_flags' scratch: both are issued from the engine's one host thread, in stream order)
    // (through pinned memory: a device-to-host copy into pageable memory makes hipMemcpyAsync itself wait for the stream -- with a
    // peer that never enters the all-reduce that is for ever, and the time-out in sync() below would never be reached)
    std::memcpy(host_words_, words, (size_t)n * sizeof(uint64_t));
    if (!hip_ok(hipMemcpyAsync(spread_, host_words_, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, stream), "hipMemcpyAsync", err))
        return false;
    if (!hip_ok(hipEventRecord(reduce_in_, stream), "hipEventRecord", err)) return false;
    if (!hip_ok(hipStreamWaitEvent(stream_, reduce_in_, 0), "hipStreamWaitEvent", err)) return false;
    if (!nccl_ok(rccl().all_reduce(spread_, spread_, (size_t)n, kNcclUint64, kNcclMin, comm_, stream_), "ncclAllReduce", err))
        return false;
    if (!hip_ok(hipEventRecord(reduce_out_, stream_), "hipEventRecord", err)) return false;
    if (!hip_ok(hipStreamWaitEvent(stream, reduce_out_, 0), "hipStreamWaitEvent", err)) return false;
    if (!hip_ok(hipMemcpyAsync(host_words_, spread_, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync", err))
        return false;
    if (!sync(stream, "the ranks' agreement on the next batch of steps (an all-reduce every rank must enter)", err)) return false;
    std::memcpy(words, host_words_, (size_t)n * sizeof(uint64_t));
    return true;
}

bool SlabComm::sync(hipStream_t stream, const std::string& what, std::string* err) {
    if (dead_) {
        *err = "the communicator was aborted after a time-out (an earlier call says where)";
        return false;
    }
    if (local_ || timeout_s_ <= 0) return hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize", err);
    if (!sync_ev_ && !hip_ok(hipEventCreateWithFlags(&sync_ev_, hipEventDisableTiming), "hipEventCreate", err)) return false;
    if (!hip_ok(hipEventRecord(sync_ev_, stream), "hipEventRecord", err)) return false;
    const auto t0 = std::chrono::steady_clock::now();
    int polls = 0;
    for (;;) {
        const hipError_t q = hipEventQuery(sync_ev_);
        if (q == hipSuccess) {
            if (ipc_status_ && *ipc_status_) {
                const int st = *ipc_status_;
                const std::string which = std::string((st & 0x0F) ? "the lower" : "") + ((st & 0x0F) && (st & 0xF0) ? " and " : "") + ((st & 0xF0) ? "the upper" : "");
                *err = "rank " + std::to_string(rank_) + " of " + std::to_string(nranks_) + ": " + what + ": a wait for " +
                       ((st & 0x11) ? "ghost planes from " : "the end of a step on ") + which + " neighbouring rank timed out on the device after " +
                       std::to_string((int)timeout_s_) + " s (status " + std::to_string(st) +
                       "); this rank has told its neighbours nothing since: the fields are no longer meaningful";
                dead_ = true;
                return false;
            }
            return true;
        }
        if (q != hipErrorNotReady) return hip_ok(q, "hipEventQuery", err);
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > timeout_s_) break;
        // (the first few hundred polls back to back: a batch that is nearly done costs no sleep; then 50 us naps, 1 ms after a second)
        if (++polls > 400) std::this_thread::sleep_for(std::chrono::microseconds(waited > 1.0 ? 1000 : 50));
    }
    (void)hipGetLastError();
    const bool halo_busy = stream_ && hipStreamQuery(stream_) == hipErrorNotReady;
    (void)hipGetLastError();
    std::string peers;
    if (has_lo_) peers += "rank " + std::to_string(loopback_ ? rank_ : rank_ - 1);
    if (has_hi_) peers += std::string(peers.empty() ? "" : " and ") + "rank " + std::to_string(loopback_ ? rank_ : rank_ + 1);
    *err = "rank " + std::to_string(rank_) + " of " + std::to_string(nranks_) + ": " + what + " did not finish within " +
           std::to_string((int)timeout_s_) + " s; " +
           (halo_busy ? "the halo stream is still waiting in an exchange / all-reduce with " + (peers.empty() ? std::string("its peers") : peers)
                      : std::string("the halo stream has drained, the compute stream has not")) +
           " (a peer rank that died or never made the matching call, or the collective library itself); the communicator has been aborted";
    dead_ = true;
    // end what is in flight where the library can (RCCL: the send / receive kernels spinning on a peer that will not answer)
    Rccl& r = rccl();
    if (comm_ && r.comm_abort) (void)r.comm_abort(comm_);
    comm_ = nullptr;
    return false;
}

bool SlabComm::or_flags(hipStream_t stream, int* flags, int n, std::string* err) {
    if (dead_) {
        *err = "the communicator was aborted after a time-out (an earlier call says where)";
        return false;
    }
    if (local_ || n <= 0) return true;  // (a one-rank communicator reduces with itself: the loopback test runs this path)
    if (n > kMaxFlags || nranks_ >= (1 << kFlagField)) {
        *err = "or_flags: too many flag words or ranks";
        return false;
    }
    if (!spread_ && !hip_ok(hipMalloc((void**)&spread_, kMaxFlags * sizeof(uint64_t)), "hipMalloc", err)) return false;
    for (hipEvent_t* e : {&reduce_in_, &reduce_out_})
        if (!*e && !hip_ok(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate", err)) return false;
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(flag_spread_kernel, dim3(grid), dim3(256), 0, stream, (const int*)flags, spread_, n);
    // The all-reduce goes to the halo stream, behind the exchanges: every RCCL call of this communicator is then issued
    // to ONE stream, in the same order on every rank -- no two of its operations can be in flight side by side.
    if (!hip_ok(hipEventRecord(reduce_in_, stream), "hipEventRecord", err)) return false;
    if (!hip_ok(hipStreamWaitEvent(stream_, reduce_in_, 0), "hipStreamWaitEvent", err)) return false;
    if (!nccl_ok(rccl().all_reduce(spread_, spread_, (size_t)n, kNcclUint64, kNcclSum, comm_, stream_), "ncclAllReduce", err))
        return false;
    if (!hip_ok(hipEventRecord(reduce_out_, stream_), "hipEventRecord", err)) return false;
    if (!hip_ok(hipStreamWaitEvent(stream, reduce_out_, 0), "hipStreamWaitEvent", err)) return false;
    hipLaunchKernelGGL(flag_gather_kernel, dim3(grid), dim3(256), 0, stream, (const uint64_t*)spread_, flags, n);
    return hip_ok(hipGetLastError(), "flag OR kernels", err);
}

}  // namespace wv
