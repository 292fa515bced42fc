// comm.h -- z-slab ghost-plane exchange, one slab per engine.
//
// New design (the reference is single-device).  Memory order is x-fastest,
// z-slowest (src/waveguide/src/cl/utils.cpp:33-36), so a slab's face plane and a ghost plane are
// each one contiguous nx*ny run.  Per step, after the two face planes of the new field are
// final, each slab hands them to its z-1 / z+1 neighbours' ghost planes on a dedicated stream
// while the interior planes are still being updated on the compute stream.  Slab chain = nearest
// neighbour only: at most 2 of a GPU's 7 xGMI links.
//
// Three transports behind the same calls (the engine's step is the same code for all):
//   RCCL   one rank per process / GPU: grouped ncclSend/ncclRecv over xGMI.  RCCL is resolved at
//          run time (dlopen) so that a process that already loaded a librccl (e.g. through
//          torch.distributed) shares that copy.
//   IPC    one rank per process / GPU like RCCL, and an RCCL communicator all the same (the ranks' agreements and the flag OR go
//          through it, and so do the handles at set-up) -- but the PLANES travel by copies into the neighbour's own fields,
//          mapped into this process with hipIpcOpenMemHandle: a copy engine moves them (hipMemcpyAsync; a peer copy between
//          GPUs), not a send / receive kernel that has to find CUs beside a march that holds every register of the chip.
//          Ordering between the ranks: counters in a small mailbox of uncached device memory per rank, written by the
//          neighbours (one-thread kernels behind their copies), waited for by one-thread kernels with a time-out.
//   local  all slabs of the chain are engines of THIS process (wv_comm_init_local): device-to-device
//          copies into the neighbour's ghost plane, ordered with events.  The engines must then be
//          stepped in lockstep from one host thread (wv_run_group), which is what makes every
//          event wait refer to a record that has already been enqueued.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace wv {

class SlabComm {
public:
    SlabComm() = default;
    ~SlabComm();
    SlabComm(const SlabComm&) = delete;
    SlabComm& operator=(const SlabComm&) = delete;

    static bool unique_id(void* bytes128, std::string* err);
    // The shared library the RCCL entry points are taken from, instead of the librccl the process finds by name
    // (null / empty: back to that).  Process-wide; before the first communicator call.
    static bool use_library(const char* path, std::string* err);

    bool init(const void* id_bytes128, int rank, int nranks, int device, hipStream_t comm_stream,
              bool has_lo, bool has_hi, std::string* err);
    bool init_local(int rank, int nranks, int device, hipStream_t comm_stream, bool has_lo, bool has_hi,
                    std::string* err);
    // IPC transport: after init() and set_fields() (ALL the fields the engine will ever exchange: the handles travel once), a
    // collective among neighbours -- handles of the fields and of the mailbox go to rank - 1 / rank + 1 through the communicator.
    bool init_ipc(std::string* err);
    bool is_ipc() const { return ipc_; }
    // local transport: the neighbouring slabs' communicators (null at the ends of the chain)
    // (slabs on different GPUs of this process: peer access is switched on where the devices offer it)
    void link_local(SlabComm* lo, SlabComm* hi);
    // The engine's field buffers: base pointers, bytes per plane, planes (ghosts included).
    void set_fields(void* const* fields, int n_fields, size_t plane_bytes, int nz);

    // The compute stream must not read the ghost planes of field buffer `field` before the exchange that fills them
    // has landed -- nor write into this slab's face planes (a source on a slab face) before its own exchanges have read them.
    bool wait_ghosts(hipStream_t compute, int field, std::string* err);
    // Faces of field buffer `field` (planes 1 and nz-2 when the matching ghost exists) are final on
    // `compute`: exchange them into the neighbours' ghost planes (planes nz-1 / 0 over there).
    // `on_halo_stream`: the faces were produced by launches on the halo stream itself (a slab's two-step pass that steps its
    // faces to t+2 there, between the two exchanges: engine_pair.hip.h) -- stream order covers it, `compute` is not involved.
    bool exchange_faces(hipStream_t compute, int field, std::string* err, bool on_halo_stream = false);
    // Everything enqueued on the halo stream so far happens before whatever `compute` is given next (the flag words of a
    // batch are read back on the compute stream; launches on the halo stream write to them too).
    bool join_halo(hipStream_t compute, std::string* err);
    // planes handed to neighbours so far (each plane_bytes() long)
    uint64_t planes_sent() const { return planes_sent_; }
    // exchanges issued so far (single steps: one per step; two-step passes: two per pass)
    uint64_t exchanges() const { return exchanges_; }
    size_t plane_bytes() const { return plane_bytes_; }
    // End of a step on `compute`: this slab has read the ghost planes of the step's `current` field
    // (the local transport may overwrite them once this has passed).
    bool step_done(hipStream_t compute, std::string* err);
    // local transport, several slabs on ONE device: the launch that fills the device (the march of a two-step pass, the interior
    // sweep of a single step) is bracketed by these.  Such launches of different slabs gain nothing from running side by side --
    // each one alone occupies every CU, two of them halve each other's share of L2 -- so they take turns, in the order the host
    // enqueued them; the small launches around them (faces, boundary nodes, copies) still overlap with another slab's turn.
    bool bulk_begin(hipStream_t compute, std::string* err);
    bool bulk_end(hipStream_t compute, std::string* err);
    // flags[i] <- bitwise OR of flags[i] over all ranks, i < n <= kMaxFlags, ordered on `stream`.
    // (RCCL transport; a local group is OR-ed on the host by wv_run_group.)  SURVEY.md 8(e)
    // "error-flag OR": one rank's NaN must stop every rank at the same step.
    static constexpr int kMaxFlags = 1024;
    bool or_flags(hipStream_t stream, int* flags, int n, std::string* err);
    // words[i] <- minimum of words[i] over all ranks (host array; returns when the answer is there).  What a chain
    // agrees on before it enqueues a batch: how many steps, and in which form (engine_batch.hip.h).
    bool agree_min(hipStream_t stream, uint64_t* words, int n, std::string* err);
    // local transport: everything this slab has pushed into its neighbours has landed
    hipStream_t halo_stream() const { return stream_; }

    // ---- the watchdog (RCCL transport) -------------------------------------------------------------------------------------
    // A peer that died, or a collective library that deadlocks, leaves this rank's streams waiting on the device for good:
    // hipStreamSynchronize would never return and the caller would never hear why.  `sync` waits for `stream` like
    // hipStreamSynchronize but gives up after the time-out (an event polled with hipEventQuery): it then says who was waiting
    // for whom (*err: rank, neighbours, which of the two streams had not drained, `what`), aborts the communicator
    // (ncclCommAbort, where the library has it: the kernels still in flight end) and marks this communicator dead -- every later
    // call fails at once and nothing synchronises with its streams again (the process is expected to end).
    // seconds <= 0: no time-out (plain hipStreamSynchronize).  The in-process transport has no peer that could die: plain, too.
    void set_timeout(double seconds) { timeout_s_ = seconds; }
    double timeout() const { return timeout_s_; }
    bool sync(hipStream_t stream, const std::string& what, std::string* err);
    bool dead() const { return dead_; }

    // Does any neighbour live on another GPU (RCCL: always; in-process: a linked slab on another device)?  What runs beside a
    // slab's march then needs a CU of THIS device while the march holds them all; slabs that share one device take turns anyway.
    bool peers_elsewhere() const {
        if (!local_) return true;
        return (lo_ && lo_->device_ != device_) || (hi_ && hi_->device_ != device_);
    }

    int rank() const { return rank_; }
    int nranks() const { return nranks_; }
    bool is_local() const { return local_; }

private:
    void* comm_ = nullptr;
    int rank_ = 0, nranks_ = 1;
    int device_ = 0;  // local transport: the GPU this slab lives on
    bool has_lo_ = false, has_hi_ = false, loopback_ = false, local_ = false;
    hipStream_t stream_ = nullptr;
    hipEvent_t faces_ready_ = nullptr;
    hipEvent_t ghosts_ready_ = nullptr;
    hipEvent_t halo_joined_ = nullptr;
    uint64_t planes_sent_ = 0, exchanges_ = 0;
    double timeout_s_ = 0;
    bool dead_ = false;
    hipEvent_t sync_ev_ = nullptr;
    hipEvent_t reduce_in_ = nullptr, reduce_out_ = nullptr;  // or_flags: compute stream -> halo stream -> compute stream
    bool pending_ = false;
    // field geometry
    void* fields_[4] = {nullptr, nullptr, nullptr, nullptr};
    int n_fields_ = 0, nz_ = 0;
    size_t plane_bytes_ = 0;
    // local transport
    SlabComm *lo_ = nullptr, *hi_ = nullptr;
    // my face of field buffer f has landed in the lower / upper neighbour (one event per buffer: a slab waits for the
    // pushes into the buffer it is about to read, whatever its neighbours have pushed since)
    hipEvent_t pushed_lo_[4] = {nullptr, nullptr, nullptr, nullptr}, pushed_hi_[4] = {nullptr, nullptr, nullptr, nullptr};
    bool pushed_lo_set_[4] = {false, false, false, false}, pushed_hi_set_[4] = {false, false, false, false};
    hipEvent_t last_own_push_ = nullptr;  // the latest of the above to be recorded
    // ends of this slab's steps / two-step passes, by parity of their count
    hipEvent_t step_done_[2] = {nullptr, nullptr};
    hipEvent_t bulk_done_ = nullptr;
    uint64_t steps_done_ = 0;
    // IPC transport.  Mailbox words (uint64, uncached device memory, written by the NEIGHBOURS): [side * 4 + field] = exchanges of
    // field buffer `field` whose plane has landed in my ghost plane on that side (side 0: from rank - 1, 1: from rank + 1);
    // [8 + side] = steps / passes that neighbour has finished (it no longer reads the ghost planes I am about to overwrite).
    static constexpr int kMailboxWords = 16;
    bool ipc_ = false;
    bool ipc_wait(hipStream_t stream, int n, const uint64_t* const* flags, const uint64_t* values, int code, std::string* err);
    bool ipc_post(hipStream_t stream, int n, uint64_t* const* flags, const uint64_t* values, std::string* err);
    uint64_t* mailbox_ = nullptr;
    uint64_t* peer_mailbox_[2] = {nullptr, nullptr};
    char* peer_field_[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    void* peer_opened_[2][5] = {{nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr}};
    int peer_nz_[2] = {0, 0};
    uint64_t pushes_[4] = {0, 0, 0, 0};
    int* ipc_status_ = nullptr;  // pinned: set by a wait that timed out on the device (which side, which kind)
    hipEvent_t own_push_ = nullptr;
    bool own_push_set_ = false;
    // flag OR
    uint64_t* spread_ = nullptr;
    uint64_t* host_words_ = nullptr;  // pinned: agree_min's way in and out (a pageable copy would wait for the stream on the host)
};

}  // namespace wv
