// engine_base.h -- what every part of the engine's host side shares: error reporting, HIP call checking, scoped
// device selection, and `wv_engine`, the interface the C ABI (engine.hip) talks to.
#pragma once
#include "../../include/wayverb_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "comm.h"

namespace wv {

// the calling thread's most recent failure (wv_last_error); returns `code`
int fail_with(int code, const std::string& msg);
inline int fail(int code, const std::string& msg) { return fail_with(code, msg); }
// the HIP status behind the calling thread's most recent WV_E_HIP (what tells "no memory for this" from a fault)
void note_hip_error(hipError_t err);
hipError_t last_hip_error();

#define WV_HIP(expr)                                                                            \
    do {                                                                                          \
        hipError_t err__ = (expr);                                                                \
        if (err__ != hipSuccess) {                                                                \
            ::wv::note_hip_error(err__);                                                          \
            return ::wv::fail(WV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));    \
        }                                                                                         \
    } while (0)

constexpr int kRing = 1024;  // steps per device batch (flag words / receiver rows kept on device)

// a device allocation that lives for one scope (the WV_HIP early returns must not leak it)
struct ScopedDevice {
    void* p = nullptr;
    ~ScopedDevice() {
        if (p) (void)hipFree(p);
    }
};

// Selects the engine's device for the duration of a public call and restores the caller's: two
// engines on different GPUs may be driven from one thread (wv_options::device).
struct DeviceGuard {
    int before = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&before) == hipSuccess && before != device && device >= 0)
            switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(before);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct StreamPlan {
    int variant = 2;  // 2 = plane sweep (default), 0 = z-march, 1 = naive
    int ry = 4, nwx = 1, nwy = 4;
    int zc = 0, tiles_x = 0, tiles_y = 0;
    int stripe_rows = 0, tiles_y_stripe = 0, passes = 0;
    unsigned grid = 0, block = 0;
};

}  // namespace wv

// ---------------------------------------------------------------------------------------------
struct wv_engine {
    virtual ~wv_engine() {}
    virtual int init(const wv_mesh& mesh, const wv_options& opt) = 0;
    virtual int read_value(int buffer, uint64_t index, double* v) = 0;
    virtual int write_value(int buffer, uint64_t index, double v) = 0;
    virtual int read_field(int buffer, void* dst, int elem_size) = 0;
    virtual int write_field(int buffer, const void* src, int elem_size) = 0;
    virtual int read_planes(int buffer, int z0, int planes, void* dst, int elem_size) = 0;
    virtual int write_planes(int buffer, int z0, int planes, const void* src, int elem_size) = 0;
    virtual int boundary_data(int dim, wv_boundary_data* host, bool to_device) = 0;
    virtual int set_coefficients(const wv_coefficients_canonical* c, uint32_t n) = 0;
    virtual int device_buffer(int buffer, void** p) = 0;
    virtual int checkpoint(int op) = 0;  // 0 save, 1 restore, 2 drop (wv_checkpoint / wv_rollback / wv_drop_checkpoint)
    virtual int step(int32_t* flag) = 0;
    virtual int swap() = 0;
    virtual int set_source(int kind, uint64_t node, const double* signal, uint64_t n) = 0;
    virtual int set_receivers(const uint64_t* nodes, uint32_t n) = 0;
    virtual int run(uint64_t n_steps, uint64_t* done, int32_t* flag) = 0;
    virtual int fetch_receivers(uint64_t first, uint64_t n, double* dst) = 0;
    virtual int kernel_time(double* mean_ms, uint64_t* launches, uint64_t* steps) = 0;
    virtual int synchronize() = 0;
    virtual int query(int what, uint64_t* value) = 0;
    virtual int set_tuning(int variant, int ry, int nwx, int nwy, int zchunks) = 0;
    virtual int comm_init(const void* id, int rank, int nranks) = 0;
    virtual int comm_init_local(int rank, int nranks) = 0;
    virtual wv::SlabComm* comm() = 0;
    virtual int comm_destroy() = 0;
    // a batch of steps in parts, so that a group of slabs can be driven in lockstep (wv_run_group)
    virtual uint64_t plan_batch(uint64_t remaining) = 0;
    // next_kind: what follows in the same batch -- 0 nothing, 1 a single step, 2 a two-step pass
    virtual int enqueue_batch_step(uint64_t i, uint64_t batch, int next_kind) = 0;
    virtual int enqueue_batch_pair(uint64_t i, int part, int next_kind) = 0;
    // Two-step passes for the batch being planned, decided by all slabs of a chain together (their exchanges must
    // pair up): first the cheap question -- would this engine take them at all -- and only when every slab says
    // yes the set-up that costs memory and time (two more fields, the pair map, the lists); *singles_first = the
    // number of single full sweeps this engine needs first (0, 1 or 2: a caller wrote into outside nodes).
    virtual int batch_pair_eligible(int* eligible) = 0;
    virtual int batch_pair_prepare(int* ready, int* singles_first) = 0;
    virtual int batch_pair_vetoed() = 0;  // the chain stays with single steps: the spare fields go back
    // three-step passes (engine_triple.hip.h): *ready = this engine can take them in the batch being planned (everything allocated and built);
    // a pass of a slab is enqueued in three parts, each part of every slab of an in-process chain before the next part of any
    virtual int batch_triple_prepare(int* ready) = 0;
    virtual int enqueue_batch_triple(uint64_t i, int part) = 0;
    virtual uint64_t role_signature() const = 0;  // which field buffer plays which role, and after how many steps
    virtual int collect_batch(uint64_t batch) = 0;
    virtual const int* batch_flags() const = 0;
    virtual int commit_batch(uint64_t batch, const int* flags, uint64_t* good, int32_t* flag) = 0;
    virtual uint64_t field_pitch() const = 0;
    uint64_t steps_done = 0;
    bool timing = false;
};
