// engine.hip -- host side of the MI355X waveguide engine + the C ABI of include/wayverb_amd.h.
//
// Replaces the body of `waveguide::run` (src/waveguide/include/waveguide/waveguide.h:36-126):
// device buffers, the step loop, the error-flag protocol, and (device-resident) the single-node
// source / node-gather receivers every caller of `run` uses.
//
// There is no CPU path in this library: without a HIP device every entry point fails.
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

#include "boundary_kernels.hip.h"
#include "comm.h"
#include "pair_kernels.hip.h"
#include "stream_kernels.hip.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

}  // namespace

namespace wv {
// for the other translation units of the library (mesh_setup.hip)
int fail_with(int code, const std::string& msg) { return fail(code, msg); }
}  // namespace wv

namespace {

#define WV_HIP(expr)                                                                            \
    do {                                                                                          \
        hipError_t err__ = (expr);                                                                \
        if (err__ != hipSuccess)                                                                  \
            return fail(WV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));          \
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

int env_int(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}

}  // namespace

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
    virtual int step(int32_t* flag) = 0;
    virtual int swap() = 0;
    virtual int set_source(int kind, uint64_t node, const double* signal, uint64_t n) = 0;
    virtual int set_receivers(const uint64_t* nodes, uint32_t n) = 0;
    virtual int run(uint64_t n_steps, uint64_t* done, int32_t* flag) = 0;
    virtual int fetch_receivers(uint64_t first, uint64_t n, double* dst) = 0;
    virtual int kernel_time(double* mean_ms, uint64_t* launches, uint64_t* steps) = 0;
    virtual int synchronize() = 0;
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
    virtual int batch_pairs_ready(int* singles_first) = 0;
    virtual int collect_batch(uint64_t batch) = 0;
    virtual const int* batch_flags() const = 0;
    virtual int commit_batch(uint64_t batch, const int* flags, uint64_t* good, int32_t* flag) = 0;
    virtual uint64_t field_pitch() const = 0;
    uint64_t steps_done = 0;
    bool timing = false;
};

namespace {

template <typename Real>
class Engine final : public wv_engine {
public:
    ~Engine() override { release(); }

    int init(const wv_mesh& m, const wv_options& opt) override {
        opt_ = opt;
        nx_ = m.nx;
        ny_ = m.ny;
        nz_ = m.nz;
        if (nx_ < 1 || ny_ < 1 || nz_ < 1) return fail(WV_E_INVALID_ARGUMENT, "mesh dimensions must be positive");
        n_nodes_ = (uint64_t)nx_ * ny_ * nz_;
        // stored rows are padded to the wave tile width (64 lanes x 16 B), see stream_kernels.hip.h
        constexpr int kTile = 64 * (16 / (int)sizeof(Real));
        pitch_ = (nx_ + kTile - 1) / kTile * kTile;
        stored_nodes_ = (uint64_t)pitch_ * ny_ * nz_;
        if (stored_nodes_ >= 0xFFFFFFFEull)
            return fail(WV_E_INVALID_ARGUMENT,
                        "more than 2^32-2 stored nodes in one engine: decompose into z-slabs (32-bit local node indices)");
        if (!m.nodes || (!m.coefficients && m.num_coefficients))
            return fail(WV_E_INVALID_ARGUMENT, "mesh arrays missing");
        z_begin_ = opt.ghost_lo ? 1 : 0;
        z_end_ = opt.ghost_hi ? nz_ - 1 : nz_;
        if (z_end_ <= z_begin_) return fail(WV_E_INVALID_ARGUMENT, "slab has no owned planes");

        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
            return fail(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
        if (opt.device >= count) return fail(WV_E_INVALID_ARGUMENT, "no such HIP device");
        DeviceGuard guard(opt.device);  // the caller's current device is restored on return
        WV_HIP(hipGetDevice(&device_));
        WV_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        WV_HIP(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));

        // ---- pressure fields (zeroed: make_zeroed_buffer, waveguide.h:47-56) -------------------
        field_bytes_ = stored_nodes_ * sizeof(Real);
        for (int i = 0; i < 2; ++i) {
            WV_HIP(hipMalloc((void**)&field_[i], field_bytes_ + 256));
            WV_HIP(hipMemsetAsync(field_[i], 0, field_bytes_ + 256, stream_));
        }
        prv_ = 0;  // field_[0] = previous, field_[1] = current; [2], [3]: outputs of a two-step pass (ensure_pair)
        cur_ = 1;

        // ---- class map + compact boundary lists ------------------------------------------------
        cls_pitch_ = pitch_ / 4;
        // one 32-bit word per (row group of 4, quad of 4 nodes): see cls_word_index
        const uint64_t cls_bytes = (uint64_t)cls_pitch_ * 4u * (uint64_t)((ny_ + 3) / 4) * nz_;
        WV_HIP(hipMalloc((void**)&cls_, cls_bytes + 16));
        WV_HIP(hipMemsetAsync(cls_, 0, cls_bytes + 16, stream_));
        n1_ = (uint32_t)m.num_boundary_1;
        n2_ = (uint32_t)m.num_boundary_2;
        n3_ = (uint32_t)m.num_boundary_3;
        if (m.num_boundary_1 + m.num_boundary_2 + m.num_boundary_3 >= 0xFFFFFFFFull)
            return fail(WV_E_INVALID_ARGUMENT, "too many boundary nodes");
        n_entries_ = n1_ + n2_ + n3_;
        n_slots_ = n1_ + 2u * n2_ + 3u * n3_;
        const size_t ne = std::max<size_t>(n_entries_, 1), ns = std::max<size_t>(n_slots_, 1);
        WV_HIP(hipMalloc((void**)&bnode_, ne * sizeof(uint32_t)));
        WV_HIP(hipMemsetAsync(bnode_, 0xFF, ne * sizeof(uint32_t), stream_));
        WV_HIP(hipMalloc((void**)&btype_, ne));
        WV_HIP(hipMemsetAsync(btype_, 0, ne, stream_));
        WV_HIP(hipMalloc((void**)&fmem_, ns * 6 * sizeof(double)));
        WV_HIP(hipMemsetAsync(fmem_, 0, ns * 6 * sizeof(double), stream_));
        WV_HIP(hipMalloc((void**)&cidx_, ns * sizeof(uint32_t)));
        WV_HIP(hipMalloc((void**)&status_, 4 * sizeof(int)));
        WV_HIP(hipMemsetAsync(status_, 0, 4 * sizeof(int), stream_));
        static_flag_dev_ = status_ + 1;

        // host nodes are staged through a bounded device buffer, whole x-rows at a time; nodes that
        // already live on this device (wv_scene_mesh_create_engine) are classified where they are
        {
            const bool resident = opt.nodes_on_device != 0;
            const int64_t rows_total = (int64_t)ny_ * nz_;
            const int64_t rows_per_chunk = resident ? rows_total : std::max<int64_t>(1, (int64_t)(32 << 20) / nx_);
            ScopedDevice stage_mem;
            if (!resident) WV_HIP(hipMalloc(&stage_mem.p, (size_t)rows_per_chunk * nx_ * sizeof(wv::NodeRec)));
            const wv::NodeRec* stage = resident ? reinterpret_cast<const wv::NodeRec*>(m.nodes)
                                                : static_cast<const wv::NodeRec*>(stage_mem.p);
            for (int64_t row = 0; row < rows_total; row += rows_per_chunk) {
                const int64_t rows = std::min(rows_per_chunk, rows_total - row);
                const int64_t first = row * nx_, cnt = rows * nx_;
                if (!resident)
                    WV_HIP(hipMemcpyAsync(stage_mem.p, m.nodes + first, (size_t)cnt * sizeof(wv::NodeRec),
                                          hipMemcpyHostToDevice, stream_));
                wv::SetupArgs a{};
                a.nodes = stage;
                a.first_row = row;
                a.rows = rows;
                a.nx = nx_;
                a.ny = ny_;
                a.pitch = pitch_;
                a.cls_pitch = cls_pitch_;
                a.cls = cls_;
                a.bnode = bnode_;
                a.btype = btype_;
                a.n1 = n1_;
                a.n2 = n2_;
                a.n3 = n3_;
                a.status = status_;
                a.z_begin = z_begin_;
                a.z_end = z_end_;
                const int64_t n_bytes = rows * cls_pitch_;
                const unsigned grid = (unsigned)std::min<int64_t>((n_bytes + 255) / 256, 65536);
                hipLaunchKernelGGL(wv::setup_classify_kernel, dim3(grid), dim3(256), 0, stream_, a);
                WV_HIP(hipGetLastError());
                WV_HIP(hipStreamSynchronize(stream_));  // `stage` is reused by the next chunk
            }
        }
        if (n_entries_) {
            wv::ValidateArgs v{};
            v.bnode = bnode_;
            v.btype = btype_;
            v.cls = cls_;
            v.n_entries = n_entries_;
            v.nx = nx_;
            v.ny = ny_;
            v.nz = nz_;
            v.pitch = pitch_;
            v.cls_pitch = cls_pitch_;
            v.static_flag = static_flag_dev_;
            hipLaunchKernelGGL(wv::setup_validate_kernel, dim3((n_entries_ + 255) / 256), dim3(256), 0, stream_, v);
            WV_HIP(hipGetLastError());
        }
        int status_host[4] = {0, 0, 0, 0};
        WV_HIP(hipMemcpyAsync(status_host, status_, sizeof(status_host), hipMemcpyDeviceToHost, stream_));
        WV_HIP(hipStreamSynchronize(stream_));
        if (status_host[0] & 1)
            return fail(WV_E_INVALID_MESH,
                        "node with an invalid boundary_type (boundary bits must be 1-3 direction bits on distinct axes)");
        if (status_host[0] & 2) return fail(WV_E_INVALID_MESH, "boundary_index exceeds the boundary array length");
        static_flag_ = status_host[1];

        // ---- processing order of the boundary entries: inside each dimensionality class, sort by
        // 64 x 8 x 8 (x, y, z) brick, then z, y, x inside the brick.  Runs along x stay runs (the
        // y- and z-walls keep their coalescing); nodes isolated in x (the x-walls) end up as 8 x 8
        // (y, z) patches per wave, so that a wave's `current` neighbours share cache lines instead
        // of touching four private lines per node.  Filter slots follow the processing order;
        // `ref_to_pos_` translates the caller's boundary_index wherever it crosses the ABI.
        std::vector<uint32_t> ref_to_pos(ne);
        for (uint32_t e = 0; e < n_entries_; ++e) ref_to_pos[e] = e;
        WV_HIP(hipMalloc((void**)&ref_to_pos_, ne * sizeof(uint32_t)));
        if (n_entries_ && env_int("WV_BOUNDARY_ORDER", 1) != 0) {
            std::vector<uint32_t> bnode(n_entries_);
            std::vector<uint8_t> btype(n_entries_);
            WV_HIP(hipMemcpy(bnode.data(), bnode_, (size_t)n_entries_ * sizeof(uint32_t), hipMemcpyDeviceToHost));
            WV_HIP(hipMemcpy(btype.data(), btype_, (size_t)n_entries_, hipMemcpyDeviceToHost));
            const uint32_t nd[3] = {n1_, n2_, n3_};
            const uint64_t bricks_x = ((uint64_t)pitch_ + 63) / 64, bricks_y = ((uint64_t)ny_ + 7) / 8;
            std::vector<std::pair<uint64_t, uint32_t>> keyed(n_entries_);  // (sort key, entry): ties keep list order
            std::vector<uint32_t> by_pos(n_entries_);
            uint32_t off = 0;
            for (int d = 0; d < 3; ++d) {
                for (uint32_t k = 0; k < nd[d]; ++k) {
                    const uint32_t idx = bnode[off + k];
                    uint64_t kk = ~0ull >> 8;  // entries this engine does not own go last
                    if (idx != wv::INVALID_NODE) {
                        const uint64_t x = idx % (uint32_t)pitch_, q = idx / (uint32_t)pitch_;
                        const uint64_t y = q % (uint32_t)ny_, z = q / (uint32_t)ny_;
                        const uint64_t brick = ((z >> 3) * bricks_y + (y >> 3)) * bricks_x + (x >> 6);
                        kk = (brick << 12) | ((z & 7) << 9) | ((y & 7) << 6) | (x & 63);
                    }
                    keyed[off + k] = {kk, off + k};
                }
                std::sort(keyed.begin() + off, keyed.begin() + off + nd[d]);
                for (uint32_t k = 0; k < nd[d]; ++k) by_pos[off + k] = keyed[off + k].second;
                off += nd[d];
            }
            std::vector<uint32_t> bnode2(n_entries_);
            std::vector<uint8_t> btype2(n_entries_);
            for (uint32_t pos = 0; pos < n_entries_; ++pos) {
                bnode2[pos] = bnode[by_pos[pos]];
                btype2[pos] = btype[by_pos[pos]];
                ref_to_pos[by_pos[pos]] = pos;
            }
            WV_HIP(hipMemcpy(bnode_, bnode2.data(), (size_t)n_entries_ * sizeof(uint32_t), hipMemcpyHostToDevice));
            WV_HIP(hipMemcpy(btype_, btype2.data(), (size_t)n_entries_, hipMemcpyHostToDevice));
        }
        WV_HIP(hipMemcpy(ref_to_pos_, ref_to_pos.data(), ne * sizeof(uint32_t), hipMemcpyHostToDevice));

        // ---- filter state: coefficient indices per filter slot (get_boundary_data<N>, setup.h:68-85)
        {
            std::vector<uint32_t> cidx(ns, 0u);
            const uint32_t* src[3] = {m.boundary_indices_1, m.boundary_indices_2, m.boundary_indices_3};
            const uint32_t nd[3] = {n1_, n2_, n3_};
            uint32_t base = 0, entry_off = 0;
            for (int d = 1; d <= 3; ++d) {
                if (nd[d - 1] && !src[d - 1]) return fail(WV_E_INVALID_ARGUMENT, "boundary index array missing");
                for (uint32_t k = 0; k < nd[d - 1]; ++k)
                    for (int i = 0; i < d; ++i) {
                        const uint32_t c = src[d - 1][(size_t)k * d + i];
                        if (c >= m.num_coefficients)
                            return fail(WV_E_INVALID_MESH, "coefficient index exceeds the coefficient array length");
                        cidx[base + (uint32_t)i * nd[d - 1] + (ref_to_pos[entry_off + k] - entry_off)] = c;
                    }
                base += (uint32_t)d * nd[d - 1];
                entry_off += nd[d - 1];
            }
            WV_HIP(hipMemcpy(cidx_, cidx.data(), ns * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        n_coeffs_ = m.num_coefficients;
        WV_HIP(hipMalloc((void**)&coeffs_, std::max<size_t>(n_coeffs_, 1) * sizeof(wv_coefficients_canonical)));
        if (n_coeffs_)
            WV_HIP(hipMemcpy(coeffs_, m.coefficients, n_coeffs_ * sizeof(wv_coefficients_canonical),
                             hipMemcpyHostToDevice));

        // ---- per-step rings ---------------------------------------------------------------------
        WV_HIP(hipMalloc((void**)&flags_, (kRing + 1) * sizeof(int)));  // + one word for collective decisions
        WV_HIP(hipHostMalloc((void**)&flags_host_, kRing * sizeof(int), hipHostMallocDefault));
        WV_HIP(hipMalloc((void**)&scratch_, 64));

        // courant numbers in the pressure type (program.cpp:12-13)
        courant_ = (Real)1 / (Real)std::sqrt((Real)3);
        courant_sq_ = (Real)1 / (Real)3;

        plan_stream();
        const int n_ev = 2 * kRing;
        events_.resize(n_ev);
        for (auto& e : events_) WV_HIP(hipEventCreate(&e));
        return WV_OK;
    }

    // -------------------------------------------------------------------------------------------
    int set_tuning(int variant, int ry, int nwx, int nwy, int zchunks) override {
        if (variant < 0 || variant > 3) return fail(WV_E_INVALID_ARGUMENT, "unknown stream variant");
        tune_variant_ = variant;
        tune_ry_ = ry;
        tune_nwx_ = nwx;
        tune_nwy_ = nwy;
        tune_zchunks_ = zchunks;
        plan_stream();
        return WV_OK;
    }

    // Boundary entries sorted by plane (stable: list order inside a plane), so that the boundary
    // nodes of a plane range are one contiguous run of `zorder_`.  Only the slab path needs it: the
    // face planes' boundary nodes must be final before the halo exchange, the rest follow the
    // interior sweep.
    int build_plane_order() {
        if (zorder_ || !n_entries_) return WV_OK;
        std::vector<uint32_t> bnode(n_entries_);
        WV_HIP(hipMemcpy(bnode.data(), bnode_, (size_t)n_entries_ * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t plane = (uint32_t)pitch_ * (uint32_t)ny_;
        plane_start_.assign((size_t)nz_ + 1, 0);
        for (uint32_t e = 0; e < n_entries_; ++e)
            if (bnode[e] != wv::INVALID_NODE) ++plane_start_[bnode[e] / plane + 1];
        for (int z = 0; z < nz_; ++z) plane_start_[z + 1] += plane_start_[z];
        std::vector<uint32_t> order(std::max<uint32_t>(plane_start_[nz_], 1)), cursor(plane_start_.begin(), plane_start_.end() - 1);
        for (uint32_t e = 0; e < n_entries_; ++e)
            if (bnode[e] != wv::INVALID_NODE) order[cursor[bnode[e] / plane]++] = e;
        uint32_t* staged = nullptr;
        WV_HIP(hipMalloc((void**)&staged, order.size() * sizeof(uint32_t)));
        if (hipMemcpy(staged, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(staged);
            return fail(WV_E_HIP, "copying the plane order of the boundary entries to the device failed");
        }
        zorder_ = staged;
        return WV_OK;
    }

    // variant 2 (default): plane sweep, L2-resident z reuse; 0: register z-march; 1: naive
    void plan_stream() {
        lists_built_ = false;  // tile shapes may change
        StreamPlan& p = plan_;
        constexpr int VX = 16 / (int)sizeof(Real);
        constexpr int WX = 64 * VX;
        p.variant = tune_variant_ >= 0 ? tune_variant_ : env_int("WV_STREAM_VARIANT", opt_.stream_variant);
        if (p.variant < 0 || p.variant > 3) p.variant = 2;
        p.ry = tune_ry_ > 0 ? tune_ry_ : env_int("WV_STREAM_RY", 4);
        // measured best shape (profiles/r01/variant_scan_*): 1 x 4 waves (a 4-wave column shares its
        // y halos through LDS) in both precisions
        p.nwx = tune_nwx_ > 0 ? tune_nwx_ : env_int("WV_STREAM_NWX", 1);
        p.nwy = tune_nwy_ > 0 ? tune_nwy_ : env_int("WV_STREAM_NWY", 4);
        if (p.ry != 2 && p.ry != 4) p.ry = 4;
        {
            const int key = p.nwx * 10 + p.nwy;
            const int ok[] = {11, 14, 18, 22, 24, 41, 42, 81};
            bool found = false;
            for (int k : ok) found = found || k == key;
            if (!found) {
                p.nwx = 1;
                p.nwy = 4;
            }
        }
        p.tiles_x = (pitch_ + WX * p.nwx - 1) / (WX * p.nwx);
        p.tiles_y = (ny_ + p.ry * p.nwy - 1) / (p.ry * p.nwy);
        p.block = 64u * (unsigned)(p.nwx * p.nwy);
        const int owned = z_end_ - z_begin_;
        const int64_t knob = tune_zchunks_ > 0 ? tune_zchunks_ : env_int("WV_STREAM_ZCHUNKS", 0);
        if (p.variant == 2 || p.variant == 3) {
            // stripe height: three `cur` planes of a stripe should sit comfortably in one XCD's
            // 4 MiB L2 (measured best at 0.75-1.5 MiB), at least 8 stripes so every XCD has one
            const int tile_rows = p.ry * p.nwy;
            int64_t rows = knob > 0 ? knob : (int64_t)(1600 * 1024) / (3ll * pitch_ * (int64_t)sizeof(Real));
            if (knob > 0) {
                rows = std::max<int64_t>(tile_rows, rows / tile_rows * tile_rows);  // explicit: any whole number of tiles
            } else {
                int pow2 = tile_rows;
                while (pow2 * 2 <= rows) pow2 *= 2;
                rows = pow2;
            }
            const int per_xcd = (((ny_ + 7) / 8) + tile_rows - 1) / tile_rows * tile_rows;
            if (knob <= 0) rows = std::min<int64_t>(rows, per_xcd);
            rows = std::max<int64_t>(rows, 1);
            p.stripe_rows = (int)rows;
            p.tiles_y_stripe = (p.stripe_rows + tile_rows - 1) / tile_rows;
            const int stripes = (ny_ + p.stripe_rows - 1) / p.stripe_rows;
            p.passes = (stripes + 7) / 8;
            return;
        }
        if (p.variant == 1) {
            p.block = 256;
            p.grid = (unsigned)std::min<uint64_t>((n_nodes_ + 255) / 256, 256ull * 64);
            return;
        }
        // variant 0: enough workgroups to fill 256 CUs a few times over
        const int64_t wave_tiles = (int64_t)p.tiles_x * p.tiles_y * p.nwx * p.nwy;
        int64_t want = knob;
        if (want <= 0) want = (65536 + wave_tiles - 1) / wave_tiles;
        want = std::max<int64_t>(1, std::min<int64_t>(want, owned));
        p.zc = (int)((owned + want - 1) / want);
    }

    template <int RY, int NWX, int NWY>
    void launch_shape(const wv::StreamArgs<Real>& a, unsigned grid) {
        if (plan_.variant == 2) {
            hipLaunchKernelGGL((wv::stream_sweep_kernel<Real, RY, NWX, NWY>), dim3(grid), dim3(64 * NWX * NWY), 0,
                               stream_, a);
        } else if (plan_.variant == 3) {
            hipLaunchKernelGGL((wv::stream_sweep_nolds_kernel<Real, RY, NWX, NWY>), dim3(grid), dim3(64 * NWX * NWY), 0,
                               stream_, a);
        } else {
            hipLaunchKernelGGL((wv::stream_march_kernel<Real, RY, NWX, NWY>), dim3(grid), dim3(64 * NWX * NWY), 0,
                               stream_, a);
        }
    }
    template <int RY>
    void launch_ry(const wv::StreamArgs<Real>& a, unsigned grid) {
        switch (plan_.nwx * 10 + plan_.nwy) {
            case 11: launch_shape<RY, 1, 1>(a, grid); break;
            case 22: launch_shape<RY, 2, 2>(a, grid); break;
            case 41: launch_shape<RY, 4, 1>(a, grid); break;
            case 42: launch_shape<RY, 4, 2>(a, grid); break;
            case 81: launch_shape<RY, 8, 1>(a, grid); break;
            case 18: launch_shape<RY, 1, 8>(a, grid); break;
            case 24: launch_shape<RY, 2, 4>(a, grid); break;
            default: launch_shape<RY, 1, 4>(a, grid); break;
        }
    }

    // the pressure update of planes [z0, z1)
    // Work lists for the plane sweep (variant 2).  A workgroup tile takes part only if it holds an
    // inside or re-entrant node: outside nodes are 0 and stay 0, boundary nodes belong to the
    // boundary kernel.  Whole stripes are dealt to the 8 XCDs heaviest first (each XCD still
    // sweeps its stripes plane by plane, so the z reuse in its L2 is unchanged); a mesh that is
    // almost all room (a box) keeps the arithmetic mapping.
    int build_tile_lists(int z0, int z1) {
        if (lists_built_) return WV_OK;
        lists_built_ = true;
        lists_z0_ = z0;
        lists_z1_ = z1;
        if (tile_list_) {
            (void)hipFree(tile_list_);
            tile_list_ = nullptr;
        }
        if ((plan_.variant != 2 && plan_.variant != 3) || env_int("WV_TILE_LISTS", opt_.all_tiles ? 0 : 1) == 0) return WV_OK;
        // activity per wave tile (ry rows x one wave of columns); a workgroup tile is nwy x nwx of them
        const int wave_cols = 64 * (16 / (int)sizeof(Real));
        const int wtiles_x = plan_.tiles_x * plan_.nwx;
        const int wtiles_y = (ny_ + plan_.ry - 1) / plan_.ry;
        const int tile_rows = plan_.ry * plan_.nwy;
        const int tiles_y = (ny_ + tile_rows - 1) / tile_rows;
        const int64_t n_tiles = (int64_t)nz_ * wtiles_y * wtiles_x;
        ScopedDevice act_mem;
        WV_HIP(hipMalloc(&act_mem.p, (size_t)n_tiles));
        wv::TileActivityArgs t{};
        t.cls = cls_;
        t.active = static_cast<uint8_t*>(act_mem.p);
        t.ny = ny_;
        t.nz = nz_;
        t.pitch = pitch_;
        t.cls_pitch = cls_pitch_;
        t.tile_rows = plan_.ry;
        t.tile_cols = wave_cols;
        t.tiles_x = wtiles_x;
        t.tiles_y = wtiles_y;
        hipLaunchKernelGGL(wv::tile_activity_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, stream_, t);
        WV_HIP(hipGetLastError());
        std::vector<uint8_t> active((size_t)n_tiles);
        WV_HIP(hipMemcpyAsync(active.data(), act_mem.p, (size_t)n_tiles, hipMemcpyDeviceToHost, stream_));
        WV_HIP(hipStreamSynchronize(stream_));

        const int stripes = (ny_ + plan_.stripe_rows - 1) / plan_.stripe_rows;
        const int tys = plan_.tiles_y_stripe;
        // wave mask of workgroup tile (z, ty, tx): bit wy * nwx + wx
        auto wave_mask = [&](int z, int ty, int tx) -> uint32_t {
            uint32_t m = 0;
            for (int wy = 0; wy < plan_.nwy; ++wy) {
                const int wty = ty * plan_.nwy + wy;
                if (wty >= wtiles_y) break;
                for (int wx = 0; wx < plan_.nwx; ++wx) {
                    const int wtx = tx * plan_.nwx + wx;
                    if (wtx < wtiles_x && active[((size_t)z * wtiles_y + wty) * wtiles_x + wtx]) m |= 1u << (wy * plan_.nwx + wx);
                }
            }
            return m;
        };
        std::vector<uint64_t> per_stripe((size_t)stripes, 0);
        uint64_t total_active = 0, total = 0;
        for (int z = z0; z < z1; ++z)
            for (int ty = 0; ty < tiles_y; ++ty)
                for (int tx = 0; tx < plan_.tiles_x; ++tx) {
                    const uint64_t on = (uint64_t)__builtin_popcount(wave_mask(z, ty, tx));
                    per_stripe[(size_t)(ty / tys)] += on;
                    total_active += on;
                    total += (uint64_t)(plan_.nwx * plan_.nwy);
                }
        tile_active_frac_ = total ? (double)total_active / (double)total : 1.0;
        if (total_active * 100 >= total * 92 || stripes >= (1 << 16) || nz_ >= (1 << 20) ||
            (int64_t)plan_.tiles_x * tys >= (1 << 20) || plan_.nwx * plan_.nwy > 8)
            return WV_OK;  // (nearly) everything is room: the arithmetic mapping is as good

        // heaviest stripe first onto the least loaded XCD
        std::vector<int> order((size_t)stripes);
        for (int i = 0; i < stripes; ++i) order[(size_t)i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return per_stripe[(size_t)a] > per_stripe[(size_t)b]; });
        std::vector<std::vector<int>> mine(8);
        uint64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int sidx : order) {
            int best = 0;
            for (int k = 1; k < 8; ++k)
                if (load[k] < load[best]) best = k;
            mine[(size_t)best].push_back(sidx);
            load[best] += per_stripe[(size_t)sidx];
        }
        std::vector<uint64_t> list;
        list.reserve((size_t)total_active / 2 + 16);
        list_longest_ = 0;
        for (int k = 0; k < 8; ++k) {
            list_start_[k] = (uint32_t)list.size();
            for (int sidx : mine[(size_t)k])
                for (int z = z0; z < z1; ++z)
                    for (int tyl = 0; tyl < tys; ++tyl) {
                        const int ty = sidx * tys + tyl;
                        if (ty >= tiles_y) break;
                        for (int tx = 0; tx < plan_.tiles_x; ++tx) {
                            const uint32_t m = wave_mask(z, ty, tx);
                            if (m)
                                list.push_back(((uint64_t)sidx << 48) | ((uint64_t)m << 40) | ((uint64_t)z << 20) |
                                               (uint64_t)(tyl * plan_.tiles_x + tx));
                        }
                    }
            list_longest_ = std::max<uint32_t>(list_longest_, (uint32_t)list.size() - list_start_[k]);
        }
        list_start_[8] = (uint32_t)list.size();
        if (list.empty()) return WV_OK;
        uint64_t* staged = nullptr;  // (a list that did not arrive whole must never be launched with)
        WV_HIP(hipMalloc((void**)&staged, list.size() * sizeof(uint64_t)));
        if (hipMemcpy(staged, list.data(), list.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(staged);
            return fail(WV_E_HIP, "copying the tile work list to the device failed");
        }
        tile_list_ = staged;
        return WV_OK;
    }

    // `out`: where the new field goes (null: in place, over `prev`)
    int launch_stream(Real* prev, const Real* cur, int* flag, int z0, int z1, bool timed, Real* out = nullptr) {
        if (z0 >= z1) return WV_OK;
        wv::StreamArgs<Real> a{};
        a.prev = prev;
        a.next = out ? out : prev;
        a.cur = cur;
        a.cls = cls_;
        a.flag = flag;
        a.nx = nx_;
        a.ny = ny_;
        a.nz = nz_;
        a.pitch = pitch_;
        a.cls_pitch = cls_pitch_;
        a.z_begin = z0;
        a.z_end = z1;
        a.tiles_x = plan_.tiles_x;
        a.tiles_y = plan_.tiles_y;
        unsigned grid = plan_.grid;
        if (plan_.variant == 2 || plan_.variant == 3) {
            a.stripe_rows = plan_.stripe_rows;
            a.tiles_y_stripe = plan_.tiles_y_stripe;
            a.passes = plan_.passes;
            grid = 8u * (unsigned)plan_.passes * (unsigned)(z1 - z0) * (unsigned)(a.tiles_x * a.tiles_y_stripe);
            // rooms that leave much of the mesh outside: visit only the tiles with something to
            // update -- valid while the outside nodes hold zeros in both fields (outside_dirty_)
            // (built for the engine's big launch: all owned planes, or the interior planes of a slab)
            if ((int64_t)(z1 - z0) * 2 > (int64_t)(z_end_ - z_begin_) && outside_dirty_ == 0) {
                int rc = build_tile_lists(z0, z1);
                if (rc) return rc;
                if (tile_list_ && z0 == lists_z0_ && z1 == lists_z1_) {
                    a.tile_list = tile_list_;
                    for (int k = 0; k < 9; ++k) a.list_start[k] = list_start_[k];
                    grid = 8u * list_longest_;
                }
            }
        } else if (plan_.variant == 0) {
            a.zc = std::min(plan_.zc, z1 - z0);
            a.chunks_z = (z1 - z0 + a.zc - 1) / a.zc;
            a.total_tiles = a.tiles_x * a.tiles_y * a.chunks_z;
            a.tiles_per_xcd = (a.total_tiles + 7) / 8;
            grid = (unsigned)a.tiles_per_xcd * 8u;
        }
        timed = timed && time_this_launch();
        if (timed) WV_HIP(hipEventRecord(events_[ev_used_], stream_));
        if (plan_.variant == 1) {
            hipLaunchKernelGGL(wv::stream_naive_kernel<Real>, dim3(grid), dim3(plan_.block), 0, stream_, a);
        } else if (plan_.ry == 2) {
            launch_ry<2>(a, grid);
        } else {
            launch_ry<4>(a, grid);
        }
        if (timed) {
            WV_HIP(hipEventRecord(events_[ev_used_ + 1], stream_));
            ev_used_ += 2;
            timed_steps_ += 1;
        }
        return WV_OK;
    }

    wv::BoundaryArgs<Real> boundary_args(Real* prev, const Real* cur, int* flag) const {
        wv::BoundaryArgs<Real> b{};
        b.prev = prev;
        b.next = prev;  // one step at a time: the next field replaces `previous` in place
        b.cur = cur;
        b.flag = flag;
        b.bnode = bnode_;
        b.btype = btype_;
        b.fmem = fmem_;
        b.cidx = cidx_;
        b.coeffs = coeffs_;
        b.n_coeffs = n_coeffs_;
        b.n1 = n1_;
        b.n2 = n2_;
        b.n3 = n3_;
        b.n_slots = n_slots_;
        b.nx = nx_;
        b.ny = ny_;
        b.nz = nz_;
        b.pitch = pitch_;
        b.z_begin = z_begin_;
        b.z_end = z_end_;
        b.courant = courant_;
        b.courant_sq = courant_sq_;
        return b;
    }

    // Boundary nodes of planes [z0, z1).  MUST be enqueued after the streaming launch that covers
    // those planes (the sweep writes boundary nodes' old values back, see X_STORE_ALL).
    // `out` (two-step passes): the new values go to another field instead of replacing `prev`.
    int launch_boundary(Real* prev, const Real* cur, int* flag, int z0, int z1,
                        const wv::PrePostArgs<Real>* next = nullptr, Real* out = nullptr, bool fix_inner = false) {
        if (!n_entries_ || z0 >= z1) return WV_OK;
        wv::BoundaryArgs<Real> b = boundary_args(prev, cur, flag);
        if (out) b.next = out;
        b.fix_z0 = z0;  // (fix_inner: second launch of a two-step pass over the marched planes)
        b.fix_z1 = z1;
        wv::PrePostArgs<Real> nx{};  // fused == 0: nothing rides in this launch
        if (next) {
            nx = *next;
            nx.fused = 1;
        }
        uint32_t n = n_entries_;
        if (z0 > z_begin_ || z1 < z_end_) {
            const int rc = build_plane_order();
            if (rc != WV_OK) return rc;
            b.order = zorder_ + plane_start_[z0];
            b.n_order = plane_start_[z1] - plane_start_[z0];
            n = b.n_order;
            if (!n) return WV_OK;
        }
        const bool lds = n_coeffs_ <= wv::kMaxLdsCoefficientSets && env_int("WV_BOUNDARY_LDS", 1) != 0;
        const dim3 grid((n + 255) / 256), block(256);
        if (lds && fix_inner)
            hipLaunchKernelGGL((wv::boundary_kernel<Real, true, true>), grid, block, 0, stream_, b, nx);
        else if (lds)
            hipLaunchKernelGGL((wv::boundary_kernel<Real, true, false>), grid, block, 0, stream_, b, nx);
        else if (fix_inner)
            hipLaunchKernelGGL((wv::boundary_kernel<Real, false, true>), grid, block, 0, stream_, b, nx);
        else
            hipLaunchKernelGGL((wv::boundary_kernel<Real, false, false>), grid, block, 0, stream_, b, nx);
        return WV_OK;
    }

    // One loop body: [pre/post on device] + pressure update + boundary update; flag -> flags_[slot]
    // reset a step's flag word to the mesh-static bits (setup_validate_kernel), inject the source
    // sample into `cur`, gather the receivers from it
    wv::PrePostArgs<Real> pre_post_args(Real* cur, int slot, bool with_pre_post, uint64_t signal_pos, bool source_live) const {
        const bool io = with_pre_post && (n_recv_ || source_live);
        wv::PrePostArgs<Real> pp{};
        pp.cur = cur;
        pp.signal = signal_;
        pp.signal_pos = signal_pos;
        pp.signal_base = graph_capturing_ ? signal_base_dev_ : nullptr;
        pp.source_node = source_node_;
        pp.source_kind = io && source_live ? source_kind_ : 0;
        pp.recv = recv_nodes_;
        pp.recv_out = recv_out_ + (size_t)slot * std::max<uint32_t>(n_recv_, 1);
        pp.n_recv = io ? n_recv_ : 0;
        pp.flag = flags_ + slot;
        pp.flag_init = static_flag_;
        return pp;
    }

    // `fuse_next` (1: a single step follows in this batch, 2: a two-step pass): what follows gets its pre/post
    // work done by this step's boundary launch instead of a launch of its own -- one launch less per
    // step, which is what small meshes are bound by.
    int enqueue_step(int slot, bool with_pre_post, uint64_t signal_pos, bool source_live, int fuse_next = 0) {
        Real* prev = field_[prv_];
        Real* cur = field_[cur_];
        int* flag = flags_ + slot;
        int rc;
        std::string cerr;
        // ghost planes of `cur` come from the exchange issued at the end of the previous step
        if (comm_ && !comm_->wait_ghosts(stream_, &cerr)) return fail(WV_E_COMM, cerr);
        if (!pre_post_done_) {
            const wv::PrePostArgs<Real> pp = pre_post_args(cur, slot, with_pre_post, signal_pos, source_live);
            hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
        }
        pre_post_done_ = false;
        // Order on the one compute stream: a plane's sweep, then that plane's boundary nodes.
        if (comm_) {
            // slab faces first, so that their exchange overlaps the interior update
            const int lo = opt_.ghost_lo ? 1 : 0, hi = opt_.ghost_hi ? 1 : 0;
            const int zi0 = std::min(z_begin_ + lo, z_end_), zi1 = std::max(z_end_ - hi, zi0);
            if ((rc = launch_stream(prev, cur, flag, z_begin_, zi0, false))) return rc;
            if ((rc = launch_stream(prev, cur, flag, zi1, z_end_, false))) return rc;
            if ((rc = launch_boundary(prev, cur, flag, z_begin_, zi0))) return rc;
            if ((rc = launch_boundary(prev, cur, flag, zi1, z_end_))) return rc;
            WV_HIP(hipGetLastError());
            if (!comm_->exchange_faces(stream_, prv_, &cerr)) return fail(WV_E_COMM, cerr);
            if ((rc = launch_stream(prev, cur, flag, zi0, zi1, true))) return rc;
            if ((rc = launch_boundary(prev, cur, flag, zi0, zi1))) return rc;
        } else {
            if ((rc = launch_stream(prev, cur, flag, z_begin_, z_end_, true))) return rc;
            if (fuse_next && n_entries_) {
                // the next step's `current` is this step's `prev`
                wv::PrePostArgs<Real> nx = pre_post_args(prev, slot + 1, true, signal_pos + 1, source_live);
                if (fuse_next == 2) nx.flag2 = flags_ + slot + 2;  // a two-step pass follows: both its flag words
                if ((rc = launch_boundary(prev, cur, flag, z_begin_, z_end_, &nx))) return rc;
                pre_post_done_ = true;
            } else if ((rc = launch_boundary(prev, cur, flag, z_begin_, z_end_))) {
                return rc;
            }
        }
        WV_HIP(hipGetLastError());
        if (comm_ && !comm_->step_done(stream_, &cerr)) return fail(WV_E_COMM, cerr);
        // every plane has been through a full sweep once more: outside nodes of `prev` are 0 now
        if (outside_dirty_ > 0 && outside_dirty_ < (1 << 30)) --outside_dirty_;
        return WV_OK;
    }

    // ---- two steps per pass (pair_kernels.hip.h) ----------------------------------------------------
    // May this engine take two-step passes right now?  Needs: the product sweep on a whole, unsliced
    // mesh whose rows fit one workgroup, outside nodes known to hold zeros, and (unless forced) a
    // mesh big enough to be bound by HBM bytes rather than by launches or the Infinity Cache --
    // two more fields are allocated the first time (288 GB of HBM: 4 x 8.6 GB at 1024^3).
    bool pair_eligible() {
        constexpr int WX = 64 * (16 / (int)sizeof(Real));
        if (pair_mode_ == 0 || pair_failed_) return false;
        if ((opt_.ghost_lo || opt_.ghost_hi) && (!comm_ || z_end_ - z_begin_ < 4)) return false;
        // (outside nodes a caller wrote to are zeroed by two single full sweeps first: batch_pairs_ready)
        // (rows of more than kPairMaxWaves waves are shared by several workgroups: the WIDE march, up to 50 waves)
        const int max_waves = env_int("WV_PAIR_WIDE", 1) ? wv::kPairMaxWindows * (wv::kPairMaxWaves - 2) + 2 : wv::kPairMaxWaves;
        if (plan_.variant != 2 || pitch_ > max_waves * WX || outside_dirty_ > 2) return false;
        if (pair_mode_ < 0) {
            // Measured (profiles/r02/pair_vs_single_small_meshes.txt), fp64, Gnode-updates/s single / two-step:
            // 96^3 55 / 35, 128^3 96 / 72 (launches, not bytes), 160^3 86 / 101, 192^3 115 / 134, 256^3 191 / 205,
            // 288^3 136 / 183, 384^3 210 / 253, 512^3 220 / 264-282, 768^3 180 / 282-296, 1024^3 236-243 / 317-334.
            // (Until the fix-up launch and the two source / receiver launches of a pass went -- three launches per
            // pass now -- single steps held out up to 256^3.)  fp32: half the bytes for the same arithmetic; the
            // march was bound by its instruction stream (383 vs 444 at 1024^3) until div3: 532-558 vs 452.
            if (stored_nodes_ < pair_min_nodes_) return false;
        }
        return true;
    }

    // spare fields, the pair map and the fix-up list for the current source node
    int ensure_pair() {
        const uint64_t src = source_kind_ != WV_SOURCE_NONE ? source_node_ : ~0ull;
        if (pair_map_ && pair_source_ == src) return WV_OK;
        for (int i = 0; i < 2; ++i) {
            Real*& f = field_[spare_[i]];
            if (!f) {
                if (hipMalloc((void**)&f, field_bytes_ + 256) != hipSuccess) {
                    (void)hipGetLastError();
                    f = nullptr;
                    pair_failed_ = true;  // not enough memory for four fields: stay with single steps
                    return WV_OK;
                }
                WV_HIP(hipMemsetAsync(f, 0, field_bytes_ + 256, stream_));
            }
        }
        if (comm_) {  // the exchange has to know the two new fields
            void* fields[4] = {field_[0], field_[1], field_[2], field_[3]};
            comm_->set_fields(fields, 4, (size_t)pitch_ * ny_ * sizeof(Real), nz_);
        }
        const uint64_t cls_bytes = (uint64_t)cls_pitch_ * 4u * (uint64_t)((ny_ + 3) / 4) * nz_;
        if (!pair_map_) {
            WV_HIP(hipMalloc((void**)&pair_map_, cls_bytes + 16));
            WV_HIP(hipMemsetAsync(pair_map_, 0, cls_bytes + 16, stream_));
        }
        if (!pair_counter_) WV_HIP(hipMalloc((void**)&pair_counter_, 3 * sizeof(uint32_t)));
        wv::PairMapArgs m{};
        m.cls = cls_;
        m.pair_map = pair_map_;
        m.counter = pair_counter_;
        m.source_node = src;
        m.nx = nx_;
        m.ny = ny_;
        m.nz = nz_;
        m.pitch = pitch_;
        m.cls_pitch = cls_pitch_;
        m.z_begin = z_begin_;
        m.z_end = z_end_;
        // a slab's face planes are not marched: their t+2 needs the neighbour's t+1 face (enqueue_pair)
        pair_z0_ = z_begin_ + (opt_.ghost_lo ? 1 : 0);
        pair_z1_ = z_end_ - (opt_.ghost_hi ? 1 : 0);
        m.march_begin = pair_z0_;
        m.march_end = pair_z1_;
        // may boundary entries finish the inside nodes they face?  (once per mesh)
        if (pair_inner_ok_ < 0) {
            pair_inner_ok_ = 0;
            if (n_entries_ && env_int("WV_PAIR_INNER_FIX", 1) != 0) {
                wv::PairInnerCheckArgs c{};
                c.bnode = bnode_;
                c.btype = btype_;
                c.cls = cls_;
                c.n_entries = n_entries_;
                c.nx = nx_;
                c.ny = ny_;
                c.nz = nz_;
                c.pitch = pitch_;
                c.cls_pitch = cls_pitch_;
                c.violated = reinterpret_cast<int*>(pair_counter_);
                int violated = 0;
                WV_HIP(hipMemsetAsync(pair_counter_, 0, sizeof(uint32_t), stream_));
                hipLaunchKernelGGL(wv::pair_inner_check_kernel, dim3((n_entries_ + 255) / 256), dim3(256), 0, stream_, c);
                WV_HIP(hipMemcpyAsync(&violated, pair_counter_, sizeof(int), hipMemcpyDeviceToHost, stream_));
                WV_HIP(hipStreamSynchronize(stream_));
                pair_inner_ok_ = violated ? 0 : 1;
            }
        }
        m.cover = pair_inner_ok_;
        const int64_t n_bytes = (int64_t)cls_pitch_ * ny_ * nz_;
        const unsigned grid = (unsigned)((n_bytes + 255) / 256);
        uint32_t count[3] = {0, 0, 0};
        WV_HIP(hipMemsetAsync(pair_counter_, 0, 3 * sizeof(uint32_t), stream_));
        hipLaunchKernelGGL(wv::pair_map_kernel, dim3(grid), dim3(256), 0, stream_, m);  // count
        WV_HIP(hipMemcpyAsync(count, pair_counter_, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
        WV_HIP(hipStreamSynchronize(stream_));
        // a short list none of whose nodes has a boundary node for a neighbour (typically: the source node's
        // neighbours) can be served by the workgroup that puts the t+1 source sample in place (enqueue_pair_a)
        pair_list_early_ok_ = count[2] == 0 && count[0] <= 2048;
        if (pair_list_) {
            (void)hipFree(pair_list_);
            pair_list_ = nullptr;
        }
        pair_list_n_ = count[0];
        pair_face_n_ = count[1];
        const uint32_t total = count[0] + count[1];
        if (total) {
            // one allocation: [marched planes' nodes][face planes' nodes]
            WV_HIP(hipMalloc((void**)&pair_list_, (size_t)total * sizeof(uint32_t)));
            m.list = pair_list_;
            m.list_face = pair_list_ + count[0];
            WV_HIP(hipMemsetAsync(pair_counter_, 0, 3 * sizeof(uint32_t), stream_));
            hipLaunchKernelGGL(wv::pair_map_kernel, dim3(grid), dim3(256), 0, stream_, m);  // fill
            WV_HIP(hipGetLastError());
            // processing order: 64 x 8 x 8 bricks like the boundary entries (init), so that a wave's
            // neighbour reads share cache lines; the values do not depend on the order
            std::vector<uint32_t> list(total);
            WV_HIP(hipMemcpyAsync(list.data(), pair_list_, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
            WV_HIP(hipStreamSynchronize(stream_));
            const uint64_t bricks_x = ((uint64_t)pitch_ + 63) / 64, bricks_y = ((uint64_t)ny_ + 7) / 8;
            for (int part = 0; part < 2; ++part) {
                const uint32_t first = part ? count[0] : 0u, n = count[part];
                std::vector<uint64_t> keyed(n);
                for (uint32_t i = 0; i < n; ++i) {
                    const uint64_t idx = list[first + i];
                    const uint64_t x = idx % (uint32_t)pitch_, q = idx / (uint32_t)pitch_;
                    const uint64_t y = q % (uint32_t)ny_, z = q / (uint32_t)ny_;
                    const uint64_t brick = ((z >> 3) * bricks_y + (y >> 3)) * bricks_x + (x >> 6);
                    keyed[i] = (((brick << 12) | ((z & 7) << 9) | ((y & 7) << 6) | (x & 63)) << 32) | idx;  // brick < 2^20
                }
                parallel_sort(keyed);
                for (uint32_t i = 0; i < n; ++i) list[first + i] = (uint32_t)keyed[i];
            }
            WV_HIP(hipMemcpy(pair_list_, list.data(), (size_t)total * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        pair_source_ = src;
        // march geometry: strips of 4 rows, all planes unless there are too few strips to fill the chip
        constexpr int WX = 64 * (16 / (int)sizeof(Real));
        pair_nw_ = pitch_ / WX;
        pair_windows_ = 0;
        if (pair_nw_ > wv::kPairMaxWaves) {
            // windows of up to kPairMaxWaves waves, one halo wave on every interior side (pair_march_kernel<.., WIDE>)
            const int row_waves = pair_nw_;
            int at = 0, widest = 0;
            while (at < row_waves && pair_windows_ < wv::kPairMaxWindows) {
                const int lo_halo = at > 0 ? 1 : 0;
                int end = at + wv::kPairMaxWaves - lo_halo;            // storing [at, end) with no halo above ...
                if (end < row_waves) end -= 1;                          // ... or one wave less and a halo wave
                end = std::min(end, row_waves);
                const int first = at - lo_halo, count = end + (end < row_waves ? 1 : 0) - first;
                pair_win_[0][pair_windows_] = (uint8_t)first;
                pair_win_[1][pair_windows_] = (uint8_t)count;
                pair_win_[2][pair_windows_] = (uint8_t)at;
                pair_win_[3][pair_windows_] = (uint8_t)end;
                widest = std::max(widest, count);
                ++pair_windows_;
                at = end;
            }
            if (at < row_waves) return fail(WV_E_STATE, "row too long for the two-step pass");  // (pair_eligible rules it out)
            pair_nw_ = widest;  // waves per workgroup
        }
        pair_strips_ = (ny_ + wv::kPairRows - 1) / wv::kPairRows;
        const int owned = pair_z1_ - pair_z0_;
        // Workgroups the chip holds at once: 256 CUs x (8 wave slots at 2 waves / SIMD) / waves per
        // workgroup.  Chunks along z are chosen so that the workgroups fill whole rounds of that, weighed
        // against the three planes every chunk recomputes or loads before its first output plane.
        const int64_t slots = 256ll * std::max(1, wv::kPairMaxWaves / pair_nw_);
        int chunks = env_int("WV_PAIR_CHUNKS", 0);
        if (chunks <= 0) {
            double best = 0;
            for (int c = 1; c <= std::max(1, owned / 8) && c <= 256; ++c) {
                const int64_t wgs = (int64_t)pair_strips_ * c;
                const int64_t rounds = (wgs + slots - 1) / slots;
                const double zc = (double)((owned + c - 1) / c);
                const double cost = (double)(rounds * slots) / (double)wgs * (zc + 3.0) / zc;
                if (chunks <= 0 || cost < best - 1e-9) {
                    best = cost;
                    chunks = c;
                }
            }
        }
        chunks = std::max(1, std::min(chunks, std::max(1, owned / 8)));
        pair_zc_ = (owned + chunks - 1) / chunks;
        pair_chunks_ = (owned + pair_zc_ - 1) / pair_zc_;
        return build_pair_units(owned);
    }

    // Rooms that leave much of the mesh outside: a unit of the march (a strip of 4 rows through one chunk
    // of planes) without a single node to update produces nothing but the zeros its outputs already hold,
    // so only the other units are launched -- each XCD a run of neighbouring strips with about the same
    // number of units.  (A mesh that is nearly all room keeps the arithmetic mapping.)
    int build_pair_units(int owned) {
        if (pair_units_) {
            (void)hipFree(pair_units_);
            pair_units_ = nullptr;
        }
        pair_sparse_ok_ = true;
        if (env_int("WV_TILE_LISTS", opt_.all_tiles ? 0 : 1) == 0 || pair_strips_ >= (1 << 16) || pair_windows_) return WV_OK;
        // activity per (plane, strip)
        const int64_t n_cells = (int64_t)nz_ * pair_strips_;
        ScopedDevice act_mem;
        WV_HIP(hipMalloc(&act_mem.p, (size_t)n_cells));
        wv::TileActivityArgs t{};
        t.cls = cls_;
        t.active = static_cast<uint8_t*>(act_mem.p);
        t.ny = ny_;
        t.nz = nz_;
        t.pitch = pitch_;
        t.cls_pitch = cls_pitch_;
        t.tile_rows = wv::kPairRows;
        t.tile_cols = pitch_;
        t.tiles_x = 1;
        t.tiles_y = pair_strips_;
        hipLaunchKernelGGL(wv::tile_activity_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, stream_, t);
        WV_HIP(hipGetLastError());
        std::vector<uint8_t> active((size_t)n_cells);
        WV_HIP(hipMemcpyAsync(active.data(), act_mem.p, (size_t)n_cells, hipMemcpyDeviceToHost, stream_));
        WV_HIP(hipStreamSynchronize(stream_));
        uint64_t live = 0;
        for (int z = pair_z0_; z < pair_z1_; ++z)
            for (int sidx = 0; sidx < pair_strips_; ++sidx) live += active[(size_t)z * pair_strips_ + sidx];
        if (live * 100 >= (uint64_t)owned * pair_strips_ * 92) return WV_OK;  // (nearly) all room
        // finer chunks than a full mesh would take: skipping works in whole units
        const int zc = std::max(8, std::min(pair_zc_, env_int("WV_PAIR_UNIT_PLANES", 32)));
        const int chunks = (owned + zc - 1) / zc;
        if (chunks >= (1 << 9)) return WV_OK;  // (9 bits of a list entry)
        // Which waves of a row does a unit need?  Those between the first and the last column block that holds anything
        // but `none` nodes in the unit's rows +- a strip and planes +- 2 (all it reads, produces or hands on): beyond
        // them every field is zero, which is what a missing neighbour counts as (pair_march_kernel, unit lists).
        std::vector<uint8_t> raw;
        pair_unit_waves_ = false;
        if (env_int("WV_PAIR_UNIT_WAVES", 1) != 0 && pair_nw_ > 1) {
            ScopedDevice raw_mem;
            WV_HIP(hipMalloc(&raw_mem.p, (size_t)n_cells));
            wv::WaveActivityArgs w{};
            w.cls = cls_;
            w.raw = static_cast<uint8_t*>(raw_mem.p);
            w.ny = ny_;
            w.nz = nz_;
            w.pitch = pitch_;
            w.cls_pitch = cls_pitch_;
            w.strips = pair_strips_;
            w.nw = pair_nw_;
            w.wave_cols = 64 * (16 / (int)sizeof(Real));
            hipLaunchKernelGGL(wv::pair_wave_activity_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, stream_, w);
            WV_HIP(hipGetLastError());
            raw.resize((size_t)n_cells);
            WV_HIP(hipMemcpyAsync(raw.data(), raw_mem.p, (size_t)n_cells, hipMemcpyDeviceToHost, stream_));
            WV_HIP(hipStreamSynchronize(stream_));
            pair_unit_waves_ = true;
        }
        std::vector<std::vector<uint32_t>> of_strip((size_t)pair_strips_);
        uint64_t total = 0, live_waves = 0;
        for (int sidx = 0; sidx < pair_strips_; ++sidx)
            for (int c = 0; c < chunks; ++c) {
                bool any = false;
                const int zb = pair_z0_ + c * zc, ze = std::min(pair_z0_ + (c + 1) * zc, pair_z1_);
                for (int z = zb; z < ze && !any; ++z) any = active[(size_t)z * pair_strips_ + sidx] != 0;
                if (!any) continue;
                uint32_t entry = (uint32_t)sidx | ((uint32_t)c << 16), span = (uint32_t)pair_nw_;
                if (pair_unit_waves_) {
                    uint32_t bits = 0;
                    for (int z = std::max(0, zb - 2); z < std::min(nz_, ze + 2); ++z)
                        for (int ss = std::max(0, sidx - 1); ss <= std::min(pair_strips_ - 1, sidx + 1); ++ss)
                            bits |= raw[(size_t)z * pair_strips_ + ss];
                    const uint32_t lo = (uint32_t)__builtin_ctz(bits | (1u << 31)), hi = 32u - (uint32_t)__builtin_clz(bits | 1u);
                    span = hi > lo ? hi - lo : 1u;
                    entry |= (std::min(lo, (uint32_t)pair_nw_ - 1) << 25) | ((span - 1) << 28);
                }
                of_strip[(size_t)sidx].push_back(entry);
                live_waves += span;
                ++total;
            }
        if (!total) return WV_OK;
        // Is the march still the better deal here?  It visits whole rows (strip x chunk units) and moves 32 B per
        // node for two steps; the sweep visits 128 x 16 x 1 tiles and moves 48 B.  Sphere inscribed in 768^3: 80 % of
        // the units against 55 % of the tiles are live, and the two run level (1.59-1.73 vs 1.63 ms per step).
        (void)build_tile_lists(z_begin_, z_end_);
        // (with the live waves of a unit only, what the march moves goes by waves, not by units)
        const double unit_frac = (double)live_waves / ((double)pair_strips_ * chunks * pair_nw_);
        pair_sparse_ok_ = unit_frac * 32.0 * 1.15 < tile_active_frac_ * 48.0;
        std::vector<uint32_t> list;
        list.reserve((size_t)total);
        pair_units_longest_ = 0;
        int sidx = 0;
        for (int k = 0; k < 8; ++k) {
            pair_unit_start_[k] = (uint32_t)list.size();
            const uint64_t want = total * (uint64_t)(k + 1) / 8;  // cumulative share of XCDs 0 .. k
            while (sidx < pair_strips_ && (list.size() < want || k == 7)) {
                list.insert(list.end(), of_strip[(size_t)sidx].begin(), of_strip[(size_t)sidx].end());
                ++sidx;
            }
            pair_units_longest_ = std::max<uint32_t>(pair_units_longest_, (uint32_t)list.size() - pair_unit_start_[k]);
        }
        pair_unit_start_[8] = (uint32_t)list.size();
        pair_zc_ = zc;
        pair_chunks_ = chunks;
        uint32_t* staged = nullptr;
        WV_HIP(hipMalloc((void**)&staged, list.size() * sizeof(uint32_t)));
        if (hipMemcpy(staged, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(staged);
            return fail(WV_E_HIP, "copying the march's unit list to the device failed");
        }
        pair_units_ = staged;
        return WV_OK;
    }

    static void parallel_sort(std::vector<uint64_t>& v) {
        const size_t n = v.size();
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        if (n < (1u << 16) || hw < 2) {
            std::sort(v.begin(), v.end());
            return;
        }
        std::vector<size_t> cut(hw + 1);
        for (unsigned t = 0; t <= hw; ++t) cut[t] = n * t / hw;
        std::vector<std::thread> workers;
        for (unsigned t = 0; t < hw; ++t) workers.emplace_back([&, t] { std::sort(v.begin() + cut[t], v.begin() + cut[t + 1]); });
        for (auto& w : workers) w.join();
        for (unsigned width = 1; width < hw; width *= 2)
            for (unsigned t = 0; t + width < hw; t += 2 * width)
                std::inplace_merge(v.begin() + cut[t], v.begin() + cut[t + width], v.begin() + cut[std::min(hw, t + 2 * width)]);
    }

    // Steps `slot` and `slot + 1` of a batch in one pass: fields (prv_, cur_) = (t-1, t) in, the spare
    // fields receive t+1 and t+2 and become (previous, current).
    //
    // On a slab the two time levels each need the neighbours' face planes, so a pass has two exchanges:
    //   part A  face planes to t+1 (sweep + boundary nodes, out of place) -> exchange #1 of the t+1 field
    //           -> march over the planes in between (t+1 and t+2) + their boundary nodes to t+1,
    //           overlapping the exchange
    //   part B  ghosts of t+1 landed -> source / receivers on t+1 -> face planes to t+2 (fix-up list of all
    //           their nodes + boundary nodes) -> exchange #2 of the t+2 field -> the other fix-up nodes and
    //           boundary nodes to t+2, overlapping it.
    // A chain inside one process (wv_run_group) enqueues part A of every slab before part B of any.
    // `fuse_mid`: the source / receiver work of step t+1 (and a short fix-up list) rides in the t+1 boundary launch
    int enqueue_pair_a(int slot, uint64_t signal_pos, bool source_live, bool fuse_mid) {
        Real* A = field_[prv_];
        Real* B = field_[cur_];
        Real* O1 = field_[spare_[0]];
        Real* O2 = field_[spare_[1]];
        int* flag1 = flags_ + slot;
        int* flag2 = flags_ + slot + 1;
        int rc;
        std::string cerr;
        if (comm_ && !comm_->wait_ghosts(stream_, &cerr)) return fail(WV_E_COMM, cerr);
        if (!pre_post_done_) {  // step t: flag words of both steps, source sample into t, receivers from t
            wv::PrePostArgs<Real> pp = pre_post_args(B, slot, true, signal_pos, source_live);
            pp.flag2 = flag2;
            hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
        }
        pre_post_done_ = false;  // (else: the boundary launch before this pass has done it)
        if (comm_) {
            if ((rc = launch_stream(A, B, flag1, z_begin_, pair_z0_, false, O1))) return rc;
            if ((rc = launch_stream(A, B, flag1, pair_z1_, z_end_, false, O1))) return rc;
            if ((rc = launch_boundary(A, B, flag1, z_begin_, pair_z0_, nullptr, O1))) return rc;
            if ((rc = launch_boundary(A, B, flag1, pair_z1_, z_end_, nullptr, O1))) return rc;
            WV_HIP(hipGetLastError());
            if (!comm_->exchange_faces(stream_, spare_[0], &cerr)) return fail(WV_E_COMM, cerr);
        }
        wv::PairArgs<Real> a{};
        a.prev = A;
        a.cur = B;
        a.out1 = O1;
        a.out2 = O2;
        a.pair_map = pair_map_;
        a.flag1 = flag1;
        a.flag2 = flag2;
        a.ny = ny_;
        a.nz = nz_;
        a.pitch = pitch_;
        a.cls_pitch = cls_pitch_;
        a.z_begin = pair_z0_;
        a.z_end = pair_z1_;
        a.nw = pair_nw_;
        a.zc = pair_zc_;
        a.chunks = pair_chunks_;
        a.strips = pair_strips_;
        a.strips_per_xcd = (pair_strips_ + 7) / 8;
        unsigned grid = 8u * (unsigned)a.strips_per_xcd * (unsigned)pair_chunks_;
        if (pair_units_) {
            a.unit_list = pair_units_;
            for (int k = 0; k < 9; ++k) a.list_start[k] = pair_unit_start_[k];
            grid = 8u * pair_units_longest_;
        }
        const bool timed = time_this_launch();
        if (timed) WV_HIP(hipEventRecord(events_[ev_used_], stream_));
        // (a variant with the row length as a compile-time constant, NWC, was worth 6 % until the divide sequence went
        // (div3); at the memory ceiling it runs level with this one: tools/pair_tune.hip still prices it)
        if (pair_units_ && pair_unit_waves_) {  // rooms narrower than their rows: the live waves of each unit only
            hipLaunchKernelGGL((wv::pair_march_kernel<Real, 0, 0, true>), dim3(grid), dim3(64u * (unsigned)pair_nw_), 0, stream_, a);
        } else if (pair_windows_) {
            a.windows = pair_windows_;
            for (int k = 0; k < pair_windows_; ++k) {
                a.win_first |= (uint64_t)pair_win_[0][k] << (8 * k);
                a.win_count |= (uint64_t)pair_win_[1][k] << (8 * k);
                a.win_store_lo |= (uint64_t)pair_win_[2][k] << (8 * k);
                a.win_store_hi |= (uint64_t)pair_win_[3][k] << (8 * k);
            }
            hipLaunchKernelGGL((wv::pair_march_kernel<Real, 0, 0, true>), dim3(grid * (unsigned)pair_windows_),
                               dim3(64u * (unsigned)pair_nw_), 0, stream_, a);
        } else {
            hipLaunchKernelGGL((wv::pair_march_kernel<Real, 0, 0>), dim3(grid), dim3(64u * (unsigned)pair_nw_), 0, stream_, a);
        }
        if (timed) {
            WV_HIP(hipEventRecord(events_[ev_used_ + 1], stream_));
            ev_used_ += 2;
            timed_steps_ += 2;
        }
        // boundary nodes, t+1: own old value from t-1, neighbours from t, result into the t+1 field
        pair_mid_done_ = pair_list_done_ = false;
        if (fuse_mid && n_entries_ && (n_recv_ || source_live)) {
            // ... and, by its last workgroup, step t+1's source sample / receivers (none of those nodes is a
            // boundary node: they have been final since the march) and then the few listed nodes
            wv::PrePostArgs<Real> nx = pre_post_args(O1, slot + 1, true, signal_pos + 1, source_live);
            nx.flag = nullptr;  // reset with step t's, and already written to by the march
            if (pair_list_early_ok_ && pair_list_n_) {
                nx.fix_nodes = pair_list_;
                nx.fix_n = pair_list_n_;
                nx.fix_cur = B;
                nx.fix_out2 = O2;
                nx.fix_flag = flag2;
                nx.nx = nx_;
                nx.ny = ny_;
                nx.nz = nz_;
                nx.pitch = pitch_;
                pair_list_done_ = true;
            }
            if ((rc = launch_boundary(A, B, flag1, pair_z0_, pair_z1_, &nx, O1))) return rc;
            pair_mid_done_ = true;
        } else if ((rc = launch_boundary(A, B, flag1, pair_z0_, pair_z1_, nullptr, O1))) {
            return rc;
        }
        WV_HIP(hipGetLastError());
        return WV_OK;
    }

    int launch_fixup(uint32_t first, uint32_t n, const Real* t1, const Real* cur, Real* out2, int* flag2) {
        if (!n) return WV_OK;
        wv::PairFixupArgs<Real> f{};
        f.nodes = pair_list_ + first;
        f.n = n;
        f.t1 = t1;
        f.cur = cur;
        f.out2 = out2;
        f.flag2 = flag2;
        f.nx = nx_;
        f.ny = ny_;
        f.nz = nz_;
        f.pitch = pitch_;
        hipLaunchKernelGGL(wv::pair_fixup_kernel<Real>, dim3((n + 255) / 256), dim3(256), 0, stream_, f);
        return WV_OK;
    }

    // `fuse_next` (1: a single step follows in this batch, 2: another pass): its pre/post work rides in the
    // t+2 boundary launch
    int enqueue_pair_b(int slot, uint64_t signal_pos, bool source_live, int fuse_next) {
        Real* B = field_[cur_];
        Real* O1 = field_[spare_[0]];
        Real* O2 = field_[spare_[1]];
        int* flag2 = flags_ + slot + 1;
        int rc;
        std::string cerr;
        if (comm_ && !comm_->wait_ghosts(stream_, &cerr)) return fail(WV_E_COMM, cerr);  // ghost planes of t+1
        if (!pair_mid_done_ && (n_recv_ || source_live)) {  // step t+1: source sample into t+1, receivers from it
            wv::PrePostArgs<Real> pp = pre_post_args(O1, slot + 1, true, signal_pos + 1, source_live);
            pp.flag = nullptr;  // reset in part A, and already written to by the march
            hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
        }
        if (comm_) {
            if ((rc = launch_fixup(pair_list_n_, pair_face_n_, O1, B, O2, flag2))) return rc;
            if ((rc = launch_boundary(B, O1, flag2, z_begin_, pair_z0_, nullptr, O2))) return rc;
            if ((rc = launch_boundary(B, O1, flag2, pair_z1_, z_end_, nullptr, O2))) return rc;
            WV_HIP(hipGetLastError());
            if (!comm_->exchange_faces(stream_, spare_[1], &cerr)) return fail(WV_E_COMM, cerr);
        }
        // t+2 of the nodes next to a boundary node / the source, from the complete t+1; then the boundary nodes
        // (most of them are faced by a boundary node and finished by its entry in the launch after this one)
        if (!pair_list_done_ && (rc = launch_fixup(0, pair_list_n_, O1, B, O2, flag2))) return rc;
        if (fuse_next && n_entries_ && io_nodes_unfaced()) {
            // what follows reads its source / receiver nodes from the t+2 field: none of them is written by
            // this launch (no boundary node, no node an entry finishes)
            wv::PrePostArgs<Real> nx = pre_post_args(O2, slot + 2, true, signal_pos + 2, source_live);
            if (fuse_next == 2) nx.flag2 = flags_ + slot + 3;
            if ((rc = launch_boundary(B, O1, flag2, pair_z0_, pair_z1_, &nx, O2, pair_inner_ok_ > 0))) return rc;
            pre_post_done_ = true;
        } else if ((rc = launch_boundary(B, O1, flag2, pair_z0_, pair_z1_, nullptr, O2, pair_inner_ok_ > 0))) {
            return rc;
        }
        WV_HIP(hipGetLastError());
        if (comm_ && !comm_->step_done(stream_, &cerr)) return fail(WV_E_COMM, cerr);
        // roles: (previous, current) = (t+1, t+2); the fields that held t-1 and t are the spares now
        const int a_idx = prv_, b_idx = cur_;
        prv_ = spare_[0];
        cur_ = spare_[1];
        spare_[0] = a_idx;
        spare_[1] = b_idx;
        return WV_OK;
    }

    // part 0 / 1 of the two-step pass that covers steps i and i + 1 of the batch
    int enqueue_batch_pair(uint64_t i, int part, int next_kind) override {
        DeviceGuard guard(device_);
        return part == 0 ? enqueue_pair_a((int)i, signal_pos_ + i, batch_source_live_, batch_can_fuse_)
                         : enqueue_pair_b((int)i, signal_pos_ + i, batch_source_live_, batch_can_fuse_ ? next_kind : 0);
    }

    // Would this engine take two-step passes in the batch being planned?  *singles_first = -1: no;
    // otherwise the number of single steps (full sweeps) that must come first because a caller wrote
    // into outside nodes (0, 1 or 2).  Decided per batch, and by all slabs of a chain together: they
    // must agree, or their exchanges would not pair up.
    int batch_pairs_ready(int* singles_first) override {
        DeviceGuard guard(device_);
        *singles_first = -1;
        if (!pair_eligible()) return WV_OK;
        const int rc = ensure_pair();
        if (rc) return rc;
        if (!pair_failed_ && (pair_mode_ > 0 || pair_sparse_ok_)) *singles_first = outside_dirty_;
        return WV_OK;
    }

    // Kernel timing (wv_enable_kernel_timing): a pair of events around the dominant kernel.  The two records cost
    // about 11 us of stream time (measured at 256^3: 6 % of a pass; the launches without them follow each other
    // within a microsecond), so below 512^3 only every eighth launch is timed.
    bool time_this_launch() {
        if (!timing || ev_used_ + 2 > (int)events_.size()) return false;
        const unsigned stride = stored_nodes_ < (128ull << 20) ? 8u : 1u;
        return (timing_launches_++ % stride) == 0;
    }

    int drain_timing() {
        for (int i = 0; i + 1 < ev_used_; i += 2) {
            float ms = 0;
            WV_HIP(hipEventElapsedTime(&ms, events_[i], events_[i + 1]));
            time_ms_ += ms;
            ++time_n_;
        }
        ev_used_ = 0;
        return WV_OK;
    }

    // -------------------------------------------------------------------------------------------
    int step(int32_t* flag) override {
        DeviceGuard guard(device_);
        pre_post_done_ = false;  // (a batch that failed while being enqueued may have left it set)
        int rc = enqueue_step(0, false, 0, false);
        if (rc) return rc;
        WV_HIP(hipMemcpyAsync(flags_host_, flags_, sizeof(int), hipMemcpyDeviceToHost, stream_));
        WV_HIP(hipStreamSynchronize(stream_));
        if ((rc = drain_timing())) return rc;
        if (flag) *flag = flags_host_[0];
        return WV_OK;
    }

    int swap() override {
        std::swap(cur_, prv_);
        ++steps_done;
        // a step driven from outside (wv_step / wv_swap) records no receiver samples: its row of the log is NaN,
        // so that wv_fetch_receivers keeps addressing rows by step
        if (n_recv_) recv_log_.insert(recv_log_.end(), n_recv_, std::numeric_limits<double>::quiet_NaN());
        return WV_OK;
    }

    // Capture (once per batch shape) and replay a batch of `batch` steps.
    int replay_batch(uint64_t batch, bool source_live, bool can_fuse) {
        const GraphKey key{batch, cur_, source_live, can_fuse, n_recv_, source_node_, source_kind_, (uint64_t)(uintptr_t)signal_,
                           (uint64_t)(uintptr_t)recv_nodes_, lists_built_ && tile_list_ != nullptr};
        if (!graph_exec_ || !(key == graph_key_)) {
            if (graph_exec_) {
                (void)hipGraphExecDestroy(graph_exec_);
                graph_exec_ = nullptr;
            }
            if (!signal_base_dev_) WV_HIP(hipMalloc((void**)&signal_base_dev_, sizeof(uint64_t)));
            // whatever synchronises must happen before the capture starts
            if (plan_.variant == 2 || plan_.variant == 3) {
                int rc = build_tile_lists(z_begin_, z_end_);
                if (rc) return rc;
            }
            (void)io_nodes_plain();
            const int cur_before = cur_, prv_before = prv_;
            hipGraph_t graph = nullptr;
            WV_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
            graph_capturing_ = true;
            int rc = WV_OK;
            for (uint64_t i = 0; i < batch && rc == WV_OK; ++i) {
                rc = enqueue_step((int)i, true, i, source_live, can_fuse && i + 1 < batch ? 1 : 0);
                std::swap(cur_, prv_);
            }
            graph_capturing_ = false;
            const hipError_t end = hipStreamEndCapture(stream_, &graph);
            cur_ = cur_before;
            prv_ = prv_before;
            if (rc != WV_OK) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            WV_HIP(end);
            const hipError_t inst = hipGraphInstantiate(&graph_exec_, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            WV_HIP(inst);
            graph_key_ = key;
        }
        WV_HIP(hipMemcpyAsync(signal_base_dev_, &signal_pos_, sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
        WV_HIP(hipGraphLaunch(graph_exec_, stream_));
        // batch is even: the fields are back in their roles
        return WV_OK;
    }

    // ---- a batch of steps: plan / enqueue / collect / commit ---------------------------------------
    // How many of `remaining` steps the next batch may take (0: the source signal is exhausted, which
    // ends the run -- hard_source.h:18-20 returns false).
    uint64_t plan_batch(uint64_t remaining) override {
        DeviceGuard guard(device_);
        const uint64_t interval = opt_.flag_interval > 0 ? (uint64_t)opt_.flag_interval : (uint64_t)kRing;
        uint64_t batch = std::min<uint64_t>(std::min<uint64_t>(interval, kRing), remaining);
        if (source_kind_ != WV_SOURCE_NONE) {
            const uint64_t left = signal_len_ - std::min(signal_len_, signal_pos_);
            batch = std::min(batch, left);
        }
        batch_can_fuse_ = !comm_ && io_nodes_plain() && env_int("WV_FUSE_PRE_POST", 1) != 0;
        batch_source_live_ = source_kind_ != WV_SOURCE_NONE;
        // nothing rides across batches: whatever a batch that failed half-way left behind does not count
        pre_post_done_ = pair_mid_done_ = pair_list_done_ = false;
        return batch;
    }

    // `next_kind`: what step i + 1 of the batch starts (its source / receiver work may ride in this step's
    // boundary launch)
    int enqueue_batch_step(uint64_t i, uint64_t batch, int next_kind) override {
        DeviceGuard guard(device_);
        const int rc = enqueue_step((int)i, true, signal_pos_ + i, batch_source_live_,
                                    batch_can_fuse_ && i + 1 < batch ? next_kind : 0);
        if (rc) return rc;
        std::swap(cur_, prv_);
        return WV_OK;
    }

    // Flag words and receiver rows of the batch to the host.  On an RCCL slab chain the flag words are
    // OR-ed over the ranks first, so that every rank sees the same first failing step and none is left
    // waiting in a receive (waveguide.h:100-119 stops the one and only device; here all of them stop).
    int collect_batch(uint64_t batch) override {
        DeviceGuard guard(device_);
        std::string cerr;
        if (comm_ && !comm_->or_flags(stream_, flags_, (int)batch, &cerr)) return fail(WV_E_COMM, cerr);
        WV_HIP(hipMemcpyAsync(flags_host_, flags_, batch * sizeof(int), hipMemcpyDeviceToHost, stream_));
        if (n_recv_) {
            recv_stage_.resize((size_t)batch * n_recv_);
            WV_HIP(hipMemcpyAsync(recv_stage_.data(), recv_out_, recv_stage_.size() * sizeof(Real), hipMemcpyDeviceToHost,
                                  stream_));
        }
        WV_HIP(hipStreamSynchronize(stream_));
        return drain_timing();
    }
    const int* batch_flags() const override { return flags_host_; }

    // `flags[batch]`: this engine's flag words, or their OR over a group of slabs.
    int commit_batch(uint64_t batch, const int* flags, uint64_t* good_out, int32_t* flag_out) override {
        uint64_t good = batch;
        int32_t flag = 0;
        for (uint64_t i = 0; i < batch; ++i) {
            if (flags[i]) {
                good = i;
                flag = flags[i];
                break;
            }
        }
        if (n_recv_)
            for (size_t i = 0; i < (size_t)good * n_recv_; ++i) recv_log_.push_back((double)recv_stage_[i]);
        steps_done += good;
        signal_pos_ += good;
        // fields have advanced past a failing step: like the reference after its throw, the state is
        // no longer meaningful; keep the buffer roles consistent with `good` swaps
        if (flag && good < batch && ((batch - good) & 1)) std::swap(cur_, prv_);
        *good_out = good;
        *flag_out = flag;
        return WV_OK;
    }

    int run(uint64_t n_steps, uint64_t* done, int32_t* flag_out) override {
        DeviceGuard guard(device_);
        if (comm_ && comm_->is_local() && comm_->nranks() > 1)
            return fail(WV_E_STATE, "slabs joined by wv_comm_init_local are stepped together: use wv_run_group");
        uint64_t completed = 0;
        int32_t flag = 0;
        while (completed < n_steps && flag == 0) {
            const uint64_t batch = plan_batch(n_steps - completed);
            if (batch == 0) break;
            // Small meshes are bound by launches, not bytes: a full batch of steps is captured once
            // into a hipGraph and replayed (the only thing that differs between batches, the
            // position in the source signal, comes from a device scalar).  Even batch lengths only,
            // so that the two fields are back in their roles after every replay.
            const bool use_graph = graph_mode_ != 0 && !comm_ && !timing && (batch % 2) == 0 && batch >= 16 &&
                                   stored_nodes_ <= graph_max_nodes_ && outside_dirty_ == 0;
            if (use_graph) {
                int rc = replay_batch(batch, batch_source_live_, batch_can_fuse_);
                if (rc) return rc;
            } else {
                // big meshes: two steps per pass over the fields wherever a batch has two left
                int singles_first = -1;
                int rc = batch_pairs_ready(&singles_first);
                if (rc) return rc;
                if (comm_ && !comm_->is_local()) {
                    // every rank of the chain has to take the same path: one flag word, OR-ed over the
                    // ranks -- bit 3 "some rank cannot", bits 0-1 the largest number of single steps any
                    // rank needs first (thermometer code: OR = max)
                    int word = singles_first < 0 ? 8 : (singles_first >= 2 ? 3 : singles_first);
                    std::string cerr;
                    WV_HIP(hipMemcpyAsync(flags_ + kRing, &word, sizeof(int), hipMemcpyHostToDevice, stream_));
                    if (!comm_->or_flags(stream_, flags_ + kRing, 1, &cerr)) return fail(WV_E_COMM, cerr);
                    WV_HIP(hipMemcpyAsync(&word, flags_ + kRing, sizeof(int), hipMemcpyDeviceToHost, stream_));
                    WV_HIP(hipStreamSynchronize(stream_));
                    singles_first = (word & 8) ? -1 : ((word & 2) ? 2 : (word & 1));
                }
                const bool pairs = singles_first >= 0;
                auto pair_at = [&](uint64_t i) { return pairs && i >= (uint64_t)singles_first && i + 2 <= batch; };
                auto kind_at = [&](uint64_t i) { return i >= batch ? 0 : (pair_at(i) ? 2 : 1); };
                for (uint64_t i = 0; i < batch;) {
                    if (pair_at(i)) {
                        if ((rc = enqueue_batch_pair(i, 0, 0))) return rc;
                        if ((rc = enqueue_batch_pair(i, 1, kind_at(i + 2)))) return rc;
                        i += 2;
                    } else {
                        if ((rc = enqueue_batch_step(i, batch, kind_at(i + 1)))) return rc;
                        i += 1;
                    }
                }
            }
            int rc = collect_batch(batch);
            if (rc) return rc;
            uint64_t good = 0;
            if ((rc = commit_batch(batch, flags_host_, &good, &flag))) return rc;
            completed += good;
        }
        if (done) *done = completed;
        if (flag_out) *flag_out = flag;
        return WV_OK;
    }

    // -------------------------------------------------------------------------------------------
    int set_source(int kind, uint64_t node, const double* signal, uint64_t n) override {
        DeviceGuard guard(device_);
        // validate and stage first; the engine's source changes only once nothing can fail any more
        if (kind != WV_SOURCE_NONE && kind != WV_SOURCE_HARD && kind != WV_SOURCE_SOFT)
            return fail(WV_E_INVALID_ARGUMENT, "unknown source kind");
        double* staged = nullptr;
        uint32_t cls = wv::CLS_INSIDE;
        if (kind != WV_SOURCE_NONE) {
            if (node >= n_nodes_) return fail(WV_E_INVALID_ARGUMENT, "source node outside the mesh");
            if (n && !signal) return fail(WV_E_INVALID_ARGUMENT, "signal missing");
            WV_HIP(class_of(node % (uint64_t)nx_, node / (uint64_t)nx_, &cls));
            WV_HIP(hipMalloc((void**)&staged, std::max<uint64_t>(n, 1) * sizeof(double)));
            if (n && hipMemcpy(staged, signal, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(staged);
                return fail(WV_E_HIP, "copying the source signal to the device failed");
            }
        }
        if (signal_) (void)hipFree(signal_);
        signal_ = staged;
        source_kind_ = kind;
        signal_len_ = kind == WV_SOURCE_NONE ? 0 : n;
        signal_pos_ = 0;
        io_plain_known_ = false;
        io_unfaced_known_ = false;
        if (kind == WV_SOURCE_NONE) return WV_OK;
        source_node_ = stored_index(node);
        // a source in an outside node keeps writing non-zero values there: no work lists then
        if (cls == wv::CLS_NONE) outside_dirty_ = 1 << 30;
        return WV_OK;
    }

    int set_receivers(const uint64_t* nodes, uint32_t n) override {
        DeviceGuard guard(device_);
        // validate and build the new device buffers first; the engine's state changes only when
        // nothing can fail any more (a failed call leaves the engine without receivers)
        if (n && !nodes) return fail(WV_E_INVALID_ARGUMENT, "receiver node list missing");
        for (uint32_t i = 0; i < n; ++i)
            if (nodes[i] != ~0ull && nodes[i] >= n_nodes_)
                return fail(WV_E_INVALID_ARGUMENT, "receiver node outside the mesh");
        ScopedDevice new_nodes, new_out;
        if (n) {
            std::vector<uint64_t> stored(n);
            for (uint32_t i = 0; i < n; ++i) stored[i] = nodes[i] == ~0ull ? ~0ull : stored_index(nodes[i]);
            WV_HIP(hipMalloc(&new_nodes.p, n * sizeof(uint64_t)));
            WV_HIP(hipMemcpy(new_nodes.p, stored.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
            WV_HIP(hipMalloc(&new_out.p, (size_t)kRing * n * sizeof(Real)));
        }
        WV_HIP(hipStreamSynchronize(stream_));  // nothing in flight reads the old buffers
        if (recv_nodes_) (void)hipFree(recv_nodes_);
        if (recv_out_) (void)hipFree(recv_out_);
        recv_nodes_ = static_cast<uint64_t*>(new_nodes.p);
        recv_out_ = static_cast<Real*>(new_out.p);
        new_nodes.p = new_out.p = nullptr;
        recv_log_.clear();
        recv_first_step_ = steps_done;
        n_recv_ = n;
        io_plain_known_ = false;
        io_unfaced_known_ = false;
        return WV_OK;
    }

    // true when neither the source nor any receiver sits on a boundary node: those nodes are then
    // final once a step's sweep has run, before its boundary launch (which may serve them early)
    bool io_nodes_plain() {
        if (io_plain_known_) return io_plain_;
        io_plain_known_ = true;
        io_plain_ = false;
        std::vector<uint64_t> stored;
        if (!io_nodes(&stored)) return false;
        for (uint64_t idx : stored) {
            uint32_t cls = 0;
            if (class_of(idx % (uint64_t)pitch_, idx / (uint64_t)pitch_, &cls) != hipSuccess) return false;
            if (cls == wv::CLS_BOUNDARY) return false;
        }
        io_plain_ = true;
        return true;
    }

    // stored indices of the source node and the receiver nodes; false: too many to be worth looking at one by one
    bool io_nodes(std::vector<uint64_t>* stored) {
        if (n_recv_ > 64) return false;
        if (source_kind_ != WV_SOURCE_NONE) stored->push_back(source_node_);
        if (n_recv_) {
            std::vector<uint64_t> r(n_recv_);
            if (hipMemcpy(r.data(), recv_nodes_, n_recv_ * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return false;
            for (uint64_t v : r)
                if (v != ~0ull) stored->push_back(v);
        }
        return true;
    }

    // true when, besides, no source / receiver node is an inside node faced by a boundary node: in a two-step
    // pass such a node gets its t+2 value from that node's entry in the second boundary launch, which
    // therefore cannot serve it early
    bool io_nodes_unfaced() {
        if (!io_nodes_plain()) return false;
        if (io_unfaced_known_) return io_unfaced_;
        io_unfaced_known_ = true;
        io_unfaced_ = false;
        std::vector<uint64_t> stored;
        if (!io_nodes(&stored)) return false;
        for (uint64_t idx : stored) {
            const int64_t x = (int64_t)(idx % (uint64_t)pitch_), row = (int64_t)(idx / (uint64_t)pitch_);
            const int64_t y = row % ny_, z = row / ny_;
            const int64_t nb[6][3] = {{x - 1, y, z}, {x + 1, y, z}, {x, y - 1, z}, {x, y + 1, z}, {x, y, z - 1}, {x, y, z + 1}};
            for (const auto& n : nb) {
                if (n[0] < 0 || n[0] >= pitch_ || n[1] < 0 || n[1] >= ny_ || n[2] < 0 || n[2] >= nz_) continue;
                uint32_t cls = 0;
                if (class_of((uint64_t)n[0], (uint64_t)(n[2] * ny_ + n[1]), &cls) != hipSuccess) return false;
                if (cls == wv::CLS_BOUNDARY) return false;
            }
        }
        io_unfaced_ = true;
        return true;
    }

    int fetch_receivers(uint64_t first, uint64_t n, double* dst) override {
        if (first < recv_first_step_) return fail(WV_E_INVALID_ARGUMENT, "steps before wv_set_receivers are not recorded");
        const uint64_t off = first - recv_first_step_;
        if ((off + n) * n_recv_ > recv_log_.size()) return fail(WV_E_INVALID_ARGUMENT, "steps not recorded yet");
        std::memcpy(dst, recv_log_.data() + off * n_recv_, (size_t)n * n_recv_ * sizeof(double));
        return WV_OK;
    }

    // -------------------------------------------------------------------------------------------
    Real* buffer(int which) { return which == WV_BUF_CURRENT ? field_[cur_] : field_[prv_]; }
    // class (CLS_*) of the node at (x, row) of the stored layout, read back from the class map
    hipError_t class_of(uint64_t x, uint64_t row, uint32_t* cls) {
        uint8_t byte = 0;
        const int64_t at = wv::cls_byte_index((int)x, (int)(row % (uint64_t)ny_), (int)(row / (uint64_t)ny_), ny_, cls_pitch_);
        const hipError_t rc = hipMemcpyAsync(&byte, cls_ + at, 1, hipMemcpyDeviceToHost, stream_);
        if (rc != hipSuccess) return rc;
        const hipError_t rs = hipStreamSynchronize(stream_);
        *cls = (byte >> ((x & 3) * 2)) & 3u;
        return rs;
    }
    // caller's node index (x + y*nx + z*nx*ny) -> position in the stored (row-padded) field
    uint64_t stored_index(uint64_t node) const {
        const uint64_t x = node % (uint64_t)nx_, row = node / (uint64_t)nx_;
        return row * (uint64_t)pitch_ + x;
    }

    int read_value(int buffer_id, uint64_t index, double* v) override {
        DeviceGuard guard(device_);
        if (index >= n_nodes_) return fail(WV_E_INVALID_ARGUMENT, "index outside the buffer");
        Real tmp;
        WV_HIP(hipMemcpyAsync(&tmp, buffer(buffer_id) + stored_index(index), sizeof(Real), hipMemcpyDeviceToHost, stream_));
        WV_HIP(hipStreamSynchronize(stream_));
        *v = (double)tmp;
        return WV_OK;
    }
    int write_value(int buffer_id, uint64_t index, double v) override {
        DeviceGuard guard(device_);
        if (index >= n_nodes_) return fail(WV_E_INVALID_ARGUMENT, "index outside the buffer");
        const Real tmp = (Real)v;
        if (tmp != 0 && outside_dirty_ < 2) {
            // a non-zero value in an outside node is zeroed by the next two full sweeps
            uint32_t cls = 0;
            WV_HIP(class_of(index % (uint64_t)nx_, index / (uint64_t)nx_, &cls));
            if (cls == wv::CLS_NONE) outside_dirty_ = 2;
        }
        WV_HIP(hipMemcpyAsync(buffer(buffer_id) + stored_index(index), &tmp, sizeof(Real), hipMemcpyHostToDevice, stream_));
        WV_HIP(hipStreamSynchronize(stream_));
        return WV_OK;
    }

    // host field (compact: nx per row, element type Other) <-> stored field (pitch per row, Real),
    // staged through a bounded device buffer in whole rows
    template <typename Other>
    int copy_field(Real* stored, void* host, bool to_device, int z0, int planes) {
        const int64_t row0 = (int64_t)z0 * ny_, rows_total = (int64_t)planes * ny_;
        const int64_t rows_per_chunk = std::max<int64_t>(1, (64ll << 20) / nx_);
        ScopedDevice tmp_mem;
        WV_HIP(hipMalloc(&tmp_mem.p, (size_t)std::min(rows_per_chunk, rows_total) * nx_ * sizeof(Other)));
        Other* tmp = static_cast<Other*>(tmp_mem.p);
        for (int64_t row = 0; row < rows_total; row += rows_per_chunk) {
            const int64_t rows = std::min(rows_per_chunk, rows_total - row);
            const int64_t n = rows * nx_;
            const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 65536);
            Other* h = static_cast<Other*>(host) + row * nx_;
            Real* d = stored + (row0 + row) * pitch_;
            if (to_device) {
                WV_HIP(hipMemcpyAsync(tmp, h, (size_t)n * sizeof(Other), hipMemcpyHostToDevice, stream_));
                hipLaunchKernelGGL((wv::pack_rows_kernel<Real, Other>), dim3(grid), dim3(256), 0, stream_, d, pitch_,
                                   (const Other*)tmp, nx_, nx_, rows);
            } else {
                hipLaunchKernelGGL((wv::pack_rows_kernel<Other, Real>), dim3(grid), dim3(256), 0, stream_, tmp, nx_,
                                   (const Real*)d, pitch_, nx_, rows);
                WV_HIP(hipMemcpyAsync(h, tmp, (size_t)n * sizeof(Other), hipMemcpyDeviceToHost, stream_));
            }
            WV_HIP(hipStreamSynchronize(stream_));
        }
        return WV_OK;
    }

    int read_field(int buffer_id, void* dst, int elem_size) override { return read_planes(buffer_id, 0, nz_, dst, elem_size); }
    int write_field(int buffer_id, const void* src, int elem_size) override {
        return write_planes(buffer_id, 0, nz_, src, elem_size);
    }
    int read_planes(int buffer_id, int z0, int planes, void* dst, int elem_size) override {
        DeviceGuard guard(device_);
        if (z0 < 0 || planes < 0 || (int64_t)z0 + planes > nz_) return fail(WV_E_INVALID_ARGUMENT, "plane range outside the mesh");
        if (!planes) return WV_OK;
        if (!dst) return fail(WV_E_INVALID_ARGUMENT, "null argument");
        if (elem_size == 4) return copy_field<float>(buffer(buffer_id), dst, false, z0, planes);
        if (elem_size == 8) return copy_field<double>(buffer(buffer_id), dst, false, z0, planes);
        return fail(WV_E_INVALID_ARGUMENT, "elem_size must be 4 or 8");
    }
    int write_planes(int buffer_id, int z0, int planes, const void* src, int elem_size) override {
        DeviceGuard guard(device_);
        if (z0 < 0 || planes < 0 || (int64_t)z0 + planes > nz_) return fail(WV_E_INVALID_ARGUMENT, "plane range outside the mesh");
        if (!planes) return WV_OK;
        if (!src) return fail(WV_E_INVALID_ARGUMENT, "null argument");
        outside_dirty_ = std::max(outside_dirty_, 2);  // the caller may have put anything in the outside nodes
        if (elem_size == 4) return copy_field<float>(buffer(buffer_id), const_cast<void*>(src), true, z0, planes);
        if (elem_size == 8) return copy_field<double>(buffer(buffer_id), const_cast<void*>(src), true, z0, planes);
        return fail(WV_E_INVALID_ARGUMENT, "elem_size must be 4 or 8");
    }

    int boundary_data(int dim, wv_boundary_data* host, bool to_device) override {
        DeviceGuard guard(device_);
        if (dim < 1 || dim > 3) return fail(WV_E_INVALID_ARGUMENT, "dimensionality must be 1, 2 or 3");
        const uint32_t nd = dim == 1 ? n1_ : (dim == 2 ? n2_ : n3_);
        if (!nd) return WV_OK;
        const uint32_t base = dim == 1 ? 0u : (dim == 2 ? n1_ : n1_ + 2u * n2_);
        const size_t bytes = (size_t)nd * dim * sizeof(wv_boundary_data);
        if (!host) return fail(WV_E_INVALID_ARGUMENT, "boundary data array missing");
        if (to_device) {  // same rule as wv_create: a filter must name an existing coefficient set
            for (size_t i = 0; i < (size_t)nd * dim; ++i)
                if (host[i].coefficient_index >= n_coeffs_)
                    return fail(WV_E_INVALID_MESH, "coefficient index exceeds the coefficient array length");
        }
        ScopedDevice aos_mem;
        WV_HIP(hipMalloc(&aos_mem.p, bytes));
        uint64_t* aos = static_cast<uint64_t*>(aos_mem.p);
        if (to_device) WV_HIP(hipMemcpyAsync(aos, host, bytes, hipMemcpyHostToDevice, stream_));
        wv::BoundaryDataArgs a{};
        a.fmem = fmem_;
        a.cidx = cidx_;
        a.n_slots = n_slots_;
        a.slot_base = base;
        a.n_d = nd;
        a.dim = dim;
        a.aos = aos;
        a.entry_off = dim == 1 ? 0u : (dim == 2 ? n1_ : n1_ + n2_);
        a.ref_to_pos = ref_to_pos_;
        const uint32_t n = nd * (uint32_t)dim;
        hipLaunchKernelGGL(wv::boundary_data_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, stream_, a,
                           to_device ? 1 : 0);
        if (!to_device) WV_HIP(hipMemcpyAsync(host, aos, bytes, hipMemcpyDeviceToHost, stream_));
        WV_HIP(hipStreamSynchronize(stream_));
        return WV_OK;
    }

    int set_coefficients(const wv_coefficients_canonical* c, uint32_t n) override {
        DeviceGuard guard(device_);
        if (n != n_coeffs_)
            return fail(WV_E_INVALID_ARGUMENT,
                        "Size of new coefficients vector must be equal to the existing one");  // setup.cpp:43-47
        if (n) {
            WV_HIP(hipMemcpyAsync(coeffs_, c, n * sizeof(wv_coefficients_canonical), hipMemcpyHostToDevice, stream_));
            WV_HIP(hipStreamSynchronize(stream_));
        }
        return WV_OK;
    }

    int device_buffer(int buffer_id, void** p) override {
        outside_dirty_ = 1 << 30;  // raw access: stop assuming anything about the outside nodes
        *p = buffer(buffer_id);
        return WV_OK;
    }

    int kernel_time(double* mean_ms, uint64_t* launches, uint64_t* steps) override {
        DeviceGuard guard(device_);
        if (mean_ms) *mean_ms = time_n_ ? time_ms_ / (double)time_n_ : 0.0;
        if (launches) *launches = time_n_;
        if (steps) *steps = timed_steps_;
        time_ms_ = 0;
        time_n_ = 0;
        timed_steps_ = 0;
        timing_launches_ = 0;  // the next launch is timed again
        return WV_OK;
    }

    int synchronize() override {
        DeviceGuard guard(device_);
        WV_HIP(hipStreamSynchronize(stream_));
        WV_HIP(hipStreamSynchronize(comm_stream_));
        return WV_OK;
    }

    int comm_init(const void* id, int rank, int nranks) override {
        DeviceGuard guard(device_);
        if (comm_) return fail(WV_E_STATE, "communicator already initialised");
        std::unique_ptr<wv::SlabComm> c(new wv::SlabComm());
        std::string err;
        if (!c->init(id, rank, nranks, device_, comm_stream_, opt_.ghost_lo != 0, opt_.ghost_hi != 0, &err))
            return fail(WV_E_COMM, err);
        return adopt_comm(std::move(c));
    }
    int comm_init_local(int rank, int nranks) override {
        DeviceGuard guard(device_);
        if (comm_) return fail(WV_E_STATE, "communicator already initialised");
        std::unique_ptr<wv::SlabComm> c(new wv::SlabComm());
        std::string err;
        if (!c->init_local(rank, nranks, device_, comm_stream_, opt_.ghost_lo != 0, opt_.ghost_hi != 0, &err))
            return fail(WV_E_COMM, err);
        return adopt_comm(std::move(c));
    }
    int adopt_comm(std::unique_ptr<wv::SlabComm> c) {
        void* fields[4] = {field_[0], field_[1], field_[2], field_[3]};
        c->set_fields(fields, 4, (size_t)pitch_ * ny_ * sizeof(Real), nz_);
        comm_ = std::move(c);
        return WV_OK;
    }
    wv::SlabComm* comm() override { return comm_.get(); }
    uint64_t field_pitch() const override { return (uint64_t)pitch_; }
    int comm_destroy() override {
        DeviceGuard guard(device_);
        comm_.reset();
        return WV_OK;
    }

private:
    void release() {
        DeviceGuard guard(device_);
        comm_.reset();
        if (stream_) (void)hipStreamSynchronize(stream_);
        for (auto& e : events_) (void)hipEventDestroy(e);
        events_.clear();
        for (int i = 0; i < 4; ++i)
            if (field_[i]) (void)hipFree(field_[i]);
        if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
        void* ptrs[] = {pair_units_, pair_map_, pair_list_, pair_counter_, signal_base_dev_, tile_list_, ref_to_pos_, cls_,   bnode_,      btype_,    fmem_,  cidx_,
                        status_,          coeffs_,    flags_,      scratch_, signal_, recv_nodes_, recv_out_, zorder_};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (flags_host_) (void)hipHostFree(flags_host_);
        if (stream_) (void)hipStreamDestroy(stream_);
        if (comm_stream_) (void)hipStreamDestroy(comm_stream_);
    }

    wv_options opt_{};
    int nx_ = 0, ny_ = 0, nz_ = 0, z_begin_ = 0, z_end_ = 0, device_ = -1;
    uint64_t n_nodes_ = 0, stored_nodes_ = 0, field_bytes_ = 0;
    int pitch_ = 0;
    Real* field_[4] = {nullptr, nullptr, nullptr, nullptr};
    int cur_ = 1, prv_ = 0, spare_[2] = {2, 3};  // which field_ holds which role
    uint8_t* cls_ = nullptr;
    int cls_pitch_ = 0;
    uint32_t n1_ = 0, n2_ = 0, n3_ = 0, n_entries_ = 0, n_slots_ = 0, n_coeffs_ = 0;
    uint32_t* bnode_ = nullptr;
    uint64_t* tile_list_ = nullptr;   // sweep work list (build_tile_lists), null = arithmetic mapping
    uint32_t list_start_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t list_longest_ = 0;
    bool lists_built_ = false;
    int lists_z0_ = 0, lists_z1_ = 0;  // plane range the lists were built for
    struct GraphKey {
        uint64_t batch;
        int cur;
        bool source_live, can_fuse;
        uint32_t n_recv;
        uint64_t source_node;
        int source_kind;
        uint64_t signal_ptr, recv_ptr;
        bool lists;
        bool operator==(const GraphKey& o) const {
            return batch == o.batch && cur == o.cur && source_live == o.source_live && can_fuse == o.can_fuse &&
                   n_recv == o.n_recv && source_node == o.source_node && source_kind == o.source_kind &&
                   signal_ptr == o.signal_ptr && recv_ptr == o.recv_ptr && lists == o.lists;
        }
    };
    hipGraphExec_t graph_exec_ = nullptr;
    GraphKey graph_key_{};
    uint64_t* signal_base_dev_ = nullptr;
    bool graph_capturing_ = false;
    int graph_mode_ = env_int("WV_GRAPH", 0);
    uint64_t graph_max_nodes_ = 64ull << 20;
    bool pre_post_done_ = false;      // this step's pre/post work was done by the previous boundary launch
    // two-step passes
    int pair_inner_ok_ = -1;  // boundary entries finish the inside nodes they face (ensure_pair): -1 not checked yet
    int pair_mode_ = env_int("WV_PAIR", -1);   // 1 always (where eligible), 0 never, -1 from pair_min_nodes_ up
    uint64_t pair_min_nodes_ = 4ull << 20;      // stored nodes: between 128^3 (single steps win) and 160^3 (passes win)
    bool pair_failed_ = false;
    uint8_t* pair_map_ = nullptr;
    uint32_t* pair_list_ = nullptr;
    uint32_t* pair_counter_ = nullptr;
    uint32_t* pair_units_ = nullptr;               // march work list (build_pair_units), null = every unit
    uint32_t pair_unit_start_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t pair_units_longest_ = 0;
    bool pair_sparse_ok_ = true;                   // sparse room: the march's live units cost less than the sweep's live tiles
    double tile_active_frac_ = 1.0;
    uint32_t pair_list_n_ = 0, pair_face_n_ = 0;  // fix-up nodes of the marched planes / of a slab's face planes
    int pair_z0_ = 0, pair_z1_ = 0;                // planes the march produces
    uint64_t pair_source_ = 0;
    int pair_nw_ = 1, pair_strips_ = 0, pair_zc_ = 0, pair_chunks_ = 1;
    uint64_t timed_steps_ = 0;
    bool batch_can_fuse_ = false, batch_source_live_ = false;  // plan_batch's decisions for the batch being enqueued
    bool io_plain_known_ = false, io_plain_ = false;
    bool io_unfaced_known_ = false, io_unfaced_ = false;
    bool pair_list_early_ok_ = false;             // ensure_pair
    bool pair_unit_waves_ = false;                // the unit list carries each unit's live waves (build_pair_units)
    int pair_windows_ = 0;                        // WIDE march: workgroups side by side per row (0: one)
    uint8_t pair_win_[4][wv::kPairMaxWindows] = {};  // first wave, waves, first storing wave, end of the storing waves
    bool pair_mid_done_ = false, pair_list_done_ = false;  // part A of the pass in flight has served t+1's source / receivers, the list
    int outside_dirty_ = 0;           // steps until the outside nodes are known to be 0 in both fields again
    uint32_t* ref_to_pos_ = nullptr;  // [n_entries] caller's (class offset + boundary_index) -> processing position
    uint8_t* btype_ = nullptr;
    double* fmem_ = nullptr;
    uint32_t* cidx_ = nullptr;
    // boundary entries by plane (build_plane_order; slab path only)
    uint32_t* zorder_ = nullptr;
    std::vector<uint32_t> plane_start_;
    int* status_ = nullptr;
    int* static_flag_dev_ = nullptr;
    int static_flag_ = 0;
    double* coeffs_ = nullptr;
    int* flags_ = nullptr;
    int* flags_host_ = nullptr;
    void* scratch_ = nullptr;
    Real courant_ = 0, courant_sq_ = 0;
    hipStream_t stream_ = nullptr, comm_stream_ = nullptr;
    StreamPlan plan_;
    int tune_variant_ = -1, tune_ry_ = 0, tune_nwx_ = 0, tune_nwy_ = 0, tune_zchunks_ = 0;
    std::vector<hipEvent_t> events_;
    unsigned timing_launches_ = 0;
    int ev_used_ = 0;
    double time_ms_ = 0;
    uint64_t time_n_ = 0;
    // source / receivers
    int source_kind_ = WV_SOURCE_NONE;
    uint64_t source_node_ = 0, signal_len_ = 0, signal_pos_ = 0;
    double* signal_ = nullptr;
    uint64_t* recv_nodes_ = nullptr;
    Real* recv_out_ = nullptr;
    uint32_t n_recv_ = 0;
    uint64_t recv_first_step_ = 0;
    std::vector<Real> recv_stage_;
    std::vector<double> recv_log_;
    std::unique_ptr<wv::SlabComm> comm_;
};

}  // namespace

// =============================================================================================
extern "C" {

const char* wv_last_error(void) { return g_last_error.c_str(); }

void wv_default_options(wv_options* o) {
    std::memset(o, 0, sizeof(*o));
    o->struct_size = (int32_t)sizeof(wv_options);
    o->precision = WV_PRECISION_F64;
    o->device = -1;
    o->flag_interval = 0;
    o->stream_variant = 2;
}

int wv_create(const wv_mesh* mesh, const wv_options* options, wv_engine** out) {
    if (!mesh || !out) return fail(WV_E_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    wv_options opt;
    wv_default_options(&opt);
    if (options) {
        const size_t n = std::min<size_t>(sizeof(opt), options->struct_size > 0 ? (size_t)options->struct_size : sizeof(opt));
        std::memcpy(&opt, options, n);
        opt.struct_size = (int32_t)sizeof(opt);
    }
    std::unique_ptr<wv_engine> e;
    if (opt.precision == WV_PRECISION_F32) {
        e.reset(new Engine<float>());
    } else if (opt.precision == WV_PRECISION_F64) {
        e.reset(new Engine<double>());
    } else {
        return fail(WV_E_INVALID_ARGUMENT, "unknown precision");
    }
    const int rc = e->init(*mesh, opt);
    if (rc != WV_OK) return rc;
    *out = e.release();
    return WV_OK;
}

void wv_destroy(wv_engine* e) { delete e; }

#define WV_NEED(e) \
    if (!(e)) return fail(WV_E_INVALID_ARGUMENT, "null engine")

int wv_read_value(wv_engine* e, int buffer, uint64_t index, double* value) {
    WV_NEED(e);
    return e->read_value(buffer, index, value);
}
int wv_write_value(wv_engine* e, int buffer, uint64_t index, double value) {
    WV_NEED(e);
    return e->write_value(buffer, index, value);
}
int wv_read_field(wv_engine* e, int buffer, void* dst, int elem_size) {
    WV_NEED(e);
    return e->read_field(buffer, dst, elem_size);
}
int wv_write_field(wv_engine* e, int buffer, const void* src, int elem_size) {
    WV_NEED(e);
    return e->write_field(buffer, src, elem_size);
}
int wv_read_planes(wv_engine* e, int buffer, int32_t z_begin, int32_t z_count, void* dst, int elem_size) {
    WV_NEED(e);
    return e->read_planes(buffer, z_begin, z_count, dst, elem_size);
}
int wv_write_planes(wv_engine* e, int buffer, int32_t z_begin, int32_t z_count, const void* src, int elem_size) {
    WV_NEED(e);
    return e->write_planes(buffer, z_begin, z_count, src, elem_size);
}
int wv_read_boundary_data(wv_engine* e, int dim, wv_boundary_data* dst) {
    WV_NEED(e);
    return e->boundary_data(dim, dst, false);
}
int wv_write_boundary_data(wv_engine* e, int dim, const wv_boundary_data* src) {
    WV_NEED(e);
    return e->boundary_data(dim, const_cast<wv_boundary_data*>(src), true);
}
int wv_set_coefficients(wv_engine* e, const wv_coefficients_canonical* c, uint32_t n) {
    WV_NEED(e);
    return e->set_coefficients(c, n);
}
int wv_device_buffer(wv_engine* e, int buffer, void** p) {
    WV_NEED(e);
    return e->device_buffer(buffer, p);
}
int wv_step(wv_engine* e, int32_t* flag) {
    WV_NEED(e);
    return e->step(flag);
}
int wv_swap(wv_engine* e) {
    WV_NEED(e);
    return e->swap();
}
int wv_set_source(wv_engine* e, int kind, uint64_t node, const double* signal, uint64_t n) {
    WV_NEED(e);
    return e->set_source(kind, node, signal, n);
}
int wv_set_receivers(wv_engine* e, const uint64_t* nodes, uint32_t n) {
    WV_NEED(e);
    return e->set_receivers(nodes, n);
}
int wv_run(wv_engine* e, uint64_t n_steps, uint64_t* done, int32_t* flag) {
    WV_NEED(e);
    return e->run(n_steps, done, flag);
}
int wv_fetch_receivers(wv_engine* e, uint64_t first, uint64_t n, double* dst) {
    WV_NEED(e);
    return e->fetch_receivers(first, n, dst);
}
int wv_step_count(wv_engine* e, uint64_t* steps) {
    WV_NEED(e);
    *steps = e->steps_done;
    return WV_OK;
}
int wv_kernel_time_ms(wv_engine* e, double* mean_ms, uint64_t* launches) {
    WV_NEED(e);
    return e->kernel_time(mean_ms, launches, nullptr);
}
int wv_kernel_time_detail(wv_engine* e, double* mean_ms, uint64_t* launches, uint64_t* steps) {
    WV_NEED(e);
    return e->kernel_time(mean_ms, launches, steps);
}
int wv_enable_kernel_timing(wv_engine* e, int enable) {
    WV_NEED(e);
    e->timing = enable != 0;
    return WV_OK;
}
int wv_synchronize(wv_engine* e) {
    WV_NEED(e);
    return e->synchronize();
}
int wv_set_stream_tuning(wv_engine* e, int variant, int ry, int nwx, int nwy, int zchunks) {
    WV_NEED(e);
    return e->set_tuning(variant, ry, nwx, nwy, zchunks);
}
int wv_comm_unique_id(void* id_bytes) {
    std::string err;
    if (!wv::SlabComm::unique_id(id_bytes, &err)) return fail(WV_E_COMM, err);
    return WV_OK;
}
int wv_comm_init(wv_engine* e, const void* id_bytes, int rank, int nranks) {
    WV_NEED(e);
    return e->comm_init(id_bytes, rank, nranks);
}
int wv_comm_destroy(wv_engine* e) {
    WV_NEED(e);
    return e->comm_destroy();
}

int wv_comm_init_local(wv_engine* const* engines, int32_t n) {
    if (!engines || n < 1) return fail(WV_E_INVALID_ARGUMENT, "no engines");
    for (int i = 0; i < n; ++i) WV_NEED(engines[i]);
    for (int i = 0; i < n; ++i) {
        const int rc = engines[i]->comm_init_local(i, n);
        if (rc != WV_OK) {
            for (int k = 0; k < i; ++k) (void)engines[k]->comm_destroy();
            return rc;
        }
    }
    for (int i = 0; i < n; ++i)
        engines[i]->comm()->link_local(i > 0 ? engines[i - 1]->comm() : nullptr, i + 1 < n ? engines[i + 1]->comm() : nullptr);
    return WV_OK;
}

int wv_run_group(wv_engine* const* engines, int32_t n, uint64_t n_steps, uint64_t* steps_done, int32_t* flag_out) {
    if (!engines || n < 1) return fail(WV_E_INVALID_ARGUMENT, "no engines");
    for (int i = 0; i < n; ++i) WV_NEED(engines[i]);
    uint64_t completed = 0;
    int32_t flag = 0;
    std::vector<int> ored;
    while (completed < n_steps && flag == 0) {
        // the shortest batch any slab allows (the slab that holds the source knows when it ends)
        uint64_t batch = n_steps - completed;
        for (int k = 0; k < n; ++k) batch = std::min(batch, engines[k]->plan_batch(n_steps - completed));
        if (batch == 0) break;
        // two-step passes only if every slab can take them, after the single steps any of them needs first
        int singles_first = 0;
        for (int k = 0; k < n && singles_first >= 0; ++k) {
            int mine = -1;
            const int rc = engines[k]->batch_pairs_ready(&mine);
            if (rc) return rc;
            singles_first = mine < 0 ? -1 : std::max(singles_first, mine);
        }
        const bool pairs = singles_first >= 0;
        // lockstep: step i of every slab is enqueued before step i + 1 of any, and the two parts of a
        // two-step pass likewise (comm.h, local transport)
        auto pair_at = [&](uint64_t i) { return pairs && i >= (uint64_t)singles_first && i + 2 <= batch; };
        for (uint64_t i = 0; i < batch;) {
            if (pair_at(i)) {
                for (int part = 0; part < 2; ++part)
                    for (int k = 0; k < n; ++k) {
                        const int rc = engines[k]->enqueue_batch_pair(i, part, 0);
                        if (rc) return rc;
                    }
                i += 2;
            } else {
                for (int k = 0; k < n; ++k) {
                    const int rc = engines[k]->enqueue_batch_step(i, batch, 0);
                    if (rc) return rc;
                }
                i += 1;
            }
        }
        ored.assign((size_t)batch, 0);
        for (int k = 0; k < n; ++k) {
            const int rc = engines[k]->collect_batch(batch);
            if (rc) return rc;
            const int* f = engines[k]->batch_flags();
            for (uint64_t i = 0; i < batch; ++i) ored[(size_t)i] |= f[i];
        }
        uint64_t good = 0;
        for (int k = 0; k < n; ++k) {
            const int rc = engines[k]->commit_batch(batch, ored.data(), &good, &flag);
            if (rc) return rc;
        }
        completed += good;
    }
    if (steps_done) *steps_done = completed;
    if (flag_out) *flag_out = flag;
    return WV_OK;
}

int wv_field_pitch(wv_engine* e, uint64_t* pitch_elements) {
    WV_NEED(e);
    *pitch_elements = e->field_pitch();
    return WV_OK;
}

// a[i] = b[i] + s * c[i] over `n` doubles, `iters` timed launches after one warm-up: the classic device
// triad, as the yardstick bench.py prints next to the stencil's own bandwidth.  Written
// the way this chip streams best: short-lived workgroups in address order, 16 B per
// lane, 16 KiB per workgroup and array.
__global__ void __launch_bounds__(256) triad_kernel(double* a, const double* b, const double* c, double s, int64_t n2) {
    typedef double V2 __attribute__((ext_vector_type(2)));
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    V2 x[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + j * 256;
        x[j] = i < n2 ? __builtin_nontemporal_load(reinterpret_cast<const V2*>(b) + i) : (V2)(0.0);
        y[j] = i < n2 ? __builtin_nontemporal_load(reinterpret_cast<const V2*>(c) + i) : (V2)(0.0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + j * 256;
        if (i < n2) __builtin_nontemporal_store(x[j] + s * y[j], reinterpret_cast<V2*>(a) + i);
    }
}

int wv_measure_triad(int32_t device, uint64_t n_doubles, int32_t iters, double* gb_per_s) {
    if (!gb_per_s || n_doubles < 2 || iters < 1) return fail(WV_E_INVALID_ARGUMENT, "bad triad arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    DeviceGuard guard(device);
    ScopedDevice a, b, c;
    const size_t bytes = (size_t)n_doubles * sizeof(double);
    WV_HIP(hipMalloc(&a.p, bytes));
    WV_HIP(hipMalloc(&b.p, bytes));
    WV_HIP(hipMalloc(&c.p, bytes));
    WV_HIP(hipMemset(b.p, 0, bytes));
    WV_HIP(hipMemset(c.p, 0, bytes));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    WV_HIP(hipEventCreate(&e0));
    WV_HIP(hipEventCreate(&e1));
    const int64_t n2 = (int64_t)(n_doubles / 2);
    const unsigned grid = (unsigned)((n2 + 1023) / 1024);
    for (int it = 0; it < iters + 1; ++it) {
        if (it == 1) WV_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(triad_kernel, dim3(grid), dim3(256), 0, 0, static_cast<double*>(a.p), static_cast<const double*>(b.p),
                           static_cast<const double*>(c.p), 0.5, n2);
    }
    WV_HIP(hipEventRecord(e1, 0));
    WV_HIP(hipEventSynchronize(e1));
    float ms = 0;
    WV_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    WV_HIP(hipGetLastError());
    *gb_per_s = 3.0 * (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
    return WV_OK;
}

int wv_filter_test_2(const float* input, float* output, double* memory, const wv_coefficients_canonical* coeffs,
                     uint32_t n_filters, uint32_t n_samples) {
    if (!input || !output || !memory || !coeffs) return fail(WV_E_INVALID_ARGUMENT, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    const size_t n = n_filters, total = (size_t)n_filters * n_samples;
    ScopedDevice m_in, m_out, m_mem, m_c;
    WV_HIP(hipMalloc(&m_in.p, std::max<size_t>(total, 1) * sizeof(float)));
    WV_HIP(hipMalloc(&m_out.p, std::max<size_t>(total, 1) * sizeof(float)));
    WV_HIP(hipMalloc(&m_mem.p, std::max<size_t>(n, 1) * 6 * sizeof(double)));
    WV_HIP(hipMalloc(&m_c.p, std::max<size_t>(n, 1) * 14 * sizeof(double)));
    float *d_in = static_cast<float*>(m_in.p), *d_out = static_cast<float*>(m_out.p);
    double *d_mem = static_cast<double*>(m_mem.p), *d_c = static_cast<double*>(m_c.p);
    if (total == 0) return WV_OK;
    WV_HIP(hipMemcpy(d_in, input, total * sizeof(float), hipMemcpyHostToDevice));
    WV_HIP(hipMemcpy(d_mem, memory, n * 6 * sizeof(double), hipMemcpyHostToDevice));
    WV_HIP(hipMemcpy(d_c, coeffs, n * 14 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(wv::filter_test_2_kernel, dim3((n_filters + 63) / 64), dim3(64), 0, 0, d_in, d_out, d_mem, d_c,
                       n_filters, n_samples);
    WV_HIP(hipGetLastError());
    WV_HIP(hipMemcpy(output, d_out, total * sizeof(float), hipMemcpyDeviceToHost));
    WV_HIP(hipMemcpy(memory, d_mem, n * 6 * sizeof(double), hipMemcpyDeviceToHost));
    return WV_OK;
}

}  // extern "C"
