// engine_io.hip.h -- what callers read and write: core::read_value / write_value / read_from_buffer on the engine's handles, filter
// memories (get_boundary_data<N>, setup.h:68-85), the device-resident source and receivers.
//
// Part of the engine behind the C ABI of include/wayverb_amd.h (engine.hip is the translation unit; see engine.hip.h for
// the class and the map of which file holds what).
#pragma once
#include "engine.hip.h"

namespace wv {

// -------------------------------------------------------------------------------------------
template <typename Real>
int Engine<Real>::set_source(int kind, uint64_t node, const double* signal, uint64_t n) {
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
    io_unfaced_known_ = io_xclear_known_ = false;
    duties_known_ = false;
    ++io_generation_;
    if (kind == WV_SOURCE_NONE) return WV_OK;
    source_node_ = stored_index(node);
    // a source in an outside node keeps writing non-zero values there: no work lists then
    if (cls == wv::CLS_NONE) outside_dirty_ = 1 << 30;
    return WV_OK;
}

template <typename Real>
int Engine<Real>::set_receivers(const uint64_t* nodes, uint32_t n) {
    DeviceGuard guard(device_);
    // validate and build the new device buffers first; the engine's state changes only when
    // nothing can fail any more (a failed call leaves the engine without receivers)
    if (n && !nodes) return fail(WV_E_INVALID_ARGUMENT, "receiver node list missing");
    for (uint32_t i = 0; i < n; ++i)
        if (nodes[i] != ~0ull && nodes[i] >= n_nodes_)
            return fail(WV_E_INVALID_ARGUMENT, "receiver node outside the mesh");
    ScopedDevice new_nodes, new_out;
    Real* new_stage = nullptr;
    if (n) {
        std::vector<uint64_t> stored(n);
        for (uint32_t i = 0; i < n; ++i) stored[i] = nodes[i] == ~0ull ? ~0ull : stored_index(nodes[i]);
        WV_HIP(hipMalloc(&new_nodes.p, n * sizeof(uint64_t)));
        WV_HIP(hipMemcpy(new_nodes.p, stored.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
        WV_HIP(hipMalloc(&new_out.p, (size_t)kRing * n * sizeof(Real)));
        WV_HIP(hipHostMalloc((void**)&new_stage, (size_t)kRing * n * sizeof(Real), hipHostMallocDefault));
    }
    WV_HIP(hipStreamSynchronize(stream_));  // nothing in flight reads the old buffers
    if (recv_nodes_) (void)hipFree(recv_nodes_);
    if (recv_out_) (void)hipFree(recv_out_);
    if (recv_stage_) (void)hipHostFree(recv_stage_);
    recv_stage_ = new_stage;
    recv_nodes_ = static_cast<uint64_t*>(new_nodes.p);
    recv_out_ = static_cast<Real*>(new_out.p);
    new_nodes.p = new_out.p = nullptr;
    recv_log_.clear();
    recv_first_step_ = steps_done;
    n_recv_ = n;
    io_plain_known_ = false;
    io_unfaced_known_ = io_xclear_known_ = false;
    duties_known_ = false;
    ++io_generation_;
    return WV_OK;
}

// true when neither the source nor any receiver sits on a boundary node: those nodes are then
// final once a step's sweep has run, before its boundary launch (which may serve them early)
template <typename Real>
bool Engine<Real>::io_nodes_plain() {
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
template <typename Real>
bool Engine<Real>::io_nodes(std::vector<uint64_t>* stored) {
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
template <typename Real>
bool Engine<Real>::io_nodes_unfaced() {
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

// true when, besides, none of them lies within two nodes of a boundary node along x: in a three-step pass the entries of x-facing walls
// finish the node they face and the one behind it at t+3 (xwall3_node), in the launch the next step's source / receiver work would ride in
template <typename Real>
bool Engine<Real>::io_nodes_clear_of_x_walls() {
    if (!io_nodes_unfaced()) return false;
    if (io_xclear_known_) return io_xclear_;
    io_xclear_known_ = true;
    io_xclear_ = false;
    std::vector<uint64_t> stored;
    if (!io_nodes(&stored)) return false;
    for (uint64_t idx : stored) {
        const int64_t x = (int64_t)(idx % (uint64_t)pitch_), row = (int64_t)(idx / (uint64_t)pitch_);
        for (int64_t dx = -2; dx <= 2; ++dx) {
            if (dx == 0 || x + dx < 0 || x + dx >= pitch_) continue;
            uint32_t cls = 0;
            if (class_of((uint64_t)(x + dx), (uint64_t)row, &cls) != hipSuccess) return false;
            if (cls == wv::CLS_BOUNDARY) return false;
        }
    }
    io_xclear_ = true;
    return true;
}

template <typename Real>
int Engine<Real>::fetch_receivers(uint64_t first, uint64_t n, double* dst) {
    if (first < recv_first_step_) return fail(WV_E_INVALID_ARGUMENT, "steps before wv_set_receivers are not recorded");
    const uint64_t off = first - recv_first_step_;
    if ((off + n) * n_recv_ > recv_log_.size()) return fail(WV_E_INVALID_ARGUMENT, "steps not recorded yet");
    std::memcpy(dst, recv_log_.data() + off * n_recv_, (size_t)n * n_recv_ * sizeof(double));
    return WV_OK;
}

// class (CLS_*) of the node at (x, row) of the stored layout, read back from the class map
template <typename Real>
hipError_t Engine<Real>::class_of(uint64_t x, uint64_t row, uint32_t* cls) {
    uint8_t byte = 0;
    const int64_t at = wv::cls_byte_index((int)x, (int)(row % (uint64_t)ny_), (int)(row / (uint64_t)ny_), ny_, cls_pitch_);
    const hipError_t rc = hipMemcpyAsync(&byte, cls_ + at, 1, hipMemcpyDeviceToHost, stream_);
    if (rc != hipSuccess) return rc;
    const hipError_t rs = hipStreamSynchronize(stream_);
    *cls = (byte >> ((x & 3) * 2)) & 3u;
    return rs;
}

// caller's node index (x + y*nx + z*nx*ny) -> position in the stored (row-padded) field
template <typename Real>
uint64_t Engine<Real>::stored_index(uint64_t node) const {
    const uint64_t x = node % (uint64_t)nx_, row = node / (uint64_t)nx_;
    return row * (uint64_t)pitch_ + x;
}

template <typename Real>
int Engine<Real>::read_value(int buffer_id, uint64_t index, double* v) {
    DeviceGuard guard(device_);
    if (index >= n_nodes_) return fail(WV_E_INVALID_ARGUMENT, "index outside the buffer");
    Real tmp;
    WV_HIP(hipMemcpyAsync(&tmp, buffer(buffer_id) + stored_index(index), sizeof(Real), hipMemcpyDeviceToHost, stream_));
    WV_HIP(hipStreamSynchronize(stream_));
    *v = (double)tmp;
    return WV_OK;
}

template <typename Real>
int Engine<Real>::write_value(int buffer_id, uint64_t index, double v) {
    DeviceGuard guard(device_);
    if (index >= n_nodes_) return fail(WV_E_INVALID_ARGUMENT, "index outside the buffer");
    const Real tmp = (Real)v;
    if (tmp != 0 && outside_dirty_ < 2) {
        // a non-zero value in an outside node is zeroed by the next two full sweeps
        uint32_t cls = 0;
        WV_HIP(class_of(index % (uint64_t)nx_, index / (uint64_t)nx_, &cls));
        if (cls == wv::CLS_NONE) outside_dirty_ = 2;
    }
    xw_valid_ = false;
    WV_HIP(hipMemcpyAsync(buffer(buffer_id) + stored_index(index), &tmp, sizeof(Real), hipMemcpyHostToDevice, stream_));
    WV_HIP(hipStreamSynchronize(stream_));
    return WV_OK;
}

// host field (compact: nx per row, element type Other) <-> stored field (pitch per row, Real),
// staged through a bounded device buffer in whole rows
template <typename Real>
template <typename Other>
int Engine<Real>::copy_field(Real* stored, void* host, bool to_device, int z0, int planes) {
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

template <typename Real>
int Engine<Real>::read_planes(int buffer_id, int z0, int planes, void* dst, int elem_size) {
    DeviceGuard guard(device_);
    if (z0 < 0 || planes < 0 || (int64_t)z0 + planes > nz_) return fail(WV_E_INVALID_ARGUMENT, "plane range outside the mesh");
    if (!planes) return WV_OK;
    if (!dst) return fail(WV_E_INVALID_ARGUMENT, "null argument");
    if (elem_size == 4) return copy_field<float>(buffer(buffer_id), dst, false, z0, planes);
    if (elem_size == 8) return copy_field<double>(buffer(buffer_id), dst, false, z0, planes);
    return fail(WV_E_INVALID_ARGUMENT, "elem_size must be 4 or 8");
}

template <typename Real>
int Engine<Real>::write_planes(int buffer_id, int z0, int planes, const void* src, int elem_size) {
    DeviceGuard guard(device_);
    if (z0 < 0 || planes < 0 || (int64_t)z0 + planes > nz_) return fail(WV_E_INVALID_ARGUMENT, "plane range outside the mesh");
    if (!planes) return WV_OK;
    if (!src) return fail(WV_E_INVALID_ARGUMENT, "null argument");
    outside_dirty_ = std::max(outside_dirty_, 2);  // the caller may have put anything in the outside nodes
    xw_valid_ = false;
    if (elem_size == 4) return copy_field<float>(buffer(buffer_id), const_cast<void*>(src), true, z0, planes);
    if (elem_size == 8) return copy_field<double>(buffer(buffer_id), const_cast<void*>(src), true, z0, planes);
    return fail(WV_E_INVALID_ARGUMENT, "elem_size must be 4 or 8");
}

template <typename Real>
int Engine<Real>::boundary_data(int dim, wv_boundary_data* host, bool to_device) {
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

template <typename Real>
int Engine<Real>::set_coefficients(const wv_coefficients_canonical* c, uint32_t n) {
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

template <typename Real>
int Engine<Real>::device_buffer(int buffer_id, void** p) {
    outside_dirty_ = 1 << 30;  // raw access: stop assuming anything about the outside nodes
    xw_valid_ = false;
    *p = buffer(buffer_id);
    return WV_OK;
}

// wv_checkpoint / wv_rollback / wv_drop_checkpoint: the state `run` carries from one loop iteration to the next (waveguide.h:80-123:
// the two pressure buffers and the boundary filter memories) copied aside on the device, and put back.  What a caller that runs
// batches of steps ahead of its per-step observers needs to hand an observer the field of a step the batch has already passed
// (include/wayverb_amd/waveguide.h, canonical_impl).
template <typename Real>
int Engine<Real>::checkpoint(int op) {
    DeviceGuard guard(device_);
    const size_t fmem_bytes = std::max<size_t>(n_slots_, 1) * 6 * sizeof(double);
    if (op == 2) {
        WV_HIP(hipStreamSynchronize(stream_));
        for (auto& f : ckpt_.field) {
            if (f) (void)hipFree(f);
            f = nullptr;
        }
        if (ckpt_.fmem) (void)hipFree(ckpt_.fmem);
        ckpt_ = Checkpoint{};
        return WV_OK;
    }
    // A slab of a chain is not alone with its state: its neighbours' ghost planes, the transport's step counters and mailbox words
    // would have to go back with it, all ranks at once.  Nobody needs that (`canonical` runs one domain): refused, not half done.
    if (comm_ && (opt_.ghost_lo || opt_.ghost_hi || comm_->nranks() > 1))
        return fail(WV_E_STATE, "wv_checkpoint / wv_rollback: not on a slab of a chain (its neighbours would not go back with it)");
    if (op == 0) {
        ckpt_.valid = false;
        for (auto& f : ckpt_.field)
            if (!f) {
                const hipError_t rc = hipMalloc((void**)&f, field_bytes_ + 256);
                if (rc != hipSuccess) {
                    f = nullptr;
                    (void)hipGetLastError();  // nothing sticky: the engine itself is untouched and goes on without a checkpoint
                    return fail(WV_E_HIP, std::string("wv_checkpoint: no room for a copy of the fields: ") + hipGetErrorString(rc));
                }
            }
        if (!ckpt_.fmem) {
            const hipError_t rc = hipMalloc((void**)&ckpt_.fmem, fmem_bytes);
            if (rc != hipSuccess) {
                ckpt_.fmem = nullptr;
                (void)hipGetLastError();
                return fail(WV_E_HIP, std::string("wv_checkpoint: no room for a copy of the filter memories: ") + hipGetErrorString(rc));
            }
        }
        WV_HIP(hipMemcpyAsync(ckpt_.field[0], field_[cur_], field_bytes_, hipMemcpyDeviceToDevice, stream_));
        WV_HIP(hipMemcpyAsync(ckpt_.field[1], field_[prv_], field_bytes_, hipMemcpyDeviceToDevice, stream_));
        WV_HIP(hipMemcpyAsync(ckpt_.fmem, fmem_, fmem_bytes, hipMemcpyDeviceToDevice, stream_));
        ckpt_.steps_done = steps_done;
        ckpt_.signal_pos = signal_pos_;
        ckpt_.recv_first_step = recv_first_step_;
        ckpt_.recv_log_size = recv_log_.size();
        ckpt_.n_recv = n_recv_;
        ckpt_.outside_dirty = outside_dirty_;
        ckpt_.valid = true;
        return WV_OK;
    }
    if (op != 1) return fail(WV_E_INVALID_ARGUMENT, "unknown checkpoint operation");
    if (!ckpt_.valid) return fail(WV_E_STATE, "wv_rollback: no checkpoint has been taken");
    if (ckpt_.recv_first_step != recv_first_step_ || ckpt_.n_recv != n_recv_ || recv_log_.size() < ckpt_.recv_log_size)
        return fail(WV_E_STATE, "wv_rollback: the receivers were changed after the checkpoint");
    if (signal_pos_ < ckpt_.signal_pos)
        return fail(WV_E_STATE, "wv_rollback: the source was changed after the checkpoint");
    WV_HIP(hipMemcpyAsync(field_[cur_], ckpt_.field[0], field_bytes_, hipMemcpyDeviceToDevice, stream_));
    WV_HIP(hipMemcpyAsync(field_[prv_], ckpt_.field[1], field_bytes_, hipMemcpyDeviceToDevice, stream_));
    WV_HIP(hipMemcpyAsync(fmem_, ckpt_.fmem, fmem_bytes, hipMemcpyDeviceToDevice, stream_));
    steps_done = ckpt_.steps_done;
    signal_pos_ = ckpt_.signal_pos;
    recv_log_.resize(ckpt_.recv_log_size);
    outside_dirty_ = std::max(outside_dirty_, ckpt_.outside_dirty);  // (steps taken since may have cleaned the outside nodes: the copies have not)
    xw_valid_ = false;  // the x-facing walls' compact copies hold the abandoned steps' values
    pre_post_done_ = pair_mid_done_ = pair_list_done_ = false;
    return WV_OK;
}

}  // namespace wv
