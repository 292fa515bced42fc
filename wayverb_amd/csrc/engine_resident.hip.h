// engine_resident.hip.h -- small meshes: a whole batch of single steps in ONE launch (resident_kernels.hip.h).  The host side: which
// units a step consists of, who waits for whom, who serves the source and the receivers; when the form is taken.
//
// Part of the engine behind the C ABI of include/wayverb_amd.h (engine.hip is the translation unit; see engine.hip.h for
// the class and the map of which file holds what).
#pragma once
#include "engine.hip.h"

namespace wv {

// May this engine step in the resident form at all?
template <typename Real>
bool Engine<Real>::resident_possible() const {
    if (opt_.tuning.resident == 0 || opt_.ghost_lo || opt_.ghost_hi) return false;
    // by default: while both fields (pad columns included: they are swept like the rest) stay in the one XCD's L2 the form lives in
    if (opt_.tuning.resident < 0 && 2 * stored_nodes_ * sizeof(Real) > resident_max_bytes_) return false;
    return stored_nodes_ <= (256ull << 20);  // (the host-side owner map is one word per stored node)
}

// ... and the batch being planned?
template <typename Real>
bool Engine<Real>::resident_now(uint64_t batch) {
    if (!resident_possible() || comm_ || batch < 4 || timing) return false;
    if (plan_.variant != 2 || plan_.ry != 4 || plan_.nwx != 1 || plan_.nwy != 4) return false;  // (the body the kernel instantiates)
    // (outside nodes a caller wrote to: every tile is visited and the masked sweep stores 0 into `none` nodes, program.cpp:485, like a
    // full sweep does)
    if (resident_failed_) return false;
    if (n_recv_ > 4096) return false;
    return true;
}

template <typename Real>
void Engine<Real>::release_resident() {
    void* ptrs[] = {res_.sweep_block, res_.dep_start, res_.dep, res_.counter, res_.io_start, res_.io, res_.args};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    res_ = ResidentTables{};
}

// The units of a step and their dependency lists (once per mesh and sweep plan); the source / receiver duties (whenever those change).
template <typename Real>
int Engine<Real>::ensure_resident() {
    constexpr int WX = 64 * (16 / (int)sizeof(Real));
    const int tile_rows = 16;  // RY 4 x NWY 4
    const int nzr = z_end_ - z_begin_;
    const int per_plane = plan_.tiles_x * plan_.tiles_y_stripe;
    const uint32_t grid = 8u * (uint32_t)plan_.passes * (uint32_t)nzr * (uint32_t)per_plane;
    const uint64_t plane = (uint64_t)pitch_ * (uint64_t)ny_;
    // block index of the sweep workgroup whose tile holds stored node (x, y, z): the inverse of stream_sweep_body's mapping
    auto sweep_block_of = [&](int x, int y, int z) -> uint32_t {
        const int stripe = y / plan_.stripe_rows;
        const int tyl = (y - stripe * plan_.stripe_rows) / tile_rows;
        const int tl = tyl * plan_.tiles_x + x / WX;
        const uint32_t j = (uint32_t)(((stripe / 8) * nzr + (z - z_begin_)) * per_plane + tl);
        return j * 8u + (uint32_t)(stripe % 8);
    };
    if (!res_.built) {
        release_resident();
        std::vector<uint32_t> unit_of_block(grid, ~0u), sweep_block;
        struct Tile {
            int x0, ya, yb, z;
        };
        std::vector<Tile> tiles;
        for (uint32_t b = 0; b < grid; ++b) {
            const int xcd = (int)(b & 7u);
            int j = (int)(b >> 3);
            const int tl = j % per_plane;
            j /= per_plane;
            const int z = z_begin_ + j % nzr;
            const int stripe = (j / nzr) * 8 + xcd;
            const int tx = tl % plan_.tiles_x, tyl = tl / plan_.tiles_x;
            const int y_lo = stripe * plan_.stripe_rows, y_hi = std::min(y_lo + plan_.stripe_rows, ny_);
            const int ya = y_lo + tyl * tile_rows;
            if (ya >= y_hi || tx * WX >= pitch_) continue;  // (an idle workgroup of the launch: no unit)
            unit_of_block[b] = (uint32_t)sweep_block.size();
            sweep_block.push_back(b);
            tiles.push_back(Tile{tx * WX, ya, std::min(ya + tile_rows, y_hi), z});
        }
        const uint32_t n_sweep = (uint32_t)sweep_block.size();
        const uint32_t n_bblocks = (n_entries_ + 255u) / 256u;
        const uint32_t n_units = n_sweep + n_bblocks;
        std::vector<uint32_t> bnode(std::max<uint32_t>(n_entries_, 1));
        if (n_entries_) WV_HIP(hipMemcpy(bnode.data(), bnode_, (size_t)n_entries_ * sizeof(uint32_t), hipMemcpyDeviceToHost));
        res_.bunit_of_node.assign((size_t)stored_nodes_, ~0u);
        for (uint32_t e = 0; e < n_entries_; ++e)
            if (bnode[e] != wv::INVALID_NODE) res_.bunit_of_node[bnode[e]] = n_sweep + e / 256u;
        auto sweep_unit_of = [&](int x, int y, int z) { return unit_of_block[sweep_block_of(x, y, z)]; };
        std::vector<uint64_t> edges;  // (u << 32 | v), both directions
        auto edge = [&](uint32_t u, uint32_t v) {
            if (u == v || u == ~0u || v == ~0u) return;
            edges.push_back((uint64_t)u << 32 | v);
            edges.push_back((uint64_t)v << 32 | u);
        };
        for (uint32_t u = 0; u < n_sweep; ++u) {  // the tiles around a tile: its halos in x, y, z
            const Tile& t = tiles[u];
            if (t.x0 > 0) edge(u, sweep_unit_of(t.x0 - 1, t.ya, t.z));
            if (t.x0 + WX < pitch_) edge(u, sweep_unit_of(t.x0 + WX, t.ya, t.z));
            if (t.ya > 0) edge(u, sweep_unit_of(t.x0, t.ya - 1, t.z));
            if (t.yb < ny_) edge(u, sweep_unit_of(t.x0, t.yb, t.z));
            if (t.z > z_begin_) edge(u, sweep_unit_of(t.x0, t.ya, t.z - 1));
            if (t.z + 1 < z_end_) edge(u, sweep_unit_of(t.x0, t.ya, t.z + 1));
        }
        for (uint32_t e = 0; e < n_entries_; ++e) {
            // a boundary node: whoever writes it or one of the six nodes around it, and whoever's tile holds any of those seven
            // positions (a tile reads every position of its rows and of the ring around them, whoever writes them)
            if (bnode[e] == wv::INVALID_NODE) continue;
            const uint32_t ub = n_sweep + e / 256u;
            const int x = (int)(bnode[e] % (uint32_t)pitch_);
            const uint32_t q = bnode[e] / (uint32_t)pitch_;
            const int y = (int)(q % (uint32_t)ny_), z = (int)(q / (uint32_t)ny_);
            const int at[7][3] = {{x, y, z}, {x - 1, y, z}, {x + 1, y, z}, {x, y - 1, z}, {x, y + 1, z}, {x, y, z - 1}, {x, y, z + 1}};
            for (const auto& p : at) {
                if (p[0] < 0 || p[0] >= pitch_ || p[1] < 0 || p[1] >= ny_ || p[2] < z_begin_ || p[2] >= z_end_) continue;
                edge(ub, sweep_unit_of(p[0], p[1], p[2]));
                edge(ub, res_.bunit_of_node[(uint64_t)p[2] * plane + (uint64_t)p[1] * (uint64_t)pitch_ + (uint64_t)p[0]]);
            }
        }
        std::sort(edges.begin(), edges.end());
        edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
        std::vector<uint32_t> dep_start(n_units + 1, 0), dep(std::max<size_t>(edges.size(), 1));
        for (uint64_t e : edges) ++dep_start[(uint32_t)(e >> 32) + 1];
        for (uint32_t u = 0; u < n_units; ++u) dep_start[u + 1] += dep_start[u];
        for (size_t i = 0; i < edges.size(); ++i) dep[i] = (uint32_t)edges[i];  // (sorted by u, then v: already in CSR order)
        res_.unit_of_block = std::move(unit_of_block);
        res_.n_sweep = n_sweep;
        res_.n_units = n_units;
        WV_HIP(hipMalloc((void**)&res_.sweep_block, std::max<size_t>(n_sweep, 1) * sizeof(uint32_t)));
        WV_HIP(hipMalloc((void**)&res_.dep_start, (size_t)(n_units + 1) * sizeof(uint32_t)));
        WV_HIP(hipMalloc((void**)&res_.dep, dep.size() * sizeof(uint32_t)));
        WV_HIP(hipMalloc((void**)&res_.io_start, (size_t)(n_units + 1) * sizeof(uint32_t)));
        WV_HIP(hipMalloc((void**)&res_.counter, (std::max<size_t>(n_units, 1) + 1) * sizeof(uint32_t)));  // (+ the participants' count)
        WV_HIP(hipMemcpy(res_.sweep_block, sweep_block.data(), (size_t)n_sweep * sizeof(uint32_t), hipMemcpyHostToDevice));
        WV_HIP(hipMemcpy(res_.dep_start, dep_start.data(), dep_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        WV_HIP(hipMemcpy(res_.dep, dep.data(), dep.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        WV_HIP(hipMemset(res_.counter, 0, (std::max<size_t>(n_units, 1) + 1) * sizeof(uint32_t)));
        res_.base = 0;
        // all workgroups of the launch must be resident at once (they wait for each other)
        int per_cu = 0, cus = 0;
        const bool lds = n_coeffs_ <= wv::kMaxLdsCoefficientSets && opt_.tuning.boundary_lds != 0;
        if (lds) WV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wv::resident_kernel<Real, true>, 256, 0));
        else WV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wv::resident_kernel<Real, false>, 256, 0));
        WV_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_));
        // ... and on ONE XCD (an eighth of the CUs): K of them; the launch has 8 K workgroups, dealt to the XCDs round-robin
        res_.grid = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((int64_t)n_units, (int64_t)std::max(1, per_cu) * (int64_t)std::max(1, cus / 8)));
        if (opt_.tuning.resident_workgroups > 0) res_.grid = std::min<uint32_t>(res_.grid, (uint32_t)opt_.tuning.resident_workgroups);
        // does a launch of 8 K workgroups put K of them on XCD 0?  (If the dispatcher deals them differently -- another partition
        // mode, another driver -- the form is simply not taken.)
        {
            uint32_t* arrived = res_.counter + std::max<size_t>(n_units, 1);
            hipLaunchKernelGGL(wv::resident_probe_kernel, dim3(8u * res_.grid), dim3(256), 0, stream_, arrived);
            uint32_t got = 0;
            WV_HIP(hipMemcpyAsync(&got, arrived, sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
            WV_HIP(hipStreamSynchronize(stream_));
            if (got != res_.grid) resident_failed_ = true;
        }
        res_.io_key_valid = false;
        res_.built = true;
    }
    // ---- who serves the source and the receivers
    std::vector<uint64_t> recv(n_recv_);
    if (n_recv_) WV_HIP(hipMemcpy(recv.data(), recv_nodes_, (size_t)n_recv_ * sizeof(uint64_t), hipMemcpyDeviceToHost));
    const uint64_t src = source_kind_ != WV_SOURCE_NONE ? source_node_ : ~0ull;
    if (!res_.io_key_valid || res_.io_source != src || res_.io_kind != source_kind_ || res_.io_recv != recv) {
        auto owner = [&](uint64_t node) -> uint32_t {
            if (res_.bunit_of_node[node] != ~0u) return res_.bunit_of_node[node];
            const int x = (int)(node % (uint64_t)pitch_);
            const uint64_t q = node / (uint64_t)pitch_;
            return res_.unit_of_block[sweep_block_of(x, (int)(q % (uint64_t)ny_), (int)(q / (uint64_t)ny_))];
        };
        std::vector<std::vector<wv::ResidentIo>> duties(res_.n_units);
        if (src != ~0ull) duties[owner(src)].push_back(wv::ResidentIo{src, 0u, (uint32_t)source_kind_});  // (first in its unit's list)
        for (uint32_t c = 0; c < n_recv_; ++c)
            if (recv[c] != ~0ull) duties[owner(recv[c])].push_back(wv::ResidentIo{recv[c], c, 0u});
        std::vector<uint32_t> io_start(res_.n_units + 1, 0);
        std::vector<wv::ResidentIo> io;
        for (uint32_t u = 0; u < res_.n_units; ++u) {
            io.insert(io.end(), duties[u].begin(), duties[u].end());
            io_start[u + 1] = (uint32_t)io.size();
        }
        if (res_.io) (void)hipFree(res_.io);
        res_.io = nullptr;
        WV_HIP(hipStreamSynchronize(stream_));
        WV_HIP(hipMalloc((void**)&res_.io, std::max<size_t>(io.size(), 1) * sizeof(wv::ResidentIo)));
        if (!io.empty()) WV_HIP(hipMemcpy(res_.io, io.data(), io.size() * sizeof(wv::ResidentIo), hipMemcpyHostToDevice));
        WV_HIP(hipMemcpy(res_.io_start, io_start.data(), io_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        res_.io_source = src;
        res_.io_kind = source_kind_;
        res_.io_recv = recv;
        res_.io_key_valid = true;
    }
    return WV_OK;
}

// `batch` loop iterations (waveguide.h:80-123) in one launch.
template <typename Real>
int Engine<Real>::resident_batch(uint64_t batch, bool source_live) {
    int rc = ensure_resident();
    if (rc) return rc;
    if (resident_failed_) return WV_E_STATE;  // (the caller takes per-step launches: nothing has been touched)
    Real* cur = field_[cur_];
    Real* prev = field_[prv_];
    // every step's flag word starts from the mesh-static bits (waveguide.h:82); receiver rows of unrecorded receivers hold 0
    WV_HIP(hipMemsetD32Async((hipDeviceptr_t)flags_, static_flag_, (size_t)batch, stream_));
    if (n_recv_) WV_HIP(hipMemsetAsync(recv_out_, 0, (size_t)batch * n_recv_ * sizeof(Real), stream_));
    WV_HIP(hipMemsetAsync(status_ + 2, 0, sizeof(int), stream_));
    // step 0's source sample and receiver row: the usual launch
    if (n_recv_ || source_live) {
        wv::PrePostArgs<Real> pp = pre_post_args(cur, 0, true, signal_pos_, source_live);
        pp.flag = nullptr;
        hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
    }
    wv::ResidentArgs<Real> r{};
    StreamLaunch sw;
    if ((rc = launch_stream(prev, cur, flags_, z_begin_, z_end_, false, nullptr, 0, 0, &sw))) return rc;
    if (sw.args.tile_list) {  // (work lists re-number the workgroups: the units stand for the arithmetic mapping)
        sw.args.tile_list = nullptr;
    }
    r.s = sw.args;
    r.b = boundary_args(prev, cur, flags_);
    r.field[0] = cur;
    r.field[1] = prev;
    r.flags = flags_;
    r.steps = (uint32_t)batch;
    r.n_sweep = res_.n_sweep;
    r.n_units = res_.n_units;
    r.sweep_block = res_.sweep_block;
    r.boundary_blocks = res_.n_units - res_.n_sweep;
    r.dep_start = res_.dep_start;
    r.dep = res_.dep;
    r.counter = res_.counter;
    r.base = res_.base;
    r.workgroups = res_.grid;
    r.arrived = res_.counter + std::max<size_t>(res_.n_units, 1);
    WV_HIP(hipMemsetAsync(r.arrived, 0, sizeof(uint32_t), stream_));
    r.io_start = res_.io_start;
    r.io = res_.io;
    r.signal = signal_;
    r.signal_pos = signal_pos_;
    r.recv_out = recv_out_;
    r.n_recv = n_recv_;
    r.gave_up = status_ + 2;
    if (source_live != (source_kind_ != WV_SOURCE_NONE)) return fail(WV_E_STATE, "resident batch: the source duty does not match the batch's plan");
    // the argument block goes to device memory (one slot per launch in flight would be needed if launches overlapped: they do not,
    // one stream)
    if (!res_.args) WV_HIP(hipMalloc(&res_.args, sizeof(wv::ResidentArgs<Real>)));
    WV_HIP(hipMemcpyAsync(res_.args, &r, sizeof(r), hipMemcpyHostToDevice, stream_));
    const auto* rp = static_cast<const wv::ResidentArgs<Real>*>(res_.args);
    const bool lds = n_coeffs_ <= wv::kMaxLdsCoefficientSets && opt_.tuning.boundary_lds != 0;
    if (lds) hipLaunchKernelGGL((wv::resident_kernel<Real, true>), dim3(8u * res_.grid), dim3(256), 0, stream_, rp);
    else hipLaunchKernelGGL((wv::resident_kernel<Real, false>), dim3(8u * res_.grid), dim3(256), 0, stream_, rp);
    WV_HIP(hipGetLastError());
    res_.base += (uint32_t)batch;
    resident_steps_ += batch;
    if (batch & 1) std::swap(cur_, prv_);
    if (outside_dirty_ > 0 && outside_dirty_ < (1 << 30)) outside_dirty_ = (int)std::max<int64_t>(0, (int64_t)outside_dirty_ - (int64_t)batch);
    xw_valid_ = false;
    return WV_OK;
}

}  // namespace wv
