// engine_single.hip.h -- one time step per pass over the fields (waveguide.h:80-123, one loop body).
//
// Part of the engine behind the C ABI of include/wayverb_amd.h (engine.hip is the translation unit; see engine.hip.h for
// the class and the map of which file holds what).
#pragma once
#include "engine.hip.h"

namespace wv {

template <typename Real>
template <int RY, int NWX, int NWY>
void Engine<Real>::launch_shape(const wv::StreamArgs<Real>& a, unsigned grid) {
    if (plan_.variant == 2) {
        hipLaunchKernelGGL((wv::stream_sweep_kernel<Real, RY, NWX, NWY>), dim3(grid), dim3(64 * NWX * NWY), 0,
                           st(), a);
    } else if (plan_.variant == 3) {
        hipLaunchKernelGGL((wv::stream_sweep_nolds_kernel<Real, RY, NWX, NWY>), dim3(grid), dim3(64 * NWX * NWY), 0,
                           st(), a);
    } else {
        hipLaunchKernelGGL((wv::stream_march_kernel<Real, RY, NWX, NWY>), dim3(grid), dim3(64 * NWX * NWY), 0,
                           st(), a);
    }
}

template <typename Real>
template <int RY>
void Engine<Real>::launch_ry(const wv::StreamArgs<Real>& a, unsigned grid) {
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

// `out`: where the new field goes (null: in place, over `prev`).  Planes [z0, z1) and, in the same launch, [zb0, zb1)
// further up (a slab's two face planes).
template <typename Real>
int Engine<Real>::launch_stream(Real* prev, const Real* cur, int* flag, int z0, int z1, bool timed, Real* out, int zb0, int zb1,
                                StreamLaunch* plan_only) {
    if (zb0 < zb1 && z0 >= z1) return launch_stream(prev, cur, flag, zb0, zb1, timed, out, 0, 0, plan_only);
    if (zb0 < zb1 && plan_.variant != 2 && plan_.variant != 3) {  // (the measurement variants take one range at a time)
        const int rc = launch_stream(prev, cur, flag, z0, z1, timed, out);
        return rc ? rc : launch_stream(prev, cur, flag, zb0, zb1, false, out);
    }
    if (z0 >= z1) return WV_OK;
    const int second = zb0 < zb1 ? zb1 - zb0 : 0;
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
    a.z_end = z1 + second;
    a.z_skip_from = second ? z1 : std::numeric_limits<int>::max();
    a.z_skip = second ? zb0 - z1 : 0;
    a.tiles_x = plan_.tiles_x;
    a.tiles_y = plan_.tiles_y;
    unsigned grid = plan_.grid;
    if (plan_.variant == 2 || plan_.variant == 3) {
        a.stripe_rows = plan_.stripe_rows;
        a.tiles_y_stripe = plan_.tiles_y_stripe;
        a.passes = plan_.passes;
        grid = 8u * (unsigned)plan_.passes * (unsigned)(z1 - z0 + second) * (unsigned)(a.tiles_x * a.tiles_y_stripe);
        // rooms that leave much of the mesh outside: visit only the tiles with something to
        // update -- valid while the outside nodes hold zeros in both fields (outside_dirty_)
        // (built for the engine's big launch: all owned planes, or the interior planes of a slab)
        if (!second && (int64_t)(z1 - z0) * 2 > (int64_t)(z_end_ - z_begin_) && outside_dirty_ == 0) {
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
    if (plan_only) {
        plan_only->args = a;
        plan_only->grid = grid;
        return WV_OK;
    }
    timed = timed && !on_ && time_this_launch();
    if (timed) WV_HIP(hipEventRecord(events_[ev_used_], stream_));
    if (plan_.variant == 1) {
        hipLaunchKernelGGL(wv::stream_naive_kernel<Real>, dim3(grid), dim3(plan_.block), 0, st(), a);
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

template <typename Real>
wv::BoundaryArgs<Real> Engine<Real>::boundary_args(Real* prev, const Real* cur, int* flag) const {
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
template <typename Real>
int Engine<Real>::launch_boundary(Real* prev, const Real* cur, int* flag, int z0, int z1, const wv::PrePostArgs<Real>* next, Real* out, bool fix_inner,
                                  bool levels, BoundaryLaunch* plan_only, int xw3) {
    const bool faces = z0 == z1 && (z0 == -1 || z0 == -2);
    if (!n_entries_ || (z0 >= z1 && !faces)) return WV_OK;
    wv::BoundaryArgs<Real> b = boundary_args(prev, cur, flag);
    if (out) b.next = out;
    b.fix_z0 = z0;  // (fix_inner: second launch of a two-step pass over the marched planes)
    b.fix_z1 = z1;
    wv::PrePostArgs<Real> nx{};  // fused == 0: nothing rides in this launch
    if (next) {
        nx = *next;
        nx.fused = 1;
    }
    // a two-step pass's launches over the bulk of the mesh: the x-facing walls by position, on their compact copies
    // (all of them lie in the planes of either such launch: xwall_eligible_kernel, engine_setup.hip.h)
    // (xw3: level 1, 2 or 3 of a three-step pass that does the same -- xwall3_node)
    const bool xw = out && xw_active_ && (levels || xw3);
    uint32_t n = n_entries_ - (xw ? n_xw_ : 0u);
    if (faces) {
        const int rc = build_plane_order();
        if (rc != WV_OK) return rc;
        b.order = z0 == -1 ? face_order_ : early_order_;
        b.n_order = n = z0 == -1 ? face_n_ : early_n_;
        if (!n) return WV_OK;
    } else if (z0 > z_begin_ || z1 < z_end_) {
        const int rc = build_plane_order();
        if (rc != WV_OK) return rc;
        const uint32_t* order = xw ? zorder_rest_ : zorder_;
        const std::vector<uint32_t>& start = xw ? plane_start_rest_ : plane_start_;
        b.order = order + start[z0];
        b.n_order = start[z1] - start[z0];
        n = b.n_order;
        if (!n && !xw) return WV_OK;
    }
    if (xw) {
        xwall_args(b);
        n += b.xw_pad;
    }
    const bool lds = n_coeffs_ <= wv::kMaxLdsCoefficientSets && opt_.tuning.boundary_lds != 0;
    if (plan_only) {
        plan_only->args = b;
        plan_only->blocks = (n + 255) / 256;
        plan_only->lds = lds;
        return WV_OK;
    }
    const dim3 grid((n + 255) / 256 + (nx.fused ? 1u : 0u)), block(256);  // (+ 1: the riding source / receiver work's own workgroup)
    if (xw && xw3) {
        if (lds && xw3 == 1)
            hipLaunchKernelGGL((wv::boundary_kernel<Real, true, false, 1>), grid, block, 0, st(), b, nx);
        else if (lds && xw3 == 2 && fix_inner)
            hipLaunchKernelGGL((wv::boundary_kernel<Real, true, true, 2>), grid, block, 0, st(), b, nx);
        else if (lds && xw3 == 3)
            hipLaunchKernelGGL((wv::boundary_kernel<Real, true, false, 3>), grid, block, 0, st(), b, nx);
        else if (xw3 == 1)
            hipLaunchKernelGGL((wv::boundary_kernel<Real, false, false, 1>), grid, block, 0, st(), b, nx);
        else if (xw3 == 2 && fix_inner)
            hipLaunchKernelGGL((wv::boundary_kernel<Real, false, true, 2>), grid, block, 0, st(), b, nx);
        else if (xw3 == 3)
            hipLaunchKernelGGL((wv::boundary_kernel<Real, false, false, 3>), grid, block, 0, st(), b, nx);
        else
            return fail(WV_E_STATE, "launch_boundary: a three-step pass's second level on compact copies without its entries finishing the nodes they face");
    } else if (lds && fix_inner)
        hipLaunchKernelGGL((wv::boundary_kernel<Real, true, true>), grid, block, 0, st(), b, nx);
    else if (lds)
        hipLaunchKernelGGL((wv::boundary_kernel<Real, true, false>), grid, block, 0, st(), b, nx);
    else if (fix_inner)
        hipLaunchKernelGGL((wv::boundary_kernel<Real, false, true>), grid, block, 0, st(), b, nx);
    else
        hipLaunchKernelGGL((wv::boundary_kernel<Real, false, false>), grid, block, 0, st(), b, nx);
    return WV_OK;
}

// A slab's face plane(s) -- the first owned plane when there is a lower neighbour, the last when there is an upper one --
// one step on: the sweep over both in ONE launch, then their boundary nodes in one.  `out`: as launch_stream.
template <typename Real>
int Engine<Real>::launch_faces(Real* prev, const Real* cur, int* flag, Real* out, int planes) {
    const int lo = opt_.ghost_lo ? planes : 0, hi = opt_.ghost_hi ? planes : 0;
    const int zi0 = std::min(z_begin_ + lo, z_end_), zi1 = std::max(z_end_ - hi, zi0);
    // One launch for both halves of these planes' step (plane_kernels.hip.h) where the sweep has the product's shape: its
    // workgroups, then the boundary entries' -- no "sweep, then boundary kernel" when the sweep leaves boundary nodes alone.
    if (opt_.tuning.fuse_planes != 0 && plan_.variant == 2 && plan_.ry == 4 && plan_.nwx == 1 && plan_.nwy == 4 && n_entries_) {
        StreamLaunch sw;
        BoundaryLaunch bd;
        int rc = launch_stream(prev, cur, flag, z_begin_, zi0, false, out, zi1, z_end_, &sw);
        if (rc) return rc;
        if ((rc = launch_boundary(prev, cur, flag, -planes, -planes, nullptr, out, false, false, &bd))) return rc;
        if (sw.grid && bd.blocks) {
            const dim3 grid(sw.grid + bd.blocks), block(256);
            if (bd.lds)
                hipLaunchKernelGGL((wv::plane_step_kernel<Real, true>), grid, block, 0, st(), sw.args, bd.args, (uint32_t)sw.grid);
            else
                hipLaunchKernelGGL((wv::plane_step_kernel<Real, false>), grid, block, 0, st(), sw.args, bd.args, (uint32_t)sw.grid);
            return WV_OK;
        }
    }
    int rc = launch_stream(prev, cur, flag, z_begin_, zi0, false, out, zi1, z_end_);
    if (rc) return rc;
    return launch_boundary(prev, cur, flag, -planes, -planes, nullptr, out);
}

// One loop body: [pre/post on device] + pressure update + boundary update; flag -> flags_[slot]
// reset a step's flag word to the mesh-static bits (setup_validate_kernel), inject the source
// sample into `cur`, gather the receivers from it
template <typename Real>
wv::PrePostArgs<Real> Engine<Real>::pre_post_args(Real* cur, int slot, bool with_pre_post, uint64_t signal_pos, bool source_live) const {
    const bool io = with_pre_post && (n_recv_ || source_live);
    wv::PrePostArgs<Real> pp{};
    pp.cur = cur;
    pp.signal = signal_;
    pp.signal_pos = signal_pos;
    pp.signal_base = graph_capturing_ ? signal_base_dev_ : nullptr;
    pp.source_node = source_node_;
    pp.source_kind = io && source_live ? source_kind_ : 0;
    pp.recv = recv_nodes_;
    pp.recv_out = recv_out_ ? recv_out_ + (size_t)slot * std::max<uint32_t>(n_recv_, 1) : nullptr;  // (no receivers: no buffer)
    pp.n_recv = io ? n_recv_ : 0;
    pp.flag = flags_ + slot;
    pp.flag_init = static_flag_;
    return pp;
}

// ---- one launch per step (small meshes) ---------------------------------------------------------------------------------------
// wv_tuning::whole_step = -1: one-launch steps while both fields sit well inside the 256 MB Infinity Cache -- up to there they beat
// two launches per step AND two-step passes (us per step, one launch / passes; fp64: 160^3 36.8 / 38.5, 192^3 46.3 / 48.4, 224^3
// 63.2 / 60.1-63.9, 256^3 79.7 / 75.3; fp32: 192^3 33.3 / 37.5, 256^3 48.6 / 52.1, 320^3, whose rows are stored 512 wide, 177 / 112:
// profiles/r05/small_mesh_one_launch_steps.txt); beyond, a step is bound by HBM bytes and the march's 16 B per node-update win.
constexpr uint64_t kWholeStepMaxBytes = 160ull << 20;  // (fp64: up to 192^3 as stored, 9.4 M nodes; fp32: 256^3)

template <typename Real>
bool Engine<Real>::whole_step_sized() const {
    return 2ull * stored_nodes_ * sizeof(Real) <= kWholeStepMaxBytes;
}

// May the engine's single steps be one launch each (whole_step_kernel)?  The sweep in the product's shape, no slab chain, and the
// source / receiver nodes (at most 64 duties) all inside or re-entrant nodes: the tile that produces such a node's value serves the
// next step's sample / receiver column from its registers.  Looks at the class map once per source / receiver set (synchronises).
template <typename Real>
bool Engine<Real>::whole_step_ready() {
    if (opt_.tuning.whole_step == 0 || comm_ || opt_.ghost_lo || opt_.ghost_hi) return false;
    if (plan_.variant != 2 || plan_.ry != 4 || plan_.nwx != 1 || plan_.nwy != 4) return false;
    if (opt_.tuning.whole_step < 0 && !whole_step_sized()) return false;
    if (duties_known_) return duties_ok_;
    duties_known_ = true;
    duties_ok_ = false;
    if (n_recv_ > 63) return false;
    std::vector<uint64_t> recv(n_recv_);
    if (n_recv_ && hipMemcpy(recv.data(), recv_nodes_, (size_t)n_recv_ * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    constexpr int WX = 64 * (16 / (int)sizeof(Real));
    const int nzr = z_end_ - z_begin_;
    const int per_plane = plan_.tiles_x * plan_.tiles_y_stripe;
    // block index of the sweep workgroup whose tile holds stored node idx: the inverse of stream_sweep_body's arithmetic mapping
    auto duty = [&](uint64_t idx, uint32_t col, uint32_t kind, wv::StepDuty* out) -> bool {
        const int x = (int)(idx % (uint64_t)pitch_);
        const uint64_t row = idx / (uint64_t)pitch_;
        const int y = (int)(row % (uint64_t)ny_), z = (int)(row / (uint64_t)ny_);
        uint32_t cls = 0;
        if (z < z_begin_ || z >= z_end_ || class_of((uint64_t)x, row, &cls) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (!(cls & 1u)) return false;  // a boundary node's value comes from a boundary workgroup, an outside node's from nobody
        const int stripe = y / plan_.stripe_rows;
        const int tl = ((y - stripe * plan_.stripe_rows) / 16) * plan_.tiles_x + x / WX;  // (tiles of RY 4 x NWY 4 rows)
        const uint32_t j = (uint32_t)(((stripe / 8) * nzr + (z - z_begin_)) * per_plane + tl);
        *out = wv::StepDuty{idx, j * 8u + (uint32_t)(stripe % 8), col, kind, 0u};
        return true;
    };
    std::vector<wv::StepDuty> list;
    wv::StepDuty q{};
    if (source_kind_ != WV_SOURCE_NONE) {
        if (!duty(source_node_, 0u, source_kind_ == WV_SOURCE_HARD ? 1u : 2u, &q)) return false;
        list.push_back(q);
    }
    for (uint32_t c = 0; c < n_recv_; ++c) {
        if (recv[c] == ~0ull) continue;  // (its column is zeroed by the launch's first workgroup)
        if (!duty(recv[c], c, 0u, &q)) return false;
        list.push_back(q);
    }
    if (!duties_ && hipMalloc((void**)&duties_, 64 * sizeof(wv::StepDuty)) != hipSuccess) {
        (void)hipGetLastError();
        duties_ = nullptr;
        return false;
    }
    if (hipStreamSynchronize(stream_) != hipSuccess) return false;  // (no launch in flight reads the old list)
    if (!list.empty() && hipMemcpy(duties_, list.data(), list.size() * sizeof(wv::StepDuty), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    n_duties_ = (uint32_t)list.size();
    duties_ok_ = true;
    return true;
}

// One step = one launch: the boundary entries' workgroups and the sweep's (masked stores, every tile: no work list) side by side.
// `serve_next`: the launch also puts step slot + 1's source sample in place, records its receiver row and resets its flag word.
template <typename Real>
int Engine<Real>::launch_whole_step(Real* prev, const Real* cur, int slot, uint64_t signal_pos, bool source_live, bool serve_next) {
    int* flag = flags_ + slot;
    StreamLaunch sw;
    BoundaryLaunch bd;
    int rc = launch_stream(prev, cur, flag, z_begin_, z_end_, false, nullptr, 0, 0, &sw);
    if (rc) return rc;
    sw.args.tile_list = nullptr;
    sw.grid = 8u * (unsigned)plan_.passes * (unsigned)(z_end_ - z_begin_) * (unsigned)(sw.args.tiles_x * sw.args.tiles_y_stripe);
    if ((rc = launch_boundary(prev, cur, flag, z_begin_, z_end_, nullptr, nullptr, false, false, &bd))) return rc;
    if (!n_entries_) bd.args = boundary_args(prev, cur, flag);
    wv::StepDuties<Real> d{};
    if (serve_next) {
        const bool has_source_duty = n_duties_ && source_kind_ != WV_SOURCE_NONE;
        d.list = duties_ + (has_source_duty && !source_live ? 1 : 0);
        d.n = n_duties_ - (has_source_duty && !source_live ? 1u : 0u);
        d.signal = signal_;
        d.signal_pos = signal_pos + 1;
        d.signal_base = graph_capturing_ ? signal_base_dev_ : nullptr;
        d.recv_out = recv_out_ ? recv_out_ + (size_t)(slot + 1) * std::max<uint32_t>(n_recv_, 1) : nullptr;
        d.recv = recv_nodes_;
        d.n_recv = n_recv_;
        d.next_flag = flags_ + slot + 1;
        d.flag_init = static_flag_;
    }
    const bool timed = !on_ && time_this_launch();
    if (timed) WV_HIP(hipEventRecord(events_[ev_used_], stream_));
    bd.blocks = (bd.blocks + 7u) & ~7u;  // (the boundary workgroups come first; the sweep's workgroup j runs on XCD j % 8: keep that)
    const dim3 grid(bd.blocks + sw.grid), block(256);
    if (bd.blocks && bd.lds)
        hipLaunchKernelGGL((wv::whole_step_kernel<Real, true>), grid, block, 0, st(), sw.args, bd.args, d, (uint32_t)bd.blocks);
    else
        hipLaunchKernelGGL((wv::whole_step_kernel<Real, false>), grid, block, 0, st(), sw.args, bd.args, d, (uint32_t)bd.blocks);
    if (timed) {
        WV_HIP(hipEventRecord(events_[ev_used_ + 1], stream_));
        ev_used_ += 2;
        timed_steps_ += 1;
    }
    ++whole_steps_;
    return WV_OK;
}

// `fuse_next` (1: a single step follows in this batch, 2: a two-step pass): what follows gets its pre/post
// work done by this step's boundary launch instead of a launch of its own -- one launch less per
// step, which is what small meshes are bound by.
template <typename Real>
int Engine<Real>::enqueue_step(int slot, bool with_pre_post, uint64_t signal_pos, bool source_live, int fuse_next) {
    Real* prev = field_[prv_];
    Real* cur = field_[cur_];
    int* flag = flags_ + slot;
    int rc;
    std::string cerr;
    // ghost planes of `cur` come from the exchange issued at the end of the previous step
    if (comm_) {
        const int token = begin_halo_wait_timing();
        if (!comm_->wait_ghosts(stream_, cur_, &cerr)) return fail(WV_E_COMM, cerr);
        if ((rc = end_halo_wait_timing(token))) return rc;
    }
    // (the flag words of a batch are reset when it is planned: a slab without source or receivers has nothing to do here)
    if (!pre_post_done_ && !(batch_flags_reset_ && !n_recv_ && !(with_pre_post && source_live))) {
        const wv::PrePostArgs<Real> pp = pre_post_args(cur, slot, with_pre_post, signal_pos, source_live);
        hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
    }
    pre_post_done_ = false;
    // Order on the one compute stream: a plane's sweep, then that plane's boundary nodes.
    if (comm_) {
        // slab faces first, so that their exchange overlaps the interior update
        const int lo = opt_.ghost_lo ? 1 : 0, hi = opt_.ghost_hi ? 1 : 0;
        const int zi0 = std::min(z_begin_ + lo, z_end_), zi1 = std::max(z_end_ - hi, zi0);
        if ((rc = launch_faces(prev, cur, flag, nullptr))) return rc;
        WV_HIP(hipGetLastError());
        if (!comm_->exchange_faces(stream_, prv_, &cerr)) return fail(WV_E_COMM, cerr);
        if (!comm_->bulk_begin(stream_, &cerr)) return fail(WV_E_COMM, cerr);  // (slabs of one device take turns at the interior sweep)
        if ((rc = launch_stream(prev, cur, flag, zi0, zi1, true))) return rc;
        if (!comm_->bulk_end(stream_, &cerr)) return fail(WV_E_COMM, cerr);
        if ((rc = launch_boundary(prev, cur, flag, zi0, zi1))) return rc;
    } else if (fuse_next != 2 && whole_step_ready()) {
        // (the duties were looked up when the batch was planned: nothing synchronises here)
        if ((rc = launch_whole_step(prev, cur, slot, signal_pos, source_live, fuse_next == 1))) return rc;
        pre_post_done_ = fuse_next == 1;
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
    xw_valid_ = false;  // (a single step moves the fields on without the x-facing walls' compact copies)
    return WV_OK;
}

// Capture (once per batch shape) and replay a batch of `batch` steps.
template <typename Real>
int Engine<Real>::replay_batch(uint64_t batch, bool source_live, bool can_fuse) {
    // (the field pointers are part of the key: a veto of two-step passes frees the spare fields and a later batch
    // allocates new ones, and after passes `cur_` / `prv_` may name any two of the four)
    const GraphKey key{batch, cur_, source_live, can_fuse, n_recv_, source_node_, source_kind_, (uint64_t)(uintptr_t)signal_,
                       (uint64_t)(uintptr_t)recv_nodes_, lists_built_ && tile_list_ != nullptr,
                       (uint64_t)(uintptr_t)field_[cur_], (uint64_t)(uintptr_t)field_[prv_], io_generation_};
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
        (void)whole_step_ready();
        const int cur_before = cur_, prv_before = prv_;
        const uint64_t whole_before = whole_steps_;
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
        graph_whole_steps_ = whole_steps_ - whole_before;  // (counted per replay, not per capture)
        whole_steps_ = whole_before;
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
    whole_steps_ += graph_whole_steps_;
    // batch is even: the fields are back in their roles
    // A replay runs no host code of enqueue_step: what that clears per step has to be cleared here -- the fields have moved
    // on without the x-facing walls' compact copies (a two-step pass after this must refill them), and every plane has
    // been through `batch` full sweeps.
    xw_valid_ = false;
    return WV_OK;
}

}  // namespace wv
