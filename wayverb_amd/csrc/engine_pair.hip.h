// engine_pair.hip.h -- two time steps per pass over the fields (pair_kernels.hip.h): when, with what lists, in which launches.
//
// Part of the engine behind the C ABI of include/wayverb_amd.h (engine.hip is the translation unit; see engine.hip.h for
// the class and the map of which file holds what).
#pragma once
#include "engine.hip.h"

namespace wv {

// ---- two steps per pass (pair_kernels.hip.h) ----------------------------------------------------
// May this engine take two-step passes right now?  Needs: the product sweep on a whole, unsliced
// mesh whose rows fit one workgroup, outside nodes known to hold zeros, and (unless forced) a
// mesh big enough to be bound by HBM bytes rather than by launches or the Infinity Cache --
// two more fields are allocated the first time (288 GB of HBM: 4 x 8.6 GB at 1024^3).
template <typename Real>
bool Engine<Real>::pair_eligible() {
    constexpr int WX = 64 * (16 / (int)sizeof(Real));
    if (opt_.tuning.pair == 0 || pair_failed_) return false;
    if ((opt_.ghost_lo || opt_.ghost_hi) && (!comm_ || z_end_ - z_begin_ < 4)) return false;
    // (outside nodes a caller wrote to are zeroed by two single full sweeps first: batch_pairs_ready)
    // (rows of more than kPairMaxWaves waves are shared by several workgroups: the WIDE march, up to 50 waves)
    const int max_waves = opt_.tuning.pair_wide ? wv::kPairMaxWindows * (wv::kPairMaxWaves - 2) + 2 : wv::kPairMaxWaves;
    if (plan_.variant != 2 || pitch_ > max_waves * WX || outside_dirty_ > 2) return false;
    // a sparse room in which the march's live units cost more than the sweep's live tiles (found by ensure_pair for
    // this very source): asked once, not before every batch
    const uint64_t src = source_kind_ != WV_SOURCE_NONE ? source_node_ : ~0ull;
    if (opt_.tuning.pair < 0 && pair_map_ && pair_source_ == src && !pair_sparse_ok_) return false;
    if (opt_.tuning.pair < 0) {
        // Measured (profiles/r02/pair_vs_single_small_meshes.txt), fp64, Gnode-updates/s single / two-step:
        // 96^3 55 / 35, 128^3 96 / 72 (launches, not bytes), 160^3 86 / 101, 192^3 115 / 134, 256^3 191 / 205,
        // 288^3 136 / 183, 384^3 210 / 253, 512^3 220 / 264-282, 768^3 180 / 282-296, 1024^3 236-243 / 317-334.
        // (Until the fix-up launch and the two source / receiver launches of a pass went -- three launches per
        // pass now -- single steps held out up to 256^3.)  fp32: half the bytes for the same arithmetic; the
        // march was bound by its instruction stream (383 vs 444 at 1024^3) until div3: 532-558 vs 452.
        if (stored_nodes_ < pair_min_nodes_) return false;
        // ... and while both fields live in the Infinity Cache, single steps of ONE launch each beat the passes (engine_single.hip.h,
        // whole_step_ready: where the source / receiver nodes allow them)
        if (opt_.tuning.whole_step != 0 && whole_step_sized() && whole_step_ready()) return false;
    }
    return true;
}

// spare fields, the pair map and the fix-up list for the current source node
template <typename Real>
int Engine<Real>::ensure_pair() {
    const uint64_t src = source_kind_ != WV_SOURCE_NONE ? source_node_ : ~0ull;
    // (the spare fields first: a veto may have given them back -- batch_pair_vetoed -- while the map stayed)
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
    if (pair_map_ && pair_source_ == src) return WV_OK;
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
    // ... and when the planes next to the faces are stepped to t+1 with them (slab_early_now), the march stores its t+1 values one
    // plane further in
    pair_s0_ = z_begin_ + (opt_.ghost_lo ? 2 : 0);
    pair_s1_ = z_end_ - (opt_.ghost_hi ? 2 : 0);
    // may boundary entries finish the inside nodes they face?  (once per mesh)
    if (pair_inner_ok_ < 0) {
        pair_inner_ok_ = 0;
        if (n_entries_ && opt_.tuning.pair_inner_fix != 0) {
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
    // (a slab's face planes get their t+2 from a plain sweep once the neighbours' t+1 faces are in -- launch_faces in
    // enqueue_pair_b -- not from a list: the map marks them unfinished, that is all)
    const uint32_t total = count[0];
    if (total) {
        WV_HIP(hipMalloc((void**)&pair_list_, (size_t)total * sizeof(uint32_t)));
        m.list = pair_list_;
        m.list_face = nullptr;
        WV_HIP(hipMemsetAsync(pair_counter_, 0, 3 * sizeof(uint32_t), stream_));
        hipLaunchKernelGGL(wv::pair_map_kernel, dim3(grid), dim3(256), 0, stream_, m);  // fill
        WV_HIP(hipGetLastError());
        // processing order: 64 x 8 x 8 bricks like the boundary entries (init), so that a wave's
        // neighbour reads share cache lines; the values do not depend on the order
        std::vector<uint32_t> list(total);
        WV_HIP(hipMemcpyAsync(list.data(), pair_list_, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
        WV_HIP(hipStreamSynchronize(stream_));
        const uint64_t bricks_x = ((uint64_t)pitch_ + 63) / 64, bricks_y = ((uint64_t)ny_ + 7) / 8;
        {
            const uint32_t first = 0u, n = count[0];
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
    // x-facing walls on compact copies: needs entries that finish the nodes they face, and a source that is neither a
    // boundary node nor within two nodes of a wall along x -- level 1 captures the t+1 values of the faced node and
    // of the node behind it before step t+1's sample goes in (boundary_kernels.hip.h, xwall_node)
    xw_active_ = false;
    if (n_xw_ && pair_inner_ok_ > 0) {
        bool source_clear = true;
        if (source_kind_ != WV_SOURCE_NONE) {
            const int64_t sx = (int64_t)(source_node_ % (uint64_t)pitch_), row = (int64_t)(source_node_ / (uint64_t)pitch_);
            for (int64_t dx = -2; dx <= 2 && source_clear; ++dx) {
                if (sx + dx < 0 || sx + dx >= pitch_) continue;
                uint32_t cls = 0;
                WV_HIP(class_of((uint64_t)(sx + dx), (uint64_t)row, &cls));
                source_clear = cls != wv::CLS_BOUNDARY;
            }
        }
        if (source_clear) {
            const int rc = build_xwall();
            if (rc) return rc;
            xw_active_ = xw_built_;
        }
    }
    if (!xw_active_) xw_valid_ = false;  // passes that do not maintain the copies leave them behind
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
    } else if (opt_.tuning.pair_split_rows != 0 && pair_nw_ >= 3) {
        // Measurement (wv_tuning::pair_split_rows): a row of 3 .. 8 waves is one workgroup per CU where 8 wave slots are free, and
        // such marches run at 3.5 TB/s.  The same row as windows of at most 4 waves (halo waves included, as in
        // the WIDE march above) puts two workgroups on a CU -- for 40-57 % more waves run than stored.
        const int row_waves = pair_nw_, cap = opt_.tuning.pair_split_rows > 1 ? opt_.tuning.pair_split_rows : 4;
        int at = 0, widest = 0;
        while (at < row_waves && pair_windows_ < wv::kPairMaxWindows) {
            const int lo_halo = at > 0 ? 1 : 0;
            int end = at + cap - lo_halo;
            if (end < row_waves) end -= 1;
            end = std::max(at + 1, std::min(end, row_waves));
            const int first = at - lo_halo, count = end + (end < row_waves ? 1 : 0) - first;
            pair_win_[0][pair_windows_] = (uint8_t)first;
            pair_win_[1][pair_windows_] = (uint8_t)count;
            pair_win_[2][pair_windows_] = (uint8_t)at;
            pair_win_[3][pair_windows_] = (uint8_t)end;
            widest = std::max(widest, count);
            ++pair_windows_;
            at = end;
        }
        if (at < row_waves) return fail(WV_E_STATE, "pair_split_rows: too many windows");
        pair_nw_ = widest;
    }
    pair_strips_ = (ny_ + wv::kPairRows - 1) / wv::kPairRows;
    const int owned = pair_z1_ - pair_z0_;
    // Workgroups the chip holds at once: 256 CUs x (8 wave slots at 2 waves / SIMD) / waves per
    // workgroup.  Chunks along z are chosen so that the workgroups fill whole rounds of that, weighed
    // against the three planes every chunk recomputes or loads before its first output plane.
    const int64_t slots = 256ll * std::max(1, wv::kPairMaxWaves / pair_nw_);
    int chunks = opt_.tuning.pair_chunks;
    if (chunks <= 0) {
        // A slab with a neighbour: the exchange of its t+1 faces is to run under the march, and whatever carries it (RCCL's
        // send / receive kernels, the runtime's copy kernels) needs a CU -- a march of ONE round holds every register of every
        // CU until all its workgroups retire together at the end, and the exchange would start after it.  At least two rounds
        // then: the first round's end is where the exchange gets in.
        // (... where that costs little: a mesh that fills two rounds only with much shorter chunks keeps the unconstrained choice)
        // (slabs of one process that share this device take turns at the march -- SlabComm::bulk_begin -- and nothing runs beside
        // it that the next slab's march would not displace anyway: one round there)
        const int64_t want_rounds = ((opt_.ghost_lo || opt_.ghost_hi) && (!comm_ || comm_->peers_elsewhere())) ? 2 : 1;
        double best[2] = {0, 0};
        int at[2] = {0, 0};  // [0] any number of rounds, [1] at least `want_rounds`
        for (int c = 1; c <= std::max(1, owned / 8) && c <= 256; ++c) {
            const int64_t wgs = (int64_t)pair_strips_ * c;
            const int64_t rounds = (wgs + slots - 1) / slots;
            const double zc = (double)((owned + c - 1) / c);
            const double cost = (double)(rounds * slots) / (double)wgs * (zc + 3.0) / zc;
            for (int k = 0; k < 2; ++k)
                if ((k == 0 || rounds >= want_rounds) && (at[k] == 0 || cost < best[k] - 1e-9)) {
                    best[k] = cost;
                    at[k] = c;
                }
        }
        chunks = (at[1] && best[1] <= 1.06 * best[0]) ? at[1] : at[0];
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
template <typename Real>
int Engine<Real>::build_pair_units(int owned) {
    if (pair_units_) {
        (void)hipFree(pair_units_);
        pair_units_ = nullptr;
    }
    pair_sparse_ok_ = true;
    if (!use_work_lists() || pair_strips_ >= (1 << 16) || pair_windows_) return WV_OK;
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
    // finer chunks than a full mesh would take: skipping works in whole units.  How many planes to a unit?  About
    // pair_unit_planes (32), and among the heights near it the one whose units fill the chip's workgroup slots in the
    // fewest, fullest rounds: an XCD runs 32 x (8 / waves per workgroup) of its units at a time, a round of them takes
    // (height + 3 prologue planes), and a last round with two units in it costs as much as a full one -- the concert
    // hall at 1 600 Hz made 1 538 units of 32 planes for 256 slots: six rounds and one nearly empty.
    const int64_t slots_per_xcd = 32ll * std::max(1, wv::kPairMaxWaves / pair_nw_);
    auto rounds_cost = [&](int height) -> double {
        const int n_chunks = (owned + height - 1) / height;
        std::vector<uint32_t> per_strip((size_t)pair_strips_, 0u);
        uint64_t units = 0;
        for (int sidx = 0; sidx < pair_strips_; ++sidx)
            for (int c = 0; c < n_chunks; ++c) {
                const int zb = pair_z0_ + c * height, ze = std::min(zb + height, pair_z1_);
                bool any = false;
                for (int z = zb; z < ze && !any; ++z) any = active[(size_t)z * pair_strips_ + sidx] != 0;
                per_strip[(size_t)sidx] += any;
                units += any;
            }
        if (!units) return 0.0;
        uint64_t longest = 0, so_far = 0, start = 0;  // the same partition into eight runs of strips as below
        int sidx = 0;
        for (int k = 0; k < 8; ++k) {
            const uint64_t want = units * (uint64_t)(k + 1) / 8;
            while (sidx < pair_strips_ && (so_far < want || k == 7)) so_far += per_strip[(size_t)sidx++];
            longest = std::max(longest, so_far - start);
            start = so_far;
        }
        return (double)((longest + slots_per_xcd - 1) / slots_per_xcd) * (double)(height + 3);
    };
    int zc = std::max(8, std::min(pair_zc_, opt_.tuning.pair_unit_planes));
    if (zc == opt_.tuning.pair_unit_planes && opt_.tuning.pair_units_by_chunk != 0) {
        double best = rounds_cost(zc);
        for (int height = zc * 3 / 4; height <= zc * 5 / 4; ++height) {
            if (height < 8 || height > owned || (owned + height - 1) / height >= (1 << 9)) continue;
            const double cost = rounds_cost(height);
            if (cost > 0 && cost < best * 0.97) {  // (only a clear win moves the height)
                best = cost;
                zc = height;
            }
        }
    }
    const int chunks = (owned + zc - 1) / zc;
    if (chunks >= (1 << 9)) return WV_OK;  // (9 bits of a list entry)
    // Which waves of a row does a unit need?  Those between the first and the last column block that holds anything
    // but `none` nodes in the unit's rows +- a strip and planes +- 2 (all it reads, produces or hands on): beyond
    // them every field is zero, which is what a missing neighbour counts as (pair_march_kernel, unit lists).
    std::vector<uint8_t> raw;
    pair_unit_waves_ = false;
    if (opt_.tuning.pair_unit_waves != 0 && pair_nw_ > 1) {
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
    pair_live_frac_ = unit_frac;
    std::vector<uint32_t> list;
    list.reserve((size_t)total);
    pair_units_longest_ = 0;
    int sidx = 0;
    for (int k = 0; k < 8; ++k) {
        pair_unit_start_[k] = (uint32_t)list.size();
        const uint64_t want = total * (uint64_t)(k + 1) / 8;  // cumulative share of XCDs 0 .. k
        const size_t first = list.size();
        while (sidx < pair_strips_ && (list.size() < want || k == 7)) {
            list.insert(list.end(), of_strip[(size_t)sidx].begin(), of_strip[(size_t)sidx].end());
            ++sidx;
        }
        // An XCD takes its units chunk by chunk, the strips of a chunk side by side -- as the arithmetic mapping of a
        // full mesh does -- so that the workgroups it runs at one time are NEIGHBOURING strips at the same planes and
        // the ring rows two of them both load meet in its L2.  (Until round 3 the order was strip by strip: the 32
        // workgroups of an XCD were 29 chunks of one strip and shared nothing; every ring row came from HBM -- the
        // concert hall's march ran at 3.3 TB/s where a box's runs at 5.85.)
        if (opt_.tuning.pair_units_by_chunk != 0)
            std::stable_sort(list.begin() + (std::ptrdiff_t)first, list.end(),
                             [](uint32_t a, uint32_t b) { return ((a >> 16) & 0x1FFu) < ((b >> 16) & 0x1FFu); });
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

// In-wall neighbours of the first n_xw_ entries by entry position, and the compact copies themselves.  Once per mesh.
template <typename Real>
int Engine<Real>::build_xwall() {
    if (xw_built_) return WV_OK;
    const uint32_t n = n_xw_;
    std::vector<uint32_t> bnode(n);
    std::vector<uint8_t> btype(n);
    WV_HIP(hipMemcpy(bnode.data(), bnode_, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    WV_HIP(hipMemcpy(btype.data(), btype_, (size_t)n, hipMemcpyDeviceToHost));
    std::vector<uint64_t> by_node(n);  // (stored node index, position), sorted: who lives where
    for (uint32_t p = 0; p < n; ++p) by_node[p] = ((uint64_t)bnode[p] << 32) | p;
    parallel_sort(by_node);
    std::vector<uint32_t> nbr((size_t)4 * n, wv::XW_FIELD);
    const int64_t stride[2] = {pitch_, (int64_t)pitch_ * ny_};
    const int lim[2] = {ny_, nz_};
    auto fill = [&](uint32_t first, uint32_t last) {
        for (uint32_t p = first; p < last; ++p) {
            const uint32_t idx = bnode[p];
            const uint32_t q = idx / (uint32_t)pitch_;
            const int at[2] = {(int)(q % (uint32_t)ny_), (int)(q / (uint32_t)ny_)};
            for (int k = 0; k < 4; ++k) {
                const int ax = k >> 1, c = at[ax] + ((k & 1) ? 1 : -1);
                if (c < 0 || c >= lim[ax]) continue;
                const uint64_t want = (uint64_t)((int64_t)idx + ((k & 1) ? stride[ax] : -stride[ax])) << 32;
                const auto it = std::lower_bound(by_node.begin(), by_node.end(), want);
                if (it == by_node.end() || (*it >> 32) != (want >> 32)) continue;  // not one of these entries: from the field
                const uint32_t other = (uint32_t)*it;
                nbr[(size_t)k * n + p] = other | (btype[other] == btype[p] ? wv::XW_SAME_FACING : 0u);
            }
        }
    };
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (n < (1u << 16) || hw < 2) {
        fill(0, n);
    } else {
        std::vector<std::thread> workers;
        for (unsigned t = 0; t < hw; ++t)
            workers.emplace_back(fill, (uint32_t)((uint64_t)n * t / hw), (uint32_t)((uint64_t)n * (t + 1) / hw));
        for (auto& w : workers) w.join();
    }
    WV_HIP(hipMalloc((void**)&xw_nbr_, nbr.size() * sizeof(uint32_t)));
    WV_HIP(hipMemcpy(xw_nbr_, nbr.data(), nbr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    WV_HIP(hipMalloc((void**)&xw_val_, (size_t)9 * n * sizeof(Real)));
    WV_HIP(hipMemsetAsync(xw_val_, 0, (size_t)9 * n * sizeof(Real), stream_));
    WV_HIP(hipMalloc((void**)&xw_gok_, (size_t)n));
    WV_HIP(hipMemsetAsync(xw_gok_, 0, (size_t)n, stream_));
    xw_built_ = true;
    xw_valid_ = false;
    return WV_OK;
}

template <typename Real>
void Engine<Real>::xwall_args(wv::BoundaryArgs<Real>& b) const {
    b.xw_n = n_xw_;
    b.xw_pad = (n_xw_ + 255u) / 256u * 256u;
    b.xw_nbr = xw_nbr_;
    b.xw_a = xw_val_;
    b.xw_b = xw_val_ + (size_t)n_xw_;
    b.xw_f = xw_val_ + (size_t)2 * n_xw_;
    b.xw_f1 = xw_val_ + (size_t)3 * n_xw_;
    b.xw_g = xw_val_ + (size_t)4 * n_xw_;
    b.xw_o2 = xw_val_ + (size_t)5 * n_xw_;
    b.xw_f2 = xw_val_ + (size_t)6 * n_xw_;
    b.xw_g2 = xw_val_ + (size_t)7 * n_xw_;
    b.xw_h2 = xw_val_ + (size_t)8 * n_xw_;
    b.xw_gok = xw_gok_;
}

template <typename Real>
void Engine<Real>::parallel_sort(std::vector<uint64_t>& v) {
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
// On a slab the two time levels each need the neighbours' face planes, so a pass has two exchanges.  Both run under the march
// (round 4; slab_early_now()).  With f = a face plane, n = the owned plane next to it, g = the ghost plane beyond it:
//   compute stream  [ghosts of t in place] -> f AND n to t+1 (one launch: sweep + their boundary nodes side by side, out of place)
//                   -> MARCH over n .. n' (t+1 and t+2; it computes n's t+1 again for its own use but stores t+1 only from
//                   the plane after n: its placeholders must not land on n's finished boundary values) -> boundary nodes of
//                   the planes in between to t+1 -> [source / receivers on t+1] -> fix-up list, boundary nodes n .. n' to t+2
//   halo stream     [f, n at t+1 final] -> exchange #1 (t+1 faces) -> f to t+2: a plain step of the face plane from t+1 at g, f, n
//                   (one launch, on this stream) -> exchange #2 (t+2 faces)
// Nothing on the halo stream needs the march: f's t+2 reads t+1 at g (exchange #1), f and n (the early launches).  The compute
// stream meets the halo stream again at the next pass's "ghosts in place".  Four launches on the compute stream, one beside it.
// A source within two planes of a cut (planes g, f, n: its t+1 sample would have to be in place before f's t+2) keeps THAT slab on
// the older order, which differs in where things are enqueued, not in what is exchanged, so neighbours need not agree:
//   part A  face planes to t+1 -> exchange #1 -> march over the planes in between + their boundary nodes to t+1
//   part B  ghosts of t+1 landed -> source / receivers on t+1 -> face planes to t+2 -> exchange #2 (nothing left to hide
//           behind but the last boundary launch) -> fix-up list and boundary nodes to t+2.
// A chain inside one process (wv_run_group) enqueues part A of every slab before part B of any: part B opens with the halo
// stream's wait for the neighbours' pushes of t+1, which must have been enqueued by then (comm.h, local transport).
// `fuse_mid`: the source / receiver work of step t+1 (and a short fix-up list) rides in the t+1 boundary launch
template <typename Real>
int Engine<Real>::enqueue_pair_a(int slot, uint64_t signal_pos, bool source_live, bool fuse_mid) {
    Real* A = field_[prv_];
    Real* B = field_[cur_];
    Real* O1 = field_[spare_[0]];
    Real* O2 = field_[spare_[1]];
    int* flag1 = flags_ + slot;
    int* flag2 = flags_ + slot + 1;
    int rc;
    std::string cerr;
    const bool early = slab_early_now();
    pair_early_ = early;
    if (comm_) {
        const int token = begin_halo_wait_timing();
        if (!comm_->wait_ghosts(stream_, cur_, &cerr)) return fail(WV_E_COMM, cerr);
        if ((rc = end_halo_wait_timing(token))) return rc;
    }
    if (!pre_post_done_ && !(batch_flags_reset_ && !n_recv_ && !source_live)) {  // step t: flag words of both steps, source sample into t, receivers from t
        wv::PrePostArgs<Real> pp = pre_post_args(B, slot, true, signal_pos, source_live);
        pp.flag2 = flag2;
        hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
    }
    pre_post_done_ = false;  // (else: the boundary launch before this pass has done it)
    if (comm_) {
        if ((rc = launch_faces(A, B, flag1, O1, early ? 2 : 1))) return rc;
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
    a.out1_z0 = early ? pair_s0_ : pair_z0_;
    a.out1_z1 = early ? pair_s1_ : pair_z1_;
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
    if (comm_ && !comm_->bulk_begin(stream_, &cerr)) return fail(WV_E_COMM, cerr);  // (slabs of one device take turns at the march)
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
    // (the boundary launches of every EIGHTH timed pass: a pair of events costs ~11 us of stream time, and two more pairs per pass
    // would take 2-3 % off a 512^3 run for a figure that needs a few dozen samples)
    pass_timed_ = timed && (part_timing_calls_++ & 7u) == 0;
    if (comm_ && !comm_->bulk_end(stream_, &cerr)) return fail(WV_E_COMM, cerr);
    // boundary nodes, t+1: own old value from t-1, neighbours from t, result into the t+1 field
    pair_mid_done_ = pair_list_done_ = false;
    if (xw_active_ && !xw_valid_) {  // the x-facing walls' compact copies, from fields t-1 and t
        wv::BoundaryArgs<Real> g = boundary_args(A, B, flag1);
        xwall_args(g);
        hipLaunchKernelGGL(wv::xwall_gather_kernel<Real>, dim3(g.xw_pad / 256), dim3(256), 0, stream_, g);
        xw_valid_ = true;
    }
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
        const int token = begin_part_timing(0);
        if ((rc = launch_boundary(A, B, flag1, pair_z0_, pair_z1_, &nx, O1, false, true))) return rc;
        if ((rc = end_part_timing(0, token))) return rc;
        pair_mid_done_ = true;
    } else {
        const int token = begin_part_timing(0);
        // (early: the planes next to the faces have been to t+1 already)
        if ((rc = launch_boundary(A, B, flag1, early ? pair_s0_ : pair_z0_, early ? pair_s1_ : pair_z1_, nullptr, O1, false, true))) return rc;
        if ((rc = end_part_timing(0, token))) return rc;
    }
    WV_HIP(hipGetLastError());
    return WV_OK;
}

template <typename Real>
int Engine<Real>::launch_fixup(uint32_t first, uint32_t n, const Real* t1, const Real* cur, Real* out2, int* flag2) {
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
template <typename Real>
int Engine<Real>::enqueue_pair_b(int slot, uint64_t signal_pos, bool source_live, int fuse_next) {
    Real* B = field_[cur_];
    Real* O1 = field_[spare_[0]];
    Real* O2 = field_[spare_[1]];
    int* flag2 = flags_ + slot + 1;
    int rc;
    std::string cerr;
    const bool io_mid = !pair_mid_done_ && (n_recv_ || source_live);  // step t+1: source sample into t+1, receivers from it
    if (comm_ && pair_early_) {
        // compute stream: only a slab with a source or receivers looks at the t+1 ghosts (a receiver's neighbours may lie there).
        // Asked for BEFORE exchange #2 is enqueued: the wait then stands for exchange #1 alone ("ghosts ready" / this slab's
        // latest push are re-recorded behind every exchange), not for the t+2 faces this step's source / receiver work has no use for.
        if (io_mid) {
            const int token = begin_halo_wait_timing();
            if (!comm_->wait_ghosts(stream_, spare_[0], &cerr)) return fail(WV_E_COMM, cerr);
            if ((rc = end_halo_wait_timing(token))) return rc;
        }
        // halo stream, behind exchange #1: ghost planes of t+1 in place -> the face planes to t+2, one more plain step of theirs
        // from t+1 at the ghost plane, the face and the plane next to it (all final since part A) -> exchange #2
        if (!comm_->wait_ghosts(comm_stream_, spare_[0], &cerr)) return fail(WV_E_COMM, cerr);
        on_ = comm_stream_;
        rc = launch_faces(B, O1, flag2, O2);
        on_ = nullptr;
        if (rc) return rc;
        WV_HIP(hipGetLastError());
        if (!comm_->exchange_faces(stream_, spare_[1], &cerr, true)) return fail(WV_E_COMM, cerr);
    } else if (comm_) {
        const int token = begin_halo_wait_timing();
        if (!comm_->wait_ghosts(stream_, spare_[0], &cerr)) return fail(WV_E_COMM, cerr);  // ghost planes of t+1
        if ((rc = end_halo_wait_timing(token))) return rc;
    }
    if (io_mid) {
        wv::PrePostArgs<Real> pp = pre_post_args(O1, slot + 1, true, signal_pos + 1, source_live);
        pp.flag = nullptr;  // reset in part A, and already written to by the march
        hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
    }
    if (comm_ && !pair_early_) {
        // the face planes to t+2: one more plain step of theirs, from the t+1 field with its ghost planes in place
        if ((rc = launch_faces(B, O1, flag2, O2))) return rc;
        WV_HIP(hipGetLastError());
        if (!comm_->exchange_faces(stream_, spare_[1], &cerr)) return fail(WV_E_COMM, cerr);
    }
    // t+2 of the nodes next to a boundary node / the source, from the complete t+1; then the boundary nodes
    // (most of them are faced by a boundary node and finished by its entry in the launch after this one)
    if (!pair_list_done_ && (rc = launch_fixup(0, pair_list_n_, O1, B, O2, flag2))) return rc;
    const int part_token = begin_part_timing(1);
    if (fuse_next && n_entries_ && io_nodes_unfaced()) {
        // what follows reads its source / receiver nodes from the t+2 field: none of them is written by
        // this launch (no boundary node, no node an entry finishes)
        wv::PrePostArgs<Real> nx = pre_post_args(O2, slot + 2, true, signal_pos + 2, source_live);
        if (fuse_next == 2) nx.flag2 = flags_ + slot + 3;
        if ((rc = launch_boundary(B, O1, flag2, pair_z0_, pair_z1_, &nx, O2, pair_inner_ok_ > 0, true))) return rc;
        pre_post_done_ = true;
    } else if ((rc = launch_boundary(B, O1, flag2, pair_z0_, pair_z1_, nullptr, O2, pair_inner_ok_ > 0, true))) {
        return rc;
    }
    if ((rc = end_part_timing(1, part_token))) return rc;
    pass_timed_ = false;
    WV_HIP(hipGetLastError());
    if (comm_ && !comm_->step_done(stream_, &cerr)) return fail(WV_E_COMM, cerr);
    ++passes_taken_;
    early_passes_ += pair_early_ ? 1 : 0;
    // roles: (previous, current) = (t+1, t+2); the fields that held t-1 and t are the spares now
    const int a_idx = prv_, b_idx = cur_;
    prv_ = spare_[0];
    cur_ = spare_[1];
    spare_[0] = a_idx;
    spare_[1] = b_idx;
    return WV_OK;
}

// May this slab's next pass step its faces and the planes next to them ahead of the march (both exchanges under it)?
template <typename Real>
bool Engine<Real>::slab_early_now() const {
    if (!comm_ || opt_.tuning.slab_early == 0 || !(opt_.ghost_lo || opt_.ghost_hi)) return false;
    // (by default only where a neighbour lives on another GPU: the order sweeps two more planes per pass to get the exchanges out
    // early, and between slabs that share a device there is no transfer worth hiding -- their copies take 20 us on a chip that is
    // theirs in turns)
    if (opt_.tuning.slab_early < 0 && !comm_->peers_elsewhere()) return false;
    const int lo = opt_.ghost_lo ? 2 : 0, hi = opt_.ghost_hi ? 2 : 0;
    if (z_end_ - z_begin_ < lo + hi + 4) return false;  // (something has to be left in between)
    if (source_kind_ != WV_SOURCE_NONE) {
        // the sample of step t+1 goes in after the march; a face's t+2 would read planes g, f, n before that
        const int sz = (int)(source_node_ / ((uint64_t)pitch_ * (uint64_t)ny_));
        if ((opt_.ghost_lo && sz < z_begin_ + 2) || (opt_.ghost_hi && sz >= z_end_ - 2)) return false;
    }
    return true;
}

// part 0 / 1 of the two-step pass that covers steps i and i + 1 of the batch
template <typename Real>
int Engine<Real>::enqueue_batch_pair(uint64_t i, int part, int next_kind) {
    DeviceGuard guard(device_);
    return part == 0 ? enqueue_pair_a((int)i, signal_pos_ + i, batch_source_live_, batch_can_fuse_)
                     : enqueue_pair_b((int)i, signal_pos_ + i, batch_source_live_, batch_can_fuse_ ? next_kind : 0);
}

// Would this engine take two-step passes in the batch being planned?  Costs nothing (no allocation, no device work).
template <typename Real>
int Engine<Real>::batch_pair_eligible(int* eligible) {
    DeviceGuard guard(device_);
    *eligible = pair_eligible() ? 1 : 0;
    return WV_OK;
}

// ... then, once every slab of the chain is eligible: spare fields, pair map, lists (ensure_pair).  *ready = 0 when that
// did not work out here (no memory for four fields; a sparse room where the march's live units cost more than the
// sweep's live tiles); *singles_first = single full sweeps needed first because a caller wrote into outside nodes.
template <typename Real>
int Engine<Real>::batch_pair_prepare(int* ready, int* singles_first) {
    DeviceGuard guard(device_);
    *ready = 0;
    *singles_first = 0;
    const int rc = ensure_pair();
    if (rc == WV_E_HIP && wv::last_hip_error() == hipErrorOutOfMemory) {
        // no room for the map / the lists / the walls' compact copies either: single steps need none of them (the spare fields go
        // back with the veto that follows a batch without passes)
        (void)hipGetLastError();
        pair_failed_ = true;
        return WV_OK;
    }
    if (rc) return rc;
    if (!pair_failed_ && (opt_.tuning.pair > 0 || pair_sparse_ok_)) {
        *ready = 1;
        *singles_first = std::min(outside_dirty_, 2);
    }
    return WV_OK;
}

// The chain (or this engine alone) stays with single steps: two spare fields are a third of a slab's memory.
template <typename Real>
int Engine<Real>::batch_pair_vetoed() {
    DeviceGuard guard(device_);
    bool any = false;
    for (int i = 0; i < 2; ++i) any = any || field_[spare_[i]] != nullptr;
    if (!any) return WV_OK;
    if (comm_ && comm_->is_ipc()) return WV_OK;  // (the neighbours have these fields mapped: they stay)
    WV_HIP(hipStreamSynchronize(stream_));
    WV_HIP(hipStreamSynchronize(comm_stream_));
    for (int i = 0; i < 2; ++i) {
        Real*& f = field_[spare_[i]];
        if (f) (void)hipFree(f);
        f = nullptr;
    }
    if (comm_) {
        void* fields[4] = {field_[0], field_[1], field_[2], field_[3]};
        comm_->set_fields(fields, 4, (size_t)pitch_ * ny_ * sizeof(Real), nz_);
    }
    return WV_OK;
}

}  // namespace wv
