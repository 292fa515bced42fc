// engine_triple.hip.h -- three time steps per pass over the fields (triple_kernels.hip.h): when, with what map and lists, in which launches.
//
// Part of the engine behind the C ABI of include/wayverb_amd.h (engine.hip is the translation unit; see engine.hip.h for
// the class and the map of which file holds what).
//
// A pass takes (t-1, t) to (t+2, t+3):
//   [source sample into t, receivers from t]
//   MARCH                      reads t-1, t; writes t+2 and t+3 everywhere (placeholders where it cannot know), t+1 at shell nodes only
//   exact flags                (a launch that leaves at once unless the march has seen an inf or a nan)
//   boundary nodes -> t+1      own old value from t-1, neighbours from t, into the t+1 field
//   [source sample into t+1, receivers from t+1]
//   fix-up list 2 -> t+2       the nodes next to something that is not a plain node (those no boundary entry finishes), from the t+1 field
//   boundary nodes -> t+2      own old value from t, neighbours from t+1; a 1-D entry also finishes the inside node it faces
//   [source sample into t+2, receivers from t+2]
//   fix-up list 3 -> t+3       every shell node (something else within two nodes), from the finished t+2 field and its own t+1
//   boundary nodes -> t+3      own old value from t+1, neighbours from t+2
// Five fields: the engine's four rotate as in a two-step pass (the two that held t-1 and t receive t+2 and t+3 of the next pass), the
// fifth holds t+1 at the shell nodes and the boundary nodes of whatever pass is in flight and zeros at outside nodes.
// (the exact-flags check shares the launch of fix-up list 3; the source / receiver work rides in the boundary launches where it may:
// five launches.)  The x-facing walls' entries work on compact copies at all three levels and finish the two nodes in front of them
// (boundary_kernels.hip.h, xwall3_node); a room that leaves much of its mesh outside marches a work list (build_triple_units); a
// z-slab of a chain takes the same pass in three parts around three halo exchanges (enqueue_triple_slab).
// Results are bit-identical to three single steps (tests/test_gpu_triple.py, test_gpu_slabs.py).
#pragma once
#include "engine.hip.h"
#include "triple_kernels.hip.h"

namespace wv {

// May this engine take three-step passes right now?  Everything a two-step pass needs (it shares the pair map, the second level's
// fix-up list and the spare fields), a slab thick enough to leave the march something between its faces' neighbours, and -- unless
// forced -- a mesh big enough that bytes, not launches and warm-up planes, decide.
template <typename Real>
bool Engine<Real>::triple_eligible() {
    if (opt_.tuning.triple == 0 || triple_failed_) return false;
    const bool slab = opt_.ghost_lo || opt_.ghost_hi;
    if (slab != (comm_ != nullptr)) return false;  // (ghost planes nobody fills; a communicator with nobody behind it)
    if (slab) {
        // A z-slab (enqueue_triple_slab): the march leaves out the face plane and the plane next to it where a neighbour follows
        if (z_end_ - z_begin_ < (opt_.ghost_lo ? 2 : 0) + (opt_.ghost_hi ? 2 : 0) + 2) return false;
    }
    if (!pair_eligible()) return false;
    const int lb = triple_lane_bytes();
    const int WX = 64 * (lb / (int)sizeof(Real));
    uint8_t win[4][wv::kTripleMaxWindows];
    int widest = 0;
    if (wv::triple_windows(pitch_ / WX, win, &widest, false, wv::triple_max_waves(lb)) < 0) return false;
    if (opt_.tuning.triple < 0 && stored_nodes_ < triple_min_nodes_) return false;
    return true;
}

// Bytes of a row per lane of the march.  Doubles: 16 (half the instructions per byte, two waves per SIMD) or 8 (three narrower waves per
// SIMD, finer pieces of rows in a sparse room's work list), whichever ran faster on boxes of that row length (profiles/r06/
// lane_width_by_size.txt; the concert hall at 1 600 Hz, 640-double rows, agrees: 390 against 374): 8 up to 256 doubles (a workgroup of
// four waves, three of them per CU), 16 from 320, 8 again from 512 (eight to ten waves: one workgroup fills a CU), 16 from 768 (rows
// that need windows of 8-byte lanes).  Floats: 8, but 16 (four floats per lane, with a row less of lookahead: 249 registers) on rows of
// 513-768 floats, three waves per workgroup and two workgroups per CU: 640^3 473 -> 545, 704^3 501 -> 608, 768^3 580 -> 675
// Gnode-updates/s end to end; on rows of 1024 it loses (832^3 569 -> 505, 1024^3 734 -> 636), up to 512 it ties.
// wv_tuning::triple_lanes forces one.
template <typename Real>
int Engine<Real>::triple_lane_bytes() const {
    if (opt_.tuning.triple_lanes == 8 || opt_.tuning.triple_lanes == 16) return opt_.tuning.triple_lanes;
    if (sizeof(Real) == 4) return (pitch_ > 512 && pitch_ <= 768) ? 16 : 8;
    if (pitch_ < triple_wide_from_) return 8;
    return (pitch_ >= 512 && pitch_ < 768) ? 8 : 16;
}

template <typename Real>
int Engine<Real>::ensure_triple() {
    int rc = ensure_pair();
    if (rc) return rc;
    triple_z0_ = z_begin_ + (opt_.ghost_lo ? 2 : 0);
    triple_z1_ = z_end_ - (opt_.ghost_hi ? 2 : 0);
    if (pair_failed_ || (!pair_sparse_ok_ && opt_.tuning.triple < 0)) {  // (a room so sparse that the sweep's tiles beat the march's units keeps single steps)
        triple_ready_ = false;
        return WV_OK;
    }
    const uint64_t src = source_kind_ != WV_SOURCE_NONE ? source_node_ : ~0ull;
    if (!field1_) {
        if (hipMalloc((void**)&field1_, field_bytes_ + 256) != hipSuccess) {
            (void)hipGetLastError();
            field1_ = nullptr;
            triple_failed_ = true;  // not enough memory for a fifth field: two-step passes
            triple_ready_ = false;
            return WV_OK;
        }
        WV_HIP(hipMemsetAsync(field1_, 0, field_bytes_ + 256, stream_));
    }
    if (!suspect_) {
        WV_HIP(hipMalloc((void**)&suspect_, kRing * sizeof(int)));
        WV_HIP(hipMemsetAsync(suspect_, 0, kRing * sizeof(int), stream_));
    }
    // x-facing walls on their compact copies through all three levels where the two-step passes run on them (ensure_pair: the entries
    // finish the nodes they face, the source is clear of them)
    // (a slab: only where those entries keep clear of the planes next to the faces as well as of the faces -- xwall_eligible_kernel
    // under wv_tuning::slab_early -- which a pass here steps by gathering from the fields)
    const bool xw = xw_active_ && pair_inner_ok_ > 0 && opt_.tuning.boundary_xwall != 2 && !(comm_ && opt_.tuning.slab_early == 0);
    if (triple_map_ && triple_source_ == src && triple_io_generation_ == io_generation_ && triple_xw_ == xw) {
        triple_ready_ = !(pair_units_ && !triple_units_);  // (a sparse room whose work list could not be had stays with two-step passes)
        return WV_OK;
    }
    const uint64_t cls_bytes = (uint64_t)cls_pitch_ * 4u * (uint64_t)((ny_ + 3) / 4) * nz_;
    if (!triple_map_) {
        WV_HIP(hipMalloc((void**)&triple_map_, cls_bytes + 16));
        WV_HIP(hipMemsetAsync(triple_map_, 0, cls_bytes + 16, stream_));
    }
    // map + third-level list: count per block, scan on the host, fill
    const int64_t n_bytes = (int64_t)cls_pitch_ * ny_ * nz_;
    const unsigned blocks = (unsigned)((n_bytes + 255) / 256);
    ScopedDevice counts;
    WV_HIP(hipMalloc(&counts.p, (size_t)blocks * sizeof(uint32_t)));
    wv::TripleMapArgs m{};
    m.pair_map = pair_map_;
    m.map = triple_map_;
    m.block_count = static_cast<uint32_t*>(counts.p);
    m.ny = ny_;
    m.nz = nz_;
    m.pitch = pitch_;
    m.cls_pitch = cls_pitch_;
    m.z_begin = triple_z0_;  // (the third level's list: the march's planes)
    m.z_end = triple_z1_;
    ScopedDevice covered;  // the nodes those entries finish at the third level: not on its list
    if (xw) {
        const size_t words = (size_t)((stored_nodes_ + 31) / 32) + 1;
        WV_HIP(hipMalloc(&covered.p, words * sizeof(uint32_t)));
        WV_HIP(hipMemsetAsync(covered.p, 0, words * sizeof(uint32_t), stream_));
        wv::XwCoverArgs c{};
        c.bnode = bnode_;
        c.btype = btype_;
        c.covered = static_cast<uint32_t*>(covered.p);
        c.xw_n = n_xw_;
        c.pair_map = pair_map_;
        c.gok = xw_gok_;
        c.nx = nx_;
        c.ny = ny_;
        c.nz = nz_;
        c.pitch = pitch_;
        c.cls_pitch = cls_pitch_;
        hipLaunchKernelGGL(wv::xwall_cover_kernel, dim3((n_xw_ + 255) / 256), dim3(256), 0, stream_, c);
        WV_HIP(hipGetLastError());
        m.covered = c.covered;
    }
    triple_xw_ = xw;
    hipLaunchKernelGGL(wv::triple_map_kernel, dim3(blocks), dim3(256), 0, stream_, m);
    WV_HIP(hipGetLastError());
    std::vector<uint32_t> per_block(blocks);
    WV_HIP(hipMemcpyAsync(per_block.data(), counts.p, (size_t)blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
    WV_HIP(hipStreamSynchronize(stream_));
    uint64_t total = 0;
    for (unsigned b = 0; b < blocks; ++b) {
        const uint32_t c = per_block[b];
        per_block[b] = (uint32_t)total;
        total += c;
    }
    if (triple_list_) {
        (void)hipFree(triple_list_);
        triple_list_ = nullptr;
    }
    triple_list_n_ = 0;
    if (total >= (1ull << 32)) {
        triple_ready_ = false;
        return WV_OK;
    }
    if (total) {
        WV_HIP(hipMalloc((void**)&triple_list_, (size_t)total * sizeof(uint32_t)));
        WV_HIP(hipMemcpyAsync(counts.p, per_block.data(), (size_t)blocks * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
        m.list = triple_list_;
        hipLaunchKernelGGL(wv::triple_map_kernel, dim3(blocks), dim3(256), 0, stream_, m);
        WV_HIP(hipGetLastError());
        triple_list_n_ = (uint32_t)total;
    }
    WV_HIP(hipStreamSynchronize(stream_));  // (per_block and the bitmap belong to this call)
    // receivers and the source node: their t+1 is read / written in the t+1 field whatever lies around them
    if (n_recv_ || src != ~0ull) {
        wv::TripleMarkArgs k{};
        k.map = triple_map_;
        k.recv = recv_nodes_;
        k.n_recv = n_recv_;
        k.source_node = src;
        k.ny = ny_;
        k.pitch = pitch_;
        k.cls_pitch = cls_pitch_;
        hipLaunchKernelGGL(wv::triple_mark_kernel, dim3((n_recv_ + 1 + 255) / 256), dim3(256), 0, stream_, k);
        WV_HIP(hipGetLastError());
    }
    triple_source_ = src;
    triple_io_generation_ = io_generation_;
    // march geometry: strips of four rows; windows where a row is longer than a workgroup; chunks along z so that the workgroups fill
    // whole rounds of the chip's workgroup slots, weighed against the four warm-up planes every chunk marches before its first output
    const int lb = triple_lb_ = triple_lane_bytes();
    const int WX = 64 * (lb / (int)sizeof(Real));
    int widest = 0;
    triple_windows_ = wv::triple_windows(pitch_ / WX, triple_win_, &widest, false, wv::triple_max_waves(lb));
    if (triple_windows_ < 0) {
        triple_ready_ = false;
        return WV_OK;
    }
    triple_nw_ = widest;
    triple_strips_ = (ny_ + wv::kTripleRows - 1) / wv::kTripleRows;
    const size_t lds = wv::triple_lds_bytes(triple_nw_, false, lb);
    const int by_lds = std::max<int>(1, (int)((160u * 1024u) / lds));
    const int by_waves = std::max(1, (lb == 16 ? 8 : 12) / triple_nw_);
    const int64_t slots = 256ll * std::min(by_lds, by_waves);
    const int owned = triple_z1_ - triple_z0_;
    int chunks = opt_.tuning.triple_chunks;
    if (chunks <= 0) {
        // (a slab with a neighbour on another GPU: at least two rounds where that costs little, so that the exchange of the t+1 faces --
        // enqueued ahead of the march, but in need of a CU where it is carried by kernels -- gets in at the first round's end instead
        // of after the march: as ensure_pair chooses for the two-step march)
        const int64_t want_rounds = ((opt_.ghost_lo || opt_.ghost_hi) && comm_ && comm_->peers_elsewhere()) ? 2 : 1;
        double best[2] = {0, 0};
        int at[2] = {0, 0};  // [0] any number of rounds, [1] at least `want_rounds`
        for (int c = 1; c <= std::max(1, owned / 12) && c <= 256; ++c) {
            const int64_t wgs = (int64_t)triple_strips_ * c * std::max(1, triple_windows_);
            const int64_t rounds = (wgs + slots - 1) / slots;
            const double zc = (double)((owned + c - 1) / c);
            const double cost = (double)(rounds * slots) / (double)wgs * (zc + 4.0) / zc;
            for (int k = 0; k < 2; ++k)
                if ((k == 0 || rounds >= want_rounds) && (at[k] == 0 || cost < best[k] - 1e-9)) {
                    best[k] = cost;
                    at[k] = c;
                }
        }
        chunks = (at[1] && best[1] <= 1.06 * best[0]) ? at[1] : std::max(1, at[0]);
    }
    chunks = std::max(1, std::min(chunks, std::max(1, owned / 4)));
    triple_zc_ = (owned + chunks - 1) / chunks;
    triple_chunks_ = (owned + triple_zc_ - 1) / triple_zc_;
    // a room that leaves much of its mesh outside (the two-step march runs over a work list): so does this one
    if (triple_units_) {
        (void)hipFree(triple_units_);
        triple_units_ = nullptr;
    }
    if (pair_units_ && (rc = build_triple_units())) return rc;
    if (pair_units_ && !triple_units_) {  // (no list to be had: two-step passes)
        triple_ready_ = false;
        return WV_OK;
    }
    if (!triple_attr_set_) {
        WV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wv::triple_march_kernel<Real, 0, false, 8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        WV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wv::triple_march_kernel<Real, 0, false, kWideLaneBytes>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        triple_attr_set_ = true;
    }
    triple_ready_ = true;
    return WV_OK;
}

// The three-step march's work list for a room that leaves much of its mesh outside, after build_pair_units (engine_pair.hip.h): a unit is a
// strip of four rows through one chunk of planes, listed when it holds a node to update, with the waves of its row between the first and
// the last that hold anything but `none` nodes in what it reads or hands on (its rows +- a strip, its planes +- 3).  Each XCD takes a
// run of neighbouring strips with about the same number of units, chunk by chunk.  Sets triple_zc_ / triple_chunks_ to the units' height.
template <typename Real>
int Engine<Real>::build_triple_units() {
    // (a slab: the march's planes; a unit that ends at either end of them also stores t+2 on the plane beyond it -- TripleArgs::z2_lo /
    // z2_hi --, so a node to update there makes the unit live as well)
    const int owned = triple_z1_ - triple_z0_;
    const int z_lo = triple_z0_, z_hi = triple_z1_, extra_lo = triple_z0_ > z_begin_ ? 1 : 0, extra_hi = triple_z1_ < z_end_ ? 1 : 0;
    const int lb = triple_lb_;
    const int WX = 64 * (lb / (int)sizeof(Real));
    const int row_waves = pitch_ / WX;
    if (triple_windows_ || triple_strips_ >= (1 << 14) || row_waves > 16) return WV_OK;
    const int64_t n_cells = (int64_t)nz_ * triple_strips_;
    ScopedDevice act_mem, raw_mem;
    WV_HIP(hipMalloc(&act_mem.p, (size_t)n_cells));
    WV_HIP(hipMalloc(&raw_mem.p, (size_t)n_cells * sizeof(uint16_t)));
    wv::TileActivityArgs t{};
    t.cls = cls_;
    t.active = static_cast<uint8_t*>(act_mem.p);
    t.ny = ny_;
    t.nz = nz_;
    t.pitch = pitch_;
    t.cls_pitch = cls_pitch_;
    t.tile_rows = wv::kTripleRows;
    t.tile_cols = pitch_;
    t.tiles_x = 1;
    t.tiles_y = triple_strips_;
    hipLaunchKernelGGL(wv::tile_activity_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, stream_, t);
    WV_HIP(hipGetLastError());
    static_assert(wv::kTripleRows == wv::kPairRows, "pair_wave_activity_kernel counts strips of kPairRows rows");
    wv::WaveActivityArgs w{};
    w.cls = cls_;
    w.raw16 = static_cast<uint16_t*>(raw_mem.p);
    w.ny = ny_;
    w.nz = nz_;
    w.pitch = pitch_;
    w.cls_pitch = cls_pitch_;
    w.strips = triple_strips_;
    w.nw = row_waves;
    w.wave_cols = WX;
    hipLaunchKernelGGL(wv::pair_wave_activity_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, stream_, w);
    WV_HIP(hipGetLastError());
    std::vector<uint8_t> active((size_t)n_cells);
    std::vector<uint16_t> raw((size_t)n_cells);
    WV_HIP(hipMemcpyAsync(active.data(), act_mem.p, (size_t)n_cells, hipMemcpyDeviceToHost, stream_));
    WV_HIP(hipMemcpyAsync(raw.data(), raw_mem.p, (size_t)n_cells * sizeof(uint16_t), hipMemcpyDeviceToHost, stream_));
    WV_HIP(hipStreamSynchronize(stream_));
    // How many planes to a unit?  About wv_tuning::pair_unit_planes + 8 (a unit marches four warm-up planes before its first output where
    // a two-step unit marches three), and among the heights near that the one whose units fill the chip's workgroup slots in the fewest,
    // fullest rounds -- as build_pair_units chooses.  wv_tuning::triple_chunks > 0 sets the number of chunks instead.
    const int per_cu = std::max(1, (lb == 16 ? 8 : 12) / std::max(1, triple_nw_));
    const int64_t slots_per_xcd = 32ll * per_cu;
    auto units_of = [&](int height, std::vector<uint32_t>* per_strip) -> uint64_t {
        const int n_chunks = (owned + height - 1) / height;
        uint64_t units = 0;
        for (int sidx = 0; sidx < triple_strips_; ++sidx)
            for (int c = 0; c < n_chunks; ++c) {
                const int zb = z_lo + c * height, ze = std::min(zb + height, z_hi);
                bool any = false;
                for (int z = zb - (zb == z_lo ? extra_lo : 0); z < ze + (ze == z_hi ? extra_hi : 0) && !any; ++z) any = active[(size_t)z * triple_strips_ + sidx] != 0;
                if (per_strip) (*per_strip)[(size_t)sidx] += any;
                units += any;
            }
        return units;
    };
    auto rounds_cost = [&](int height) -> double {
        std::vector<uint32_t> per_strip((size_t)triple_strips_, 0u);
        const uint64_t units = units_of(height, &per_strip);
        if (!units) return 0.0;
        uint64_t longest = 0, so_far = 0, start = 0;  // the same partition into eight runs of strips as below
        int sidx = 0;
        for (int k = 0; k < 8; ++k) {
            const uint64_t want = units * (uint64_t)(k + 1) / 8;
            while (sidx < triple_strips_ && (so_far < want || k == 7)) so_far += per_strip[(size_t)sidx++];
            longest = std::max(longest, so_far - start);
            start = so_far;
        }
        return (double)((longest + slots_per_xcd - 1) / slots_per_xcd) * (double)(height + 4);
    };
    int zc;
    if (opt_.tuning.triple_chunks > 0) {
        zc = std::max(4, (owned + opt_.tuning.triple_chunks - 1) / opt_.tuning.triple_chunks);
    } else {
        const int base = std::max(8, std::min(owned, opt_.tuning.pair_unit_planes + 8));
        zc = base;
        double best = rounds_cost(zc);
        for (int height = base * 3 / 4; height <= base * 5 / 4; ++height) {
            if (height < 8 || height > owned) continue;
            const double cost = rounds_cost(height);
            if (cost > 0 && cost < best * 0.97) {  // (only a clear win moves the height)
                best = cost;
                zc = height;
            }
        }
    }
    const int chunks = (owned + zc - 1) / zc;
    if (chunks >= (1 << 9)) return WV_OK;  // (9 bits of a list entry)
    std::vector<std::vector<uint32_t>> of_strip((size_t)triple_strips_);
    uint64_t total = 0, live_waves = 0;
    for (int sidx = 0; sidx < triple_strips_; ++sidx)
        for (int c = 0; c < chunks; ++c) {
            const int zb = z_lo + c * zc, ze = std::min(zb + zc, z_hi);
            bool any = false;
            for (int z = zb - (zb == z_lo ? extra_lo : 0); z < ze + (ze == z_hi ? extra_hi : 0) && !any; ++z) any = active[(size_t)z * triple_strips_ + sidx] != 0;
            if (!any) continue;
            uint32_t bits = 0;
            for (int z = std::max(0, zb - 3); z < std::min(nz_, ze + 3); ++z)
                for (int ss = std::max(0, sidx - 1); ss <= std::min(triple_strips_ - 1, sidx + 1); ++ss) bits |= raw[(size_t)z * triple_strips_ + ss];
            const uint32_t lo = std::min((uint32_t)__builtin_ctz(bits | (1u << 31)), (uint32_t)row_waves - 1u);
            const uint32_t hi = std::min(32u - (uint32_t)__builtin_clz(bits | 1u), (uint32_t)row_waves);
            const uint32_t span = hi > lo ? hi - lo : 1u;
            of_strip[(size_t)sidx].push_back(wv::triple_unit_entry((uint32_t)sidx, (uint32_t)c, lo, span));
            live_waves += span;
            ++total;
        }
    if (!total) return WV_OK;
    triple_live_frac_ = (double)live_waves / ((double)triple_strips_ * chunks * row_waves);
    std::vector<uint32_t> list;
    list.reserve((size_t)total);
    triple_units_longest_ = 0;
    int sidx = 0;
    for (int k = 0; k < 8; ++k) {
        triple_unit_start_[k] = (uint32_t)list.size();
        const uint64_t want = total * (uint64_t)(k + 1) / 8;  // cumulative share of XCDs 0 .. k
        const size_t first = list.size();
        while (sidx < triple_strips_ && (list.size() < want || k == 7)) {
            list.insert(list.end(), of_strip[(size_t)sidx].begin(), of_strip[(size_t)sidx].end());
            ++sidx;
        }
        // (chunk by chunk, the strips of a chunk side by side: what an XCD runs at one time are neighbouring strips at the same planes)
        std::stable_sort(list.begin() + (std::ptrdiff_t)first, list.end(), [](uint32_t a, uint32_t b) { return ((a >> 14) & 0x1FFu) < ((b >> 14) & 0x1FFu); });
        triple_units_longest_ = std::max<uint32_t>(triple_units_longest_, (uint32_t)list.size() - triple_unit_start_[k]);
    }
    triple_unit_start_[8] = (uint32_t)list.size();
    uint32_t* staged = nullptr;
    WV_HIP(hipMalloc((void**)&staged, list.size() * sizeof(uint32_t)));
    if (hipMemcpy(staged, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(staged);
        return fail(WV_E_HIP, "copying the three-step march's unit list to the device failed");
    }
    triple_units_ = staged;
    triple_zc_ = zc;
    triple_chunks_ = chunks;
    return WV_OK;
}

// The march of a pass over the planes [triple_z0_, triple_z1_) (a slab: t+2 also on the plane next to a face, which only the march can
// supply), timed in an account of its own (WV_QUERY_TRIPLE_MARCH_NS); every eighth timed pass times its other launches too.
template <typename Real>
int Engine<Real>::launch_triple_march(int slot, const Real* A, const Real* B, Real* O1, Real* O2, Real* O3) {
    int rc;
    wv::TripleArgs<Real> a{};
    a.prev = A;
    a.cur = B;
    a.out1 = O1;
    a.out2 = O2;
    a.out3 = O3;
    a.map = triple_map_;
    a.suspect = suspect_ + slot;
    a.ny = ny_;
    a.nz = nz_;
    a.pitch = pitch_;
    a.cls_pitch = cls_pitch_;
    a.z_begin = triple_z0_;
    a.z_end = triple_z1_;
    a.z2_lo = triple_z0_ > z_begin_ ? 1 : 0;
    a.z2_hi = triple_z1_ < z_end_ ? 1 : 0;
    a.nw = triple_nw_;
    a.zc = triple_zc_;
    a.chunks = triple_chunks_;
    a.strips = triple_strips_;
    a.strips_per_xcd = (triple_strips_ + 7) / 8;
    a.windows = triple_windows_;
    for (int k = 0; k < triple_windows_; ++k) {
        a.win_first |= (uint64_t)triple_win_[0][k] << (8 * k);
        a.win_count |= (uint64_t)triple_win_[1][k] << (8 * k);
        a.win_store_lo |= (uint64_t)triple_win_[2][k] << (8 * k);
        a.win_store_hi |= (uint64_t)triple_win_[3][k] << (8 * k);
    }
    unsigned grid = 8u * (unsigned)a.strips_per_xcd * (unsigned)triple_chunks_ * (unsigned)std::max(1, triple_windows_);
    if (triple_units_) {
        a.unit_list = triple_units_;
        for (int k = 0; k < 9; ++k) a.list_start[k] = triple_unit_start_[k];
        grid = 8u * triple_units_longest_;
    }
    const bool timed = timing && time_this_launch();
    const int token = timed ? begin_part_timing(4, true) : -1;
    if (triple_lb_ == 8)
        hipLaunchKernelGGL((wv::triple_march_kernel<Real, 0, false, 8>), dim3(grid), dim3(64u * (unsigned)triple_nw_),
                           wv::triple_lds_bytes(triple_nw_, false, 8), stream_, a);
    else
        hipLaunchKernelGGL((wv::triple_march_kernel<Real, 0, false, kWideLaneBytes>), dim3(grid), dim3(64u * (unsigned)triple_nw_),
                           wv::triple_lds_bytes(triple_nw_, false, kWideLaneBytes), stream_, a);
    if ((rc = end_part_timing(4, token))) return rc;
    pass_timed_ = timed && (part_timing_calls_++ & 7u) == 0;
    return WV_OK;
}

// The third level's list -- every shell node of the march's planes from the finished t+2 field -- and the exact error bits of what the
// march was the last to write, should it have seen an inf or a nan (fields t-1 and t are still what the march read).
template <typename Real>
int Engine<Real>::launch_triple_list(int slot, const Real* A, const Real* B, const Real* O1, const Real* O2, Real* O3, bool source_live) {
    wv::PairFixupArgs<Real> f{};
    f.nodes = triple_list_;
    f.n = triple_list_n_;
    f.t1 = O2;
    f.cur = O1;
    f.out2 = O3;
    f.flag2 = flags_ + slot + 2;
    f.nx = nx_;
    f.ny = ny_;
    f.nz = nz_;
    f.pitch = pitch_;
    wv::TripleFlagsArgs<Real> g{};
    g.prev = A;
    g.cur = B;
    g.out2 = O2;
    g.out3 = O3;
    g.pair_map = pair_map_;
    g.suspect = suspect_ + slot;
    g.flag1 = flags_ + slot;
    g.flag2 = flags_ + slot + 1;
    g.flag3 = flags_ + slot + 2;
    g.source_node = source_live ? source_node_ : ~0ull;
    g.nx = nx_;
    g.ny = ny_;
    g.nz = nz_;
    g.pitch = pitch_;
    g.cls_pitch = cls_pitch_;
    g.z_begin = triple_z0_;
    g.z_end = triple_z1_;
    hipLaunchKernelGGL(wv::triple_list_kernel<Real>, dim3(std::max(1u, (triple_list_n_ + 255) / 256)), dim3(256), 0, stream_, f, g);
    return WV_OK;
}

// Steps `slot` .. `slot + 2` of a batch in one pass.  The flag words of the batch hold the mesh-static bits already (run()).
// `fuse_next` (0: nothing follows in this batch, 1: a single step or another three-step pass, 2: a two-step pass): the next step's source /
// receiver work rides in the last boundary launch -- like the source / receiver work of steps t+1 and t+2 in the first two, and the
// second level's short list with t+1's -- wherever the launch in question writes none of the nodes concerned (the same conditions as in
// a two-step pass: engine_pair.hip.h).  Five launches per pass then: march, boundary nodes to t+1, to t+2, third-level list + exact
// flags, boundary nodes to t+3.
template <typename Real>
int Engine<Real>::enqueue_triple(int slot, uint64_t signal_pos, bool source_live, int fuse_next) {
    DeviceGuard guard(device_);
    Real* A = field_[prv_];
    Real* B = field_[cur_];
    Real* O1 = field1_;
    Real* O2 = field_[spare_[0]];
    Real* O3 = field_[spare_[1]];
    int* flag1 = flags_ + slot;
    int* flag2 = flags_ + slot + 1;
    int* flag3 = flags_ + slot + 2;
    int rc;
    const bool io = n_recv_ || source_live;
    const bool fuse = batch_can_fuse_ && n_entries_ != 0;
    if (!pre_post_done_ && io) {  // step t: source sample into t, receivers from t
        wv::PrePostArgs<Real> pp = pre_post_args(B, slot, true, signal_pos, source_live);
        pp.flag = nullptr;
        hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
    }
    pre_post_done_ = false;
    if ((rc = launch_triple_march(slot, A, B, O1, O2, O3))) return rc;
    // level 1: boundary nodes to t+1 -- and, by the launch's last workgroup, step t+1's source sample / receivers (none of those nodes
    // is a boundary node: their t+1 has been final since the march) and then the second level's list where it is short and none of its
    // nodes has a boundary node for a neighbour (the source's neighbours, typically)
    bool list2_done = false;
    const bool xw = triple_xw_ && xw_active_;
    if (xw && !xw_valid_) {  // the x-facing walls' compact copies, from fields t-1 and t
        wv::BoundaryArgs<Real> g = boundary_args(A, B, flag1);
        xwall_args(g);
        hipLaunchKernelGGL(wv::xwall_gather_kernel<Real>, dim3(g.xw_pad / 256), dim3(256), 0, stream_, g);
    }
    xw_valid_ = xw;  // (passes that do not maintain the copies leave them behind)
    int token = begin_part_timing(0);
    if (fuse && io) {
        wv::PrePostArgs<Real> nx = pre_post_args(O1, slot + 1, true, signal_pos + 1, source_live);
        nx.flag = nullptr;
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
            list2_done = true;
        }
        if ((rc = launch_boundary(A, B, flag1, z_begin_, z_end_, &nx, O1, false, false, nullptr, xw ? 1 : 0))) return rc;
    } else {
        if ((rc = launch_boundary(A, B, flag1, z_begin_, z_end_, nullptr, O1, false, false, nullptr, xw ? 1 : 0))) return rc;
        if (io) {
            wv::PrePostArgs<Real> pp = pre_post_args(O1, slot + 1, true, signal_pos + 1, source_live);
            pp.flag = nullptr;
            hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
        }
    }
    if ((rc = end_part_timing(0, token))) return rc;
    // level 2: the second level's list (as in a two-step pass), then the boundary nodes, whose 1-D entries finish the nodes they face --
    // with step t+2's source / receiver work where none of those nodes is written by the launch
    if (!list2_done && (rc = launch_fixup(0, pair_list_n_, O1, B, O2, flag2))) return rc;
    token = begin_part_timing(1);
    if (fuse && io && io_nodes_unfaced()) {
        wv::PrePostArgs<Real> nx = pre_post_args(O2, slot + 2, true, signal_pos + 2, source_live);
        nx.flag = nullptr;
        if ((rc = launch_boundary(B, O1, flag2, z_begin_, z_end_, &nx, O2, pair_inner_ok_ > 0, false, nullptr, xw ? 2 : 0))) return rc;
    } else {
        if ((rc = launch_boundary(B, O1, flag2, z_begin_, z_end_, nullptr, O2, pair_inner_ok_ > 0, false, nullptr, xw ? 2 : 0))) return rc;
        if (io) {
            wv::PrePostArgs<Real> pp = pre_post_args(O2, slot + 2, true, signal_pos + 2, source_live);
            pp.flag = nullptr;
            hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
        }
    }
    if ((rc = end_part_timing(1, token))) return rc;
    // level 3: every shell node from the finished t+2 field -- and the exact error bits of what the march was the last to write, should it
    // have seen an inf or a nan (fields t-1 and t are still what the march read) --, then the boundary nodes
    token = begin_part_timing(3);
    if ((rc = launch_triple_list(slot, A, B, O1, O2, O3, source_live))) return rc;
    if ((rc = end_part_timing(3, token))) return rc;
    token = begin_part_timing(2);
    if (fuse && fuse_next && (!xw || io_nodes_clear_of_x_walls())) {
        // what follows reads its source / receiver nodes from the t+3 field: none of them is a boundary node (nor, with the x-facing walls
        // on their copies, one of the two nodes such an entry finishes)
        wv::PrePostArgs<Real> nx = pre_post_args(O3, slot + 3, true, signal_pos + 3, source_live);
        if (fuse_next == 2) nx.flag2 = flags_ + slot + 4;
        if ((rc = launch_boundary(O1, O2, flag3, z_begin_, z_end_, &nx, O3, false, false, nullptr, xw ? 3 : 0))) return rc;
        pre_post_done_ = true;
    } else if ((rc = launch_boundary(O1, O2, flag3, z_begin_, z_end_, nullptr, O3, false, false, nullptr, xw ? 3 : 0))) {
        return rc;
    }
    if ((rc = end_part_timing(2, token))) return rc;
    pass_timed_ = false;
    WV_HIP(hipGetLastError());
    ++triples_taken_;
    // roles: (previous, current) = (t+2, t+3); the fields that held t-1 and t are the spares now
    const int a_idx = prv_, b_idx = cur_;
    prv_ = spare_[0];
    cur_ = spare_[1];
    spare_[0] = a_idx;
    spare_[1] = b_idx;
    return WV_OK;
}

// ---- three-step passes of a z-slab -----------------------------------------------------------------------------------------------
// With f = a face plane (a neighbour mirrors it as its ghost plane), n = the owned plane next to it, g = the ghost plane beyond it: from
// the ghost's t alone the march can produce t+1 from f on, t+2 from n on, t+3 from the plane after n on -- so it marches
// [triple_z0_, triple_z1_), stores t+2 on n as well (TripleArgs::z2_lo / z2_hi), and f and n take plain steps (launch_faces: sweep + their
// boundary nodes, as in a two-step pass), each level as soon as the neighbour's face of the level before is here.  Three exchanges per
// pass, each enqueued ahead of some other work of about its length:
//   part 0  [ghosts of t]  source / receivers on t -> f, n to t+1 -> march -> EXCHANGE 1 (t+1 faces) -> boundary nodes of its planes to t+1
//   part 1  [ghosts of t+1]  source / receivers on t+1 -> f to t+2 -> EXCHANGE 2 (t+2 faces) -> second level's list, boundary nodes from
//           n on to t+2 (they finish the nodes they face)
//   part 2  [ghosts of t+2]  source / receivers on t+2 -> f, n to t+3 -> EXCHANGE 3 (t+3 faces) -> third level's list, boundary nodes of the
//           march's planes to t+3
// (an in-process chain enqueues part k of every slab before part k + 1 of any: each part opens with a wait for pushes that must have been
// enqueued by then -- comm.h, local transport.)
// The t+1 field is the engine's fifth (field1_), which the communicator does not know: its faces travel in the face / ghost planes of the
// t+3 field, which nobody reads or writes before part 2 (the march's stores start two planes further in) -- one plane-sized copy either
// side of the exchange.
// A source on a face plane: the face travels as computed, before the level's sample goes in -- the neighbour that holds the plane as its ghost
// adds the sample to its copy itself (as in every other form of step), and every level's source / receiver work comes after the wait for
// that level's ghosts, which also covers this slab's own push of the plane the sample goes into.
// Not here: anything riding in anything (batch_can_fuse_ is off for slabs).
template <typename Real>
int Engine<Real>::enqueue_triple_slab(int slot, int part, uint64_t signal_pos, bool source_live) {
    DeviceGuard guard(device_);
    Real* A = field_[prv_];
    Real* B = field_[cur_];
    Real* O1 = field1_;
    Real* O2 = field_[spare_[0]];
    Real* O3 = field_[spare_[1]];
    int* flag1 = flags_ + slot;
    int* flag2 = flags_ + slot + 1;
    int* flag3 = flags_ + slot + 2;
    int rc;
    std::string cerr;
    const bool io = n_recv_ || source_live;
    const bool xw = triple_xw_ && xw_active_;
    const size_t plane = (size_t)pitch_ * ny_;
    const int z0 = triple_z0_, z1 = triple_z1_;
    const int n0 = z0 - (opt_.ghost_lo ? 1 : 0), n1 = z1 + (opt_.ghost_hi ? 1 : 0);  // ... and the planes next to the faces
    auto pre_post = [&](Real* field, int step) {
        if (!io) return;
        wv::PrePostArgs<Real> pp = pre_post_args(field, slot + step, true, signal_pos + (uint64_t)step, source_live);
        pp.flag = nullptr;  // (the batch's flag words were reset in one go: plan_batch)
        hipLaunchKernelGGL(wv::pre_post_kernel<Real>, dim3(1), dim3(64), 0, stream_, pp);
    };
    auto wait_for = [&](int field) -> int {
        const int token = begin_halo_wait_timing();
        if (!comm_->wait_ghosts(stream_, field, &cerr)) return fail(WV_E_COMM, cerr);
        return end_halo_wait_timing(token);
    };
    // plane z of `src` into plane z of `dst`, on the compute stream
    auto copy_plane = [&](Real* dst, const Real* src, int z) -> int {
        WV_HIP(hipMemcpyAsync(dst + (size_t)z * plane, src + (size_t)z * plane, plane * sizeof(Real), hipMemcpyDeviceToDevice, stream_));
        return WV_OK;
    };
    if (!batch_flags_reset_) return fail(WV_E_STATE, "a slab's three-step pass without the batch's flag words reset");
    if (part == 0) {
        if ((rc = wait_for(cur_))) return rc;
        pre_post(B, 0);
        pre_post_done_ = false;
        if ((rc = launch_faces(A, B, flag1, O1, 2))) return rc;
        WV_HIP(hipGetLastError());
        if (opt_.ghost_lo && (rc = copy_plane(O3, O1, z_begin_))) return rc;
        if (opt_.ghost_hi && (rc = copy_plane(O3, O1, z_end_ - 1))) return rc;
        // (the exchange behind the march, under the boundary launch that follows: enqueued ahead of the march its copy kernels sat on the
        // device for the whole of it -- 3.2 ms for 8 MB -- and the march of 508 planes took 3.3 ms where half the single domain's is 2.9)
        const bool ahead = opt_.tuning.slab_early == 2;  // (measurement: the first form)
        if (ahead && !comm_->exchange_faces(stream_, spare_[1], &cerr)) return fail(WV_E_COMM, cerr);
        if (!comm_->bulk_begin(stream_, &cerr)) return fail(WV_E_COMM, cerr);  // (slabs of one device take turns at the march)
        if ((rc = launch_triple_march(slot, A, B, O1, O2, O3))) return rc;
        if (!comm_->bulk_end(stream_, &cerr)) return fail(WV_E_COMM, cerr);
        if (!ahead && !comm_->exchange_faces(stream_, spare_[1], &cerr)) return fail(WV_E_COMM, cerr);
        if (xw && !xw_valid_) {  // the x-facing walls' compact copies, from fields t-1 and t
            wv::BoundaryArgs<Real> g = boundary_args(A, B, flag1);
            xwall_args(g);
            hipLaunchKernelGGL(wv::xwall_gather_kernel<Real>, dim3(g.xw_pad / 256), dim3(256), 0, stream_, g);
        }
        xw_valid_ = xw;
        const int token = begin_part_timing(0);
        if ((rc = launch_boundary(A, B, flag1, z0, z1, nullptr, O1, false, false, nullptr, xw ? 1 : 0))) return rc;
        if ((rc = end_part_timing(0, token))) return rc;
    } else if (part == 1) {
        if ((rc = wait_for(spare_[1]))) return rc;
        if (opt_.ghost_lo && (rc = copy_plane(O1, O3, 0))) return rc;
        if (opt_.ghost_hi && (rc = copy_plane(O1, O3, nz_ - 1))) return rc;
        pre_post(O1, 1);
        if ((rc = launch_faces(B, O1, flag2, O2, 1))) return rc;
        WV_HIP(hipGetLastError());
        if (!comm_->exchange_faces(stream_, spare_[0], &cerr)) return fail(WV_E_COMM, cerr);
        if ((rc = launch_fixup(0, pair_list_n_, O1, B, O2, flag2))) return rc;
        const int token = begin_part_timing(1);
        if ((rc = launch_boundary(B, O1, flag2, n0, n1, nullptr, O2, pair_inner_ok_ > 0, false, nullptr, xw ? 2 : 0))) return rc;
        if ((rc = end_part_timing(1, token))) return rc;
    } else {
        if ((rc = wait_for(spare_[0]))) return rc;
        pre_post(O2, 2);
        if ((rc = launch_faces(O1, O2, flag3, O3, 2))) return rc;
        WV_HIP(hipGetLastError());
        if (!comm_->exchange_faces(stream_, spare_[1], &cerr)) return fail(WV_E_COMM, cerr);
        int token = begin_part_timing(3);
        if ((rc = launch_triple_list(slot, A, B, O1, O2, O3, source_live))) return rc;
        if ((rc = end_part_timing(3, token))) return rc;
        token = begin_part_timing(2);
        if ((rc = launch_boundary(O1, O2, flag3, z0, z1, nullptr, O3, false, false, nullptr, xw ? 3 : 0))) return rc;
        if ((rc = end_part_timing(2, token))) return rc;
        pass_timed_ = false;
        WV_HIP(hipGetLastError());
        if (!comm_->step_done(stream_, &cerr)) return fail(WV_E_COMM, cerr);
        ++triples_taken_;
        const int a_idx = prv_, b_idx = cur_;
        prv_ = spare_[0];
        cur_ = spare_[1];
        spare_[0] = a_idx;
        spare_[1] = b_idx;
    }
    return WV_OK;
}

// Can this engine take three-step passes in the batch being planned (after batch_pair_prepare said yes to passes)?  Builds what they need.
template <typename Real>
int Engine<Real>::batch_triple_prepare(int* ready) {
    DeviceGuard guard(device_);
    *ready = 0;
    if (!triple_eligible()) return WV_OK;
    const int rc = ensure_triple();
    if (rc == WV_E_HIP && wv::last_hip_error() == hipErrorOutOfMemory) {  // (no room: two-step passes)
        (void)hipGetLastError();
        triple_failed_ = true;
        return WV_OK;
    }
    if (rc) return rc;
    *ready = triple_ready_ ? 1 : 0;
    if (*ready && suspect_) WV_HIP(hipMemsetAsync(suspect_, 0, kRing * sizeof(int), stream_));
    return WV_OK;
}

// One part (0, 1, 2) of the three-step pass that starts at step i of the batch -- a slab's; a single domain's pass is one part
template <typename Real>
int Engine<Real>::enqueue_batch_triple(uint64_t i, int part) {
    if (comm_) return enqueue_triple_slab((int)i, part, signal_pos_ + i, batch_source_live_);
    return part == 0 ? enqueue_triple((int)i, signal_pos_ + i, batch_source_live_, 0) : WV_OK;
}

}  // namespace wv
