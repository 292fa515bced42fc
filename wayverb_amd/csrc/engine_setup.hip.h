// engine_setup.hip.h -- set-up half of `run` (waveguide.h:43-76): device buffers, class map, boundary entry lists, sweep plan, work lists.
//
// Part of the engine behind the C ABI of include/wayverb_amd.h (engine.hip is the translation unit; see engine.hip.h for
// the class and the map of which file holds what).
#pragma once
#include "engine.hip.h"

namespace wv {

template <typename Real>
int Engine<Real>::init(const wv_mesh& m, const wv_options& opt) {
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
    // (The halo stream at a higher priority than the compute stream was tried in round 4: one rank per GPU gains nothing -- what the
    // stream carries waits for the march's workgroups to retire either way -- and several slabs on ONE device lose a quarter, their
    // halo work cutting into each other's marches: 8 x 128 planes +32 % instead of +10 %.  HISTORY.md.)
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
    if (n_entries_ && opt_.tuning.boundary_order != 0) {
        std::vector<uint32_t> bnode(n_entries_);
        std::vector<uint8_t> btype(n_entries_);
        WV_HIP(hipMemcpy(bnode.data(), bnode_, (size_t)n_entries_ * sizeof(uint32_t), hipMemcpyDeviceToHost));
        WV_HIP(hipMemcpy(btype.data(), btype_, (size_t)n_entries_, hipMemcpyDeviceToHost));
        const uint32_t nd[3] = {n1_, n2_, n3_};
        // 1-D entries that face along x, in the planes a two-step pass marches, go first: in such passes they work on
        // compact copies indexed by position instead of gathering from the fields (boundary_kernels.hip.h, xwall_node)
        std::vector<uint8_t> eligible(std::max<uint32_t>(n1_, 1), 0);
        if (n1_ && opt_.tuning.boundary_xwall != 0) {
            ScopedDevice flags_mem;
            WV_HIP(hipMalloc(&flags_mem.p, n1_));
            wv::XwEligibleArgs x{};
            x.bnode = bnode_;
            x.btype = btype_;
            x.cls = cls_;
            x.eligible = static_cast<uint8_t*>(flags_mem.p);
            x.n1 = n1_;
            x.nx = nx_;
            x.ny = ny_;
            x.nz = nz_;
            x.pitch = pitch_;
            x.cls_pitch = cls_pitch_;
            // (a slab: not in the face planes, which the march does not produce, nor -- when its passes may step the planes next
            // to the faces ahead of the march, wv_tuning::slab_early -- in those: whoever steps a plane to t+1 outside the pass's
            // two big boundary launches does it by gathering from the fields, and the copies of its entries would go stale)
            const int off = opt_.tuning.slab_early != 0 ? 2 : 1;
            x.march_begin = z_begin_ + (opt_.ghost_lo ? off : 0);
            x.march_end = z_end_ - (opt_.ghost_hi ? off : 0);
            hipLaunchKernelGGL(wv::xwall_eligible_kernel, dim3((n1_ + 255) / 256), dim3(256), 0, stream_, x);
            WV_HIP(hipGetLastError());
            WV_HIP(hipMemcpyAsync(eligible.data(), flags_mem.p, n1_, hipMemcpyDeviceToHost, stream_));
            WV_HIP(hipStreamSynchronize(stream_));
            for (uint32_t k = 0; k < n1_; ++k) n_xw_ += eligible[k];
            if (n_xw_ >= 0x7FFFFFFFu) n_xw_ = 0;  // (bit 31 of a neighbour reference is a flag)
        }
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
                if (!(d == 0 && n_xw_ && eligible[k])) kk |= 1ull << 57;
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
template <typename Real>
int Engine<Real>::set_tuning(int variant, int ry, int nwx, int nwy, int zchunks) {
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
template <typename Real>
int Engine<Real>::build_plane_order() {
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
    // the same without the first n_xw_ entries (two-step passes take those by position: launch_boundary)
    plane_start_rest_.assign((size_t)nz_ + 1, 0);
    for (uint32_t e = n_xw_; e < n_entries_; ++e)
        if (bnode[e] != wv::INVALID_NODE) ++plane_start_rest_[bnode[e] / plane + 1];
    for (int z = 0; z < nz_; ++z) plane_start_rest_[z + 1] += plane_start_rest_[z];
    std::vector<uint32_t> rest(std::max<uint32_t>(plane_start_rest_[nz_], 1));
    cursor.assign(plane_start_rest_.begin(), plane_start_rest_.end() - 1);
    for (uint32_t e = n_xw_; e < n_entries_; ++e)
        if (bnode[e] != wv::INVALID_NODE) rest[cursor[bnode[e] / plane]++] = e;
    // the entries of a slab's face plane(s) as one list (launch_faces)
    std::vector<uint32_t> face;
    {
        const int lo = opt_.ghost_lo ? 1 : 0, hi = opt_.ghost_hi ? 1 : 0;
        const int zi0 = std::min(z_begin_ + lo, z_end_), zi1 = std::max(z_end_ - hi, zi0);
        face.assign(order.begin() + plane_start_[z_begin_], order.begin() + plane_start_[zi0]);
        face.insert(face.end(), order.begin() + plane_start_[zi1], order.begin() + plane_start_[z_end_]);
        face_n_ = (uint32_t)face.size();
        if (face.empty()) face.push_back(0);
    }
    // ... and of the TWO planes next to each neighbour (a two-step pass that steps them ahead of its march: engine_pair.hip.h)
    std::vector<uint32_t> early;
    {
        const int lo = opt_.ghost_lo ? 2 : 0, hi = opt_.ghost_hi ? 2 : 0;
        const int zi0 = std::min(z_begin_ + lo, z_end_), zi1 = std::max(z_end_ - hi, zi0);
        early.assign(order.begin() + plane_start_[z_begin_], order.begin() + plane_start_[zi0]);
        early.insert(early.end(), order.begin() + plane_start_[zi1], order.begin() + plane_start_[z_end_]);
        early_n_ = (uint32_t)early.size();
        if (early.empty()) early.push_back(0);
    }
    uint32_t* staged = nullptr;
    WV_HIP(hipMalloc((void**)&staged, (order.size() + rest.size() + face.size() + early.size()) * sizeof(uint32_t)));
    if (hipMemcpy(staged, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(staged + order.size(), rest.data(), rest.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(staged + order.size() + rest.size(), face.data(), face.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(staged + order.size() + rest.size() + face.size(), early.data(), early.size() * sizeof(uint32_t), hipMemcpyHostToDevice) !=
            hipSuccess) {
        (void)hipFree(staged);
        return fail(WV_E_HIP, "copying the plane order of the boundary entries to the device failed");
    }
    zorder_ = staged;
    zorder_rest_ = staged + order.size();
    face_order_ = zorder_rest_ + rest.size();
    early_order_ = face_order_ + face.size();
    return WV_OK;
}

// variant 2 (default): plane sweep, L2-resident z reuse; 0: register z-march; 1: naive
template <typename Real>
void Engine<Real>::plan_stream() {
    lists_built_ = false;  // tile shapes may change
    duties_known_ = false;  // (... and with them the workgroup that owns a source / receiver node: whole_step_ready)
    if (graph_exec_) {
        // a captured batch holds the old plan's launches: their work list is about to be freed, their duty list to be rewritten
        (void)hipStreamSynchronize(stream_);
        (void)hipGraphExecDestroy(graph_exec_);
        graph_exec_ = nullptr;
    }
    StreamPlan& p = plan_;
    constexpr int VX = 16 / (int)sizeof(Real);
    constexpr int WX = 64 * VX;
    p.variant = tune_variant_ >= 0 ? tune_variant_ : opt_.stream_variant;
    if (p.variant < 0 || p.variant > 3) p.variant = 2;
    p.ry = tune_ry_ > 0 ? tune_ry_ : (opt_.tuning.stream_ry > 0 ? opt_.tuning.stream_ry : 4);
    // measured best shape (profiles/r01/variant_scan_*): 1 x 4 waves (a 4-wave column shares its
    // y halos through LDS) in both precisions
    p.nwx = tune_nwx_ > 0 ? tune_nwx_ : (opt_.tuning.stream_nwx > 0 ? opt_.tuning.stream_nwx : 1);
    p.nwy = tune_nwy_ > 0 ? tune_nwy_ : (opt_.tuning.stream_nwy > 0 ? opt_.tuning.stream_nwy : 4);
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
    const int64_t knob = tune_zchunks_ > 0 ? tune_zchunks_ : opt_.tuning.stream_zchunks;
    if (p.variant == 2 || p.variant == 3) {
        // stripe height: three `cur` planes of a stripe should sit comfortably in one XCD's
        // 4 MiB L2 (measured best at 0.75-1.5 MiB), at least 8 stripes so every XCD has one
        const int tile_rows = p.ry * p.nwy;
        int64_t rows = knob > 0 ? knob : (int64_t)(1600 * 1024) / (3ll * pitch_ * (int64_t)sizeof(Real));
        if (knob > 0) {
            rows = std::max<int64_t>(tile_rows, rows / tile_rows * tile_rows);  // explicit: any whole number of tiles
        } else {
            // Stripes are dealt to the 8 XCDs in rounds, and a round takes as long as a full stripe: the height (whole
            // tiles, within the L2 budget) that wastes the fewest stripe slots, the taller of equals.  (Until round 3 this
            // was the largest power of two: 768 rows -> 12 stripes of 64 -> the second round ran on 4 XCDs of 8, which
            // is what made single steps at 768^3 slower than at 512^3 and 1024^3 -- 48 rows make 16 stripes.)
            const int64_t budget = std::max<int64_t>(rows, tile_rows);
            double best = -1.0;
            for (int64_t r = tile_rows; r <= budget; r += tile_rows) {
                const int64_t stripes = (ny_ + r - 1) / r, rounds = (stripes + 7) / 8;
                const double used = (double)ny_ / (double)(rounds * 8 * r);
                if (used >= best - 1e-12) {
                    best = used;
                    rows = r;
                }
            }
        }
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

// the pressure update of planes [z0, z1)
// Work lists for the plane sweep (variant 2).  A workgroup tile takes part only if it holds an
// inside or re-entrant node: outside nodes are 0 and stay 0, boundary nodes belong to the
// boundary kernel.  Whole stripes are dealt to the 8 XCDs heaviest first (each XCD still
// sweeps its stripes plane by plane, so the z reuse in its L2 is unchanged); a mesh that is
// almost all room (a box) keeps the arithmetic mapping.
template <typename Real>
int Engine<Real>::build_tile_lists(int z0, int z1) {
    if (lists_built_) return WV_OK;
    lists_built_ = true;
    lists_z0_ = z0;
    lists_z1_ = z1;
    if (tile_list_) {
        (void)hipFree(tile_list_);
        tile_list_ = nullptr;
    }
    if ((plan_.variant != 2 && plan_.variant != 3) || !use_work_lists()) return WV_OK;
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

template <typename Real>
void Engine<Real>::release() {
    DeviceGuard guard(device_);
    if (comm_ && comm_->dead()) {
        // the watchdog gave up on this rank's peers: its streams may never drain, and hipStreamSynchronize / hipFree would wait for
        // them.  Everything is left to the end of the process, which is what a caller does after WV_E_COMM.
        comm_.release();
        return;
    }
    comm_.reset();
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& e : events_) (void)hipEventDestroy(e);
    events_.clear();
    for (auto& e : halo_events_)
        if (e) (void)hipEventDestroy(e);
    halo_events_.clear();
    for (auto& ev : part_events_) {
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        ev.clear();
    }
    for (int i = 0; i < 4; ++i)
        if (field_[i]) (void)hipFree(field_[i]);
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    for (auto& f : ckpt_.field)
        if (f) (void)hipFree(f);
    if (ckpt_.fmem) (void)hipFree(ckpt_.fmem);
    void* ptrs[] = {field1_, triple_map_, triple_list_, triple_units_, suspect_, pair_units_, pair_map_, pair_list_, pair_counter_, signal_base_dev_, tile_list_, xw_nbr_, xw_val_, xw_gok_, ref_to_pos_, cls_,   bnode_,      btype_,    fmem_,  cidx_,
                    status_,          coeffs_,    flags_,      scratch_, signal_, recv_nodes_, recv_out_, zorder_};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (flags_host_) (void)hipHostFree(flags_host_);
    if (recv_stage_) (void)hipHostFree(recv_stage_);
    if (duties_) (void)hipFree(duties_);
    if (stream_) (void)hipStreamDestroy(stream_);
    if (comm_stream_) (void)hipStreamDestroy(comm_stream_);
}

}  // namespace wv
