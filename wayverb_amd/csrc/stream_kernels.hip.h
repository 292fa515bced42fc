// stream_kernels.hip.h -- the per-step pressure update for every non-boundary node.
//
// Replaces the `normal_waveguide_update` arm of `condensed_waveguide`
// (src/waveguide/src/program.cpp:393-412, :494-530) and its id_none arm (:485).  Boundary nodes
// (class 2) are left to boundary_kernels.hip.h, which runs after this kernel (see X_STORE_ALL).
//
// Arithmetic per updated node, in the pressure type Real, no FMA contraction:
//     s = 0; s += nx; s += px; s += ny; s += py; s += nz; s += pz   (off-grid ports contribute +0,
//     s = s / 3;  next = s - prev                                    which equals skipping them)
//
// Field layout: Real[nz][ny][pitch], pitch = nx rounded up to the wave tile width (64 lanes x
// 16 B); the pad columns are class "none", so they hold 0 forever: every row access is a full,
// 16-byte-aligned, 1 KiB wave transaction with no ragged-edge code, and x = nx reads as the
// off-grid zero it stands for.
//
// Bound: HBM.  Algorithmic traffic per node = read prev + read cur + write next = 3*sizeof(Real).
#pragma once
#include "device_common.hip.h"

namespace wv {

// Experiment switches (tools/stream_bench.hip only).  The first group changes results and exists
// to price a piece of the kernel; the nt ones are result-neutral cache hints.
enum : int { X_NO_EDGE = 1, X_NO_CLS = 2, X_MUL_THIRD = 4, X_NT_STORE = 8, X_NT_PREV = 16, X_NT_CUR = 32,
             X_NO_HALO_ROWS = 64, X_TX_FAST = 128, X_NT_BELOW = 256, X_NT_MID = 512,
             X_STORE_ALL = 1024,    // boundary nodes get their old value written back: no masked stores
             X_DUTIES = 2048 };     // one-launch steps: the tile serves the next step's source / receiver samples at its nodes (StepDuties)
// what the engine runs: `prev` and `next` are touched exactly once per step, so they carry the
// non-temporal hint and do not displace the re-used `cur` lines from L2
constexpr int X_PRODUCT = X_TX_FAST | X_NT_STORE | X_NT_PREV;

template <typename V, bool NT>
__device__ __forceinline__ V load_vec(const V* p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <typename V, bool NT>
__device__ __forceinline__ void store_vec(V* p, V v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// The 7-point update of one lane's VX nodes of one row.  c0 = this row of `cur`; ym / yp / zm / zp
// the rows at y-1, y+1, z-1, z+1; edges = x-edge halos (see load_edges); r = row slot in `edges`.
// Returns the values to store; `skip` is set if any of the lane's nodes is a boundary node.
template <typename Real, int X>
__device__ __forceinline__ typename Vec16<Real>::type update_row(
        typename Vec16<Real>::type c0, typename Vec16<Real>::type ym, typename Vec16<Real>::type yp,
        typename Vec16<Real>::type zm, typename Vec16<Real>::type zp, typename Vec16<Real>::type pv, Real edges, int r,
        uint32_t cls_bits, int& bad, bool& skip) {
    constexpr int VX = Vec16<Real>::N;
    const Real edge_l = read_lane(edges, r), edge_r = read_lane(edges, 32 + r);
    typename Vec16<Real>::type out;
#pragma unroll
    for (int j = 0; j < VX; ++j) {
        const Real left = (j == 0) ? lane_from_below(edge_l, c0[VX - 1]) : c0[j - 1];
        const Real right = (j == VX - 1) ? lane_from_above(edge_r, c0[0]) : c0[j + 1];
        Real s = Real(0) + left;
        s += right;
        s += ym[j];
        s += yp[j];
        s += zm[j];
        s += zp[j];
        s = (X & X_MUL_THIRD) ? s * (Real(1) / Real(3)) : div3(s);
        s -= pv[j];
        const uint32_t c = (cls_bits >> (2 * j)) & 3u;
        const Real o = (c & 1u) ? s : Real(0);
        bad |= bad_bits(o);
        if (X & X_STORE_ALL) {
            out[j] = (c == CLS_BOUNDARY) ? pv[j] : o;  // boundary node: its old value goes back
        } else {
            out[j] = o;
            skip |= (c == CLS_BOUNDARY);
        }
    }
    return out;
}

// Store one lane's VX results; per-element only where the wave holds a boundary node.
template <typename Real, int X>
__device__ __forceinline__ void store_row(Real* q, typename Vec16<Real>::type out, uint32_t cls_bits, bool skip) {
    using V = typename Vec16<Real>::type;
    constexpr int VX = Vec16<Real>::N;
    if ((X & X_STORE_ALL) || !__any(skip)) {
        store_vec<V, (X & X_NT_STORE) != 0>(reinterpret_cast<V*>(q), out);
    } else {
#pragma unroll
        for (int j = 0; j < VX; ++j)
            if (((cls_bits >> (2 * j)) & 3u) != CLS_BOUNDARY) q[j] = out[j];
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 1 ("naive"): one node per lane, neighbours straight from global memory (the caches do
// the reuse).  The simple baseline the tuned kernels are checked and timed against.
// ---------------------------------------------------------------------------------------------
template <typename Real>
__global__ void __launch_bounds__(256) stream_naive_kernel(const StreamArgs<Real> a) {
    const int64_t row_nodes = a.nx;
    const int64_t plane_nodes = row_nodes * a.ny;
    const int64_t count = (int64_t)(a.z_end - a.z_begin) * plane_nodes;
    const int64_t plane = (int64_t)a.pitch * a.ny;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.nx);
        const int64_t q = i / a.nx;
        const int y = (int)(q % a.ny);
        const int z = a.z_begin + (int)(q / a.ny);
        const int64_t row = (int64_t)z * a.ny + y;
        const int64_t idx = row * a.pitch + x;
        const uint32_t c = (a.cls[cls_byte_index(x, y, z, a.ny, a.cls_pitch)] >> ((x & 3) * 2)) & 3u;
        if (c == CLS_BOUNDARY) continue;
        Real out = 0;
        if (c & 1u) {
            Real s = 0;
            s += (x > 0) ? a.cur[idx - 1] : Real(0);
            s += (x + 1 < a.nx) ? a.cur[idx + 1] : Real(0);
            s += (y > 0) ? a.cur[idx - a.pitch] : Real(0);
            s += (y + 1 < a.ny) ? a.cur[idx + a.pitch] : Real(0);
            s += (z > 0) ? a.cur[idx - plane] : Real(0);
            s += (z + 1 < a.nz) ? a.cur[idx + plane] : Real(0);
            s = div3(s);
            s -= a.prev[idx];
            out = s;
            bad |= bad_bits(out);
        }
        a.next[idx] = out;
    }
    if (bad) atomicOr(a.flag, bad);
}

// Shared lane-level accessors of the two tiled kernels.
template <typename Real>
struct TileIO {
    using V = typename Vec16<Real>::type;
    static constexpr int VX = Vec16<Real>::N;
    static constexpr int WX = 64 * VX;
    const StreamArgs<Real>& a;
    int lane, x0, xl;
    int64_t plane;

    __device__ __forceinline__ TileIO(const StreamArgs<Real>& args, int lane_, int x0_)
            : a(args), lane(lane_), x0(x0_), xl(x0_ + lane_ * VX), plane((int64_t)args.pitch * args.ny) {}
    __device__ __forceinline__ int64_t at(int y, int z) const { return (int64_t)z * plane + (int64_t)y * a.pitch + xl; }

    // a row of `cur`: zeros when the row or plane is off-grid (wave-uniform test)
    template <bool NT>
    __device__ __forceinline__ V cur_row(int y, int z) const {
        if (y < 0 || y >= a.ny || z < 0 || z >= a.nz) return (V)(Real(0));
        return load_vec<V, NT>(reinterpret_cast<const V*>(a.cur + at(y, z)));
    }
    template <bool NT>
    __device__ __forceinline__ V prev_row(int y, int z) const {
        return load_vec<V, NT>(reinterpret_cast<const V*>(a.prev + at(y, z)));
    }
    // x-edge halos of rows y0..y0+RY-1 of a plane in ONE load: lane r (< RY) fetches cur[x0-1] of
    // row y0+r, lane 32+r fetches cur[x0+WX] of that row; read_lane hands them out later.
    template <int RY>
    __device__ __forceinline__ Real edges(int y0, int z) const {
        Real e = 0;
        const int r = lane & 31;
        const int y = y0 + r;
        if (r < RY && y < a.ny && z >= 0 && z < a.nz) {
            const int xe = (lane < 32) ? x0 - 1 : x0 + WX;
            if (xe >= 0 && xe < a.pitch) e = a.cur[(int64_t)z * plane + (int64_t)y * a.pitch + xe];
        }
        return e;
    }
    // class bits of this lane's VX nodes for the 4 rows of row group y >> 2: ONE dword load
    __device__ __forceinline__ uint32_t cls_word(int y, int z) const {
        return reinterpret_cast<const uint32_t*>(a.cls)[cls_word_index(xl, y, z, a.ny, a.cls_pitch)];
    }
    // ... and the 2 bits per node of row y out of that word
    __device__ __forceinline__ uint32_t cls_of_row(uint32_t word, int y) const {
        const uint32_t byte = (word >> ((y & 3) * 8)) & 0xFFu;
        return (VX == 4) ? byte : ((byte >> ((lane & 1) * 4)) & 0xFu);
    }
};

// ---------------------------------------------------------------------------------------------
// Variant 0 ("march"): wave-autonomous 2.5-D march, z-planes of `cur` rotate through registers.
//
// One wave owns a 64*VX by RY tile and marches it through `zc` planes, prefetching plane z+2 while
// it computes plane z; every `cur` value is fetched from HBM once and nothing but halos is
// re-read.  Measured ceiling on MI355X: ~66 % of peak -- thousands of long-lived private streams
// make a DRAM-unfriendly access front.  Kept as a cross-check of variant 2.
// ---------------------------------------------------------------------------------------------
template <typename Real, int RY, int NWX, int NWY, int X = X_PRODUCT>
__global__ void __launch_bounds__(64 * NWX * NWY) stream_march_kernel(const StreamArgs<Real> a) {
    using V = typename Vec16<Real>::type;
    constexpr int WX = TileIO<Real>::WX;
    constexpr bool NTP = (X & X_NT_PREV) != 0, NTC = (X & X_NT_CUR) != 0;

    const int lane = threadIdx.x & 63;
    // wave id as a scalar: everything derived from it (tile origin, row addresses) stays in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wx = wave % NWX, wy = wave / NWX;

    // XCD-aware tile map: workgroup b runs on XCD b % 8 (observed dispatch, used for locality only)
    const int b = blockIdx.x;
    const int slot = b >> 3;
    const int t = (b & 7) * a.tiles_per_xcd + slot;
    if (slot >= a.tiles_per_xcd || t >= a.total_tiles) return;
    int tx, ty, cz;
    if (X & X_TX_FAST) {  // x-adjacent tiles consecutive: an XCD's resident set is a tx-by-ty patch
        tx = t % a.tiles_x;
        const int rem = t / a.tiles_x;
        ty = rem % a.tiles_y;
        cz = rem / a.tiles_y;
    } else {
        ty = t % a.tiles_y;
        const int rem = t / a.tiles_y;
        tx = rem % a.tiles_x;
        cz = rem / a.tiles_x;
    }
    const int x0 = (tx * NWX + wx) * WX;
    const int y0 = (ty * NWY + wy) * RY;
    if (y0 >= a.ny || x0 >= a.pitch) return;
    const int zb = a.z_begin + cz * a.zc;
    const int ze = min(zb + a.zc, a.z_end);
    if (zb >= ze) return;

    const TileIO<Real> io(a, lane, x0);
    auto halo_or_cur = [&](int r, int z) -> V {  // r in [0, RY+2): rows y0-1 .. y0+RY
        if ((X & X_NO_HALO_ROWS) && (r == 0 || r == RY + 1)) return (V)(Real(0));
        return io.template cur_row<NTC>(y0 - 1 + r, z);
    };
    auto load_edges = [&](int z) -> Real { return (X & X_NO_EDGE) ? Real(0) : io.template edges<RY>(y0, z); };

    V below[RY];      // cur(z-1), rows y0 .. y0+RY-1
    V mid[RY + 2];    // cur(z),   rows y0-1 .. y0+RY
    V above[RY + 2];  // cur(z+1)
    Real mid_e, above_e;
    V pv[RY];
    uint32_t cl[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) below[r] = io.template cur_row<NTC>(y0 + r, zb - 1);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) mid[r] = halo_or_cur(r, zb);
    mid_e = load_edges(zb);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) above[r] = halo_or_cur(r, zb + 1);
    above_e = load_edges(zb + 1);
    static_assert(RY <= 4 && 4 % RY == 0, "a tile's rows must sit inside one class-map row group");
    uint32_t clw = (X & X_NO_CLS) ? 0x55555555u : io.cls_word(y0, zb);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const bool live = y0 + r < a.ny;
        pv[r] = live ? io.template prev_row<NTP>(y0 + r, zb) : (V)(Real(0));
        cl[r] = live ? io.cls_of_row(clw, y0 + r) : 0xAAu;
    }

    int bad = 0;
    for (int z = zb; z < ze; ++z) {
        // ---- issue the loads of the next iteration first: plane z+2 of cur, plane z+1 of prev
        V nxt[RY + 2], pv_n[RY];
        uint32_t cl_n[RY];
#pragma unroll
        for (int r = 0; r < RY + 2; ++r) nxt[r] = halo_or_cur(r, z + 2);
        const Real nxt_e = load_edges(z + 2);
        clw = (X & X_NO_CLS) ? 0x55555555u : (z + 1 < ze ? io.cls_word(y0, z + 1) : 0u);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const bool live = y0 + r < a.ny && z + 1 < ze;
            pv_n[r] = live ? io.template prev_row<NTP>(y0 + r, z + 1) : (V)(Real(0));
            cl_n[r] = live ? io.cls_of_row(clw, y0 + r) : 0xAAu;
        }
        // ---- update plane z
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            if (y0 + r < a.ny) {
                bool skip = false;
                const V out = update_row<Real, X>(mid[r + 1], mid[r], mid[r + 2], below[r], above[r + 1], pv[r], mid_e, r,
                                                  cl[r], bad, skip);
                store_row<Real, X>(a.next + io.at(y0 + r, z), out, cl[r], skip);
            }
        }
        // ---- rotate the register planes
        mid_e = above_e;
        above_e = nxt_e;
#pragma unroll
        for (int r = 0; r < RY; ++r) below[r] = mid[r + 1];
#pragma unroll
        for (int r = 0; r < RY + 2; ++r) {
            mid[r] = above[r];
            above[r] = nxt[r];
        }
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pv[r] = pv_n[r];
            cl[r] = cl_n[r];
        }
    }
    if (__any(bad != 0)) {
        if (bad) atomicOr(a.flag, bad);
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 3 ("sweep without LDS", kept as a cross-check of variant 2): plane sweep with L2-resident
// z-reuse; every wave loads its own halo rows.  The structure below is shared by variant 2.
//
// Measured on MI355X (tools/stream_bench.hip, profiles/): HBM delivers ~6.5 TB/s to short-lived
// workgroups that are dispatched in address order (the chip-wide set of in-flight addresses is
// a few long contiguous runs), but only ~5.3 TB/s to thousands of long-lived waves that each
// own a private stream -- which is what the z-march is.  This variant keeps the march's lane
// layout (16 B per lane, DPP x-neighbours, one-load x-edges) but makes every wave short-lived:
// a wave updates its 64*VX by RY tile of ONE plane and exits.  The z-reuse of `cur` then has to
// come from cache, so the work is laid out for the per-XCD L2 (4 MiB, private):
//   - XCD k (= blockIdx % 8) owns y-stripe s = pass*8 + k, `stripe_rows` rows tall, and sweeps it
//     through all planes; the three `cur` planes of a stripe (3 * stripe_rows * pitch * 8 B) stay in
//     that XCD's L2 while prev/next stream through it with the non-temporal hint;
//   - all 8 XCDs advance through z together, so the chip-wide access front is 8 short runs
//     per stream inside one plane.
// `cur` is then fetched from HBM once per step (plus 2 halo rows per stripe).
//
// Stores are always full 16-byte vectors: a boundary node gets its OLD `prev` value written back
// (X_STORE_ALL) instead of being masked out, which removes the per-row "does this wave hold a
// boundary node" vote and the masked-store path (measured -4 %).  The price is an ordering rule:
// this kernel must have finished a plane before the boundary kernel updates that plane's boundary
// nodes (engine.hip enforces it: sweep, then boundary, on one stream).
// ---------------------------------------------------------------------------------------------
constexpr int X_SWEEP = X_NT_STORE | X_NT_PREV | X_STORE_ALL;

template <typename Real, int RY, int NWX, int NWY, int X = X_SWEEP>
__global__ void __launch_bounds__(64 * NWX * NWY) stream_sweep_nolds_kernel(const StreamArgs<Real> a) {
    using V = typename Vec16<Real>::type;
    constexpr int WX = TileIO<Real>::WX;
    constexpr bool NTP = (X & X_NT_PREV) != 0;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wx = wave % NWX, wy = wave / NWX;

    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    int tl, z, stripe;
    if (a.tile_list) {
        // only the tiles that hold a node this kernel updates (engine.hip, build_tile_lists)
        const uint32_t first = a.list_start[xcd], count = a.list_start[xcd + 1] - first;
        if ((uint32_t)j >= count) return;
        const uint64_t e = a.tile_list[first + (uint32_t)j];
        if (!((e >> 40) & (1u << wave))) return;  // nothing to update in this wave's rows
        tl = (int)(e & 0xFFFFFu);
        z = (int)((e >> 20) & 0xFFFFFu);
        stripe = (int)(e >> 48);
    } else {
        const int per_plane = a.tiles_x * a.tiles_y_stripe;
        tl = j % per_plane;
        j /= per_plane;
        const int nzr = a.z_end - a.z_begin;
        z = a.z_begin + j % nzr;
        if (z >= a.z_skip_from) z += a.z_skip;  // (a slab's two face planes in one launch)
        stripe = (j / nzr) * 8 + xcd;
    }
    const int tx = tl % a.tiles_x, tyl = tl / a.tiles_x;

    const int y_lo = stripe * a.stripe_rows;
    const int y_hi = min(y_lo + a.stripe_rows, a.ny);
    const int x0 = (tx * NWX + wx) * WX;
    const int y0 = y_lo + (tyl * NWY + wy) * RY;
    if (y0 >= y_hi || x0 >= a.pitch) return;

    const TileIO<Real> io(a, lane, x0);

    // ---- everything this tile needs, issued back to back
    V below[RY], mid[RY + 2], above[RY], pv[RY];
    uint32_t cl[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) above[r] = io.template cur_row<(X & X_NT_CUR) != 0>(y0 + r, z + 1);  // first touch: HBM
    static_assert(RY <= 4 && 4 % RY == 0, "a tile's rows must sit inside one class-map row group");
    const uint32_t clw = (X & X_NO_CLS) ? 0x55555555u : io.cls_word(y0, z);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const bool live = y0 + r < y_hi;
        pv[r] = live ? io.template prev_row<NTP>(y0 + r, z) : (V)(Real(0));
        cl[r] = live ? io.cls_of_row(clw, y0 + r) : 0xAAu;
    }
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) mid[r] = io.template cur_row<(X & X_NT_MID) != 0>(y0 - 1 + r, z);  // L2: was z+1 a plane ago
    const Real mid_e = (X & X_NO_EDGE) ? Real(0) : io.template edges<RY>(y0, z);
#pragma unroll
    for (int r = 0; r < RY; ++r) below[r] = io.template cur_row<(X & X_NT_BELOW) != 0>(y0 + r, z - 1);  // L2: last use

    int bad = 0;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        if (y0 + r < y_hi) {
            bool skip = false;
            const V out = update_row<Real, X>(mid[r + 1], mid[r], mid[r + 2], below[r], above[r], pv[r], mid_e, r, cl[r],
                                              bad, skip);
            store_row<Real, X>(a.next + io.at(y0 + r, z), out, cl[r], skip);
        }
    }
    if (__any(bad != 0)) {
        if (bad) atomicOr(a.flag, bad);
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 2, the product kernel: the sweep above with the y halos of plane z staged through LDS.
//
// A wave needs rows y0-1 and y0+RY of the middle plane besides its own RY rows.  In the kernel
// above it loads them (L2 hits: the neighbouring wave of the same workgroup loads the same rows as
// its own).  Here every wave publishes its first and last row in LDS, one barrier, and the
// neighbours pick them up: global loads of the middle plane drop from RY+2 to RY rows per wave
// (only the workgroup's outer halo rows still come from L2).  Measured at 1024^3 fp64: 4.38 ->
// 4.15 ms (73.5 -> 77.6 % of the HBM peak): the sweep is limited by the request rate into
// L2 / the fabric, not by DRAM, so every load that does not have to be issued counts.
//
// With a work list (rooms that leave part of the mesh outside) whole workgroups are skipped by
// the list and single waves by the entry's wave mask; a skipped wave publishes nothing, so its
// neighbours load that halo row themselves.
// ---------------------------------------------------------------------------------------------
// (The body is a device function so that plane_step_kernel, boundary_kernels.hip.h, can run it beside the boundary entries of
// the same planes in one launch -- there with masked stores, X = X_SWEEP & ~X_STORE_ALL, so that it leaves boundary nodes alone.
// `block`: this workgroup's index among the sweep's workgroups.)
// One-launch steps: the duties in `mine` (bit i = entry i of d.list) belong to this workgroup's tile; `out` holds the new values of
// this lane's VX nodes from stored index `at` on.  The NEXT step's source sample goes into the value before it is stored
// (hard_source.h:21 / soft_source.h:21-24), its receivers take what the node then holds (waveguide.h:121) -- a source's duty comes
// first in the list, so a receiver on the source node reads the value with the sample in, as pre_post_body has it.
template <typename Real>
__device__ __forceinline__ void serve_duties(const StepDuties<Real>& d, uint64_t mine, int64_t at, typename Vec16<Real>::type& out) {
    constexpr int VX = Vec16<Real>::N;
    while (mine) {
        const int i = __builtin_ctzll(mine);
        mine &= mine - 1;
        const StepDuty q = d.list[i];  // (wave-uniform)
        const int64_t off = (int64_t)q.node - at;
        if (off >= 0 && off < VX) {
            Real v = 0;
#pragma unroll
            for (int j = 0; j < VX; ++j)
                if (off == j) v = out[j];
            if (q.kind) {
                const Real s = (Real)d.signal[d.signal_pos + (d.signal_base ? *d.signal_base : 0ull)];
                v = q.kind == 1 ? s : (Real)(v + s);
#pragma unroll
                for (int j = 0; j < VX; ++j)
                    if (off == j) out[j] = v;
            } else {
                d.recv_out[q.col] = v;
            }
        }
    }
}

template <typename Real, int RY, int NWX, int NWY, int X>
__device__ __forceinline__ void stream_sweep_body(const StreamArgs<Real>& a, unsigned block, const StepDuties<Real>* d = nullptr) {
    using V = typename Vec16<Real>::type;
    constexpr int WX = TileIO<Real>::WX;
    __shared__ V halo[NWY][NWX][2][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wx = wave % NWX, wy = wave / NWX;

    const int xcd = block & 7;
    int j = block >> 3;
    int tl, z, stripe;
    uint32_t mask = ~0u;  // which waves of this workgroup have something to update
    if (a.tile_list) {
        const uint32_t first = a.list_start[xcd], count = a.list_start[xcd + 1] - first;
        if ((uint32_t)j >= count) return;  // the whole workgroup leaves together
        const uint64_t e = a.tile_list[first + (uint32_t)j];
        mask = (uint32_t)(e >> 40) & 0xFFu;
        tl = (int)(e & 0xFFFFFu);
        z = (int)((e >> 20) & 0xFFFFFu);
        stripe = (int)(e >> 48);
    } else {
        const int per_plane = a.tiles_x * a.tiles_y_stripe;
        tl = j % per_plane;
        j /= per_plane;
        const int nzr = a.z_end - a.z_begin;
        z = a.z_begin + j % nzr;
        if (z >= a.z_skip_from) z += a.z_skip;  // (a slab's two face planes in one launch)
        stripe = (j / nzr) * 8 + xcd;
    }
    const int tx = tl % a.tiles_x, tyl = tl / a.tiles_x;

    const int y_lo = stripe * a.stripe_rows;
    const int y_hi = min(y_lo + a.stripe_rows, a.ny);
    const int x0 = (tx * NWX + wx) * WX;
    const int y0 = y_lo + (tyl * NWY + wy) * RY;
    auto row_wave_active = [&](int wyy) {  // wave (wx, wyy) of this workgroup: inside the stripe and not masked out
        return y_lo + (tyl * NWY + wyy) * RY < y_hi && ((mask >> (wyy * NWX + wx)) & 1u);
    };
    const bool active = x0 < a.pitch && row_wave_active(wy);  // idle waves still meet the barrier
    const bool from_lo = wy > 0 && row_wave_active(wy - 1);      // row y0-1 is wave wy-1's last row
    const bool from_hi = wy + 1 < NWY && row_wave_active(wy + 1);  // row y0+RY is wave wy+1's first row

    const TileIO<Real> io(a, lane, min(x0, a.pitch - WX));
    V below[RY], mid[RY + 2], above[RY], pv[RY];
    uint32_t cl[RY];
    Real mid_e = 0;
    uint32_t duty_block = ~0u;  // (X_DUTIES) lane i: the workgroup duty i belongs to
    if (active) {
        // ---- everything this tile needs from memory, issued back to back
        // (the rows that go through LDS first: they are needed before the barrier)
#pragma unroll
        for (int r = 0; r < RY; ++r) mid[r + 1] = io.template cur_row<false>(y0 + r, z);  // L2: was z+1 a plane ago
        if ((X & X_DUTIES) && lane < (int)d->n) duty_block = d->list[lane].block;
        if (!from_lo) mid[0] = io.template cur_row<false>(y0 - 1, z);
        if (!from_hi) mid[RY + 1] = io.template cur_row<false>(y0 + RY, z);
#pragma unroll
        for (int r = 0; r < RY; ++r) above[r] = io.template cur_row<false>(y0 + r, z + 1);  // first touch: HBM
        static_assert(RY <= 4 && 4 % RY == 0, "a tile's rows must sit inside one class-map row group");
        const uint32_t clw = io.cls_word(y0, z);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const bool live = y0 + r < y_hi;
            pv[r] = live ? io.template prev_row<true>(y0 + r, z) : (V)(Real(0));
            cl[r] = live ? io.cls_of_row(clw, y0 + r) : 0xAAu;
        }
        mid_e = io.template edges<RY>(y0, z);
#pragma unroll
        for (int r = 0; r < RY; ++r) below[r] = io.template cur_row<false>(y0 + r, z - 1);  // L2: last use
        halo[wy][wx][0][lane] = mid[1];
        halo[wy][wx][1][lane] = mid[RY];
    }
    __syncthreads();
    if (!active) return;
    if (from_lo) mid[0] = halo[wy - 1][wx][1][lane];
    if (from_hi) mid[RY + 1] = halo[wy + 1][wx][0][lane];

    int bad = 0;
    const uint64_t my_duties = (X & X_DUTIES) ? __ballot(duty_block == block) : 0ull;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        if (y0 + r < y_hi) {
            bool skip = false;
            V out = update_row<Real, X>(mid[r + 1], mid[r], mid[r + 2], below[r], above[r], pv[r], mid_e, r, cl[r], bad, skip);
            if ((X & X_DUTIES) && my_duties) serve_duties<Real>(*d, my_duties, io.at(y0 + r, z), out);
            store_row<Real, X>(a.next + io.at(y0 + r, z), out, cl[r], skip);
        }
    }
    if (__any(bad != 0)) {
        if (bad) atomicOr(a.flag, bad);
    }
}

template <typename Real, int RY, int NWX, int NWY>
__global__ void __launch_bounds__(64 * NWX * NWY) stream_sweep_kernel(const StreamArgs<Real> a) {
    stream_sweep_body<Real, RY, NWX, NWY, X_SWEEP>(a, blockIdx.x);
}

}  // namespace wv
