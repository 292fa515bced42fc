// pair_kernels.hip.h -- TWO time steps of the pressure update in one pass over the fields.
//
// Same arithmetic as stream_kernels.hip.h (`normal_waveguide_update`,
// src/waveguide/src/program.cpp:393-412, applied twice), bit for bit; what changes is the traffic.
// One step moves 3 fields per node (read previous, read current, write next = 24 B in fp64); a pass
// that produces steps t+1 AND t+2 from (t-1, t) reads two fields and writes two: 32 B per node for
// two steps instead of 48.  The sweep of stream_kernels.hip.h already runs at what the memory system
// delivers for its byte mix, so moving fewer bytes is the only thing left.
//
// Form (priced in tools/pair_bench.hip before it was built here): a workgroup owns a strip of RY = 4
// rows x the whole row (NW waves side by side) and marches it through a range of planes.  It keeps
// three planes of `current` (RY+4 rows) and three planes of t+1 (RY+2 rows) in registers; per plane it
// loads one plane of `current` (RY+4 rows) and one of `previous` (RY+2 rows), computes t+1 on RY+2
// rows of the plane ahead and t+2 on its RY rows of this plane, and stores RY rows of each.  The y
// rings are recomputed by the neighbouring strips (no dependency between workgroups); the strips of
// one XCD are neighbours and march in step, so the ring rows they share are fetched from HBM once
// and found in that XCD's L2 by the other.  x neighbours: DPP inside a wave, a few LDS words between
// the waves of the workgroup.
//
// Which nodes this kernel may finish is decided once per mesh (pair_map_kernel below) and read from
// a 2-bit map with the class map's layout:
//   0  outside the room ("none"):  t+1 = t+2 = 0
//   1  takes the 7-point update and so do all six neighbours, none of which is the source node:
//      both t+1 and t+2 are final here
//   3  takes the 7-point update but a neighbour does not (boundary node, source node, outside): t+1
//      is final, t+2 needs that neighbour's t+1 -> pair_fixup_kernel after the boundary kernel
//   2  boundary node: both levels belong to the boundary kernel (whatever is stored here is overwritten)
// The engine runs: [source/receivers on t] -> this kernel -> boundary nodes t+1 -> [source/receivers on
// t+1] -> fix-up list t+2 -> boundary nodes t+2 (engine.hip, enqueue_pair).
#pragma once
#include "device_common.hip.h"

namespace wv {

constexpr int kPairRows = 4;      // RY: rows per strip = one row group of the class map
constexpr int kPairMaxWaves = 8;  // waves side by side: rows of up to 8 * 64 * 16 B
constexpr int kPairMaxWindows = 8;  // WIDE march: workgroups side by side on longer rows (up to 8 * 6 + 2 waves)

template <typename Real>
struct PairArgs {
    const Real* prev;  // t-1
    const Real* cur;   // t
    Real* out1;        // t+1
    Real* out2;        // t+2
    const uint8_t* pair_map;
    int* flag1;        // error_code word of step t -> t+1
    int* flag2;        // ... of step t+1 -> t+2
    int ny, nz, pitch, cls_pitch;
    int z_begin, z_end;  // planes to produce
    // planes whose t+1 values are STORED, [z_begin, z_end) or one plane less at an end: on a slab the planes next to the face
    // planes get their t+1 from the launch that steps the faces (complete, boundary nodes included, before this kernel runs --
    // engine_pair.hip.h, the early-faces pass); the march still computes them, for its own t+2, but must not put its
    // placeholders over finished boundary values.  t+2 is stored on every produced plane.
    int out1_z0, out1_z1;
    int nw;              // waves per workgroup (pitch / wave tile width)
    int zc, chunks;      // planes per workgroup, workgroups along z
    int strips, strips_per_xcd;
    // optional work list (rooms that leave much of the mesh outside): workgroup j of XCD k takes unit
    // unit_list[list_start[k] + j] = strip | chunk << 16 (| first wave << 25 | waves - 1 << 28 for the march that runs
    // only the live waves of a unit); units without a node to update are not listed
    const uint32_t* unit_list;
    uint32_t list_start[9];
    // WIDE march only (rows of more than kPairMaxWaves waves): the row is covered by `windows` workgroups side by side.
    // Window k runs waves [win_first_k, win_first_k + win_count_k) of the row and stores those in
    // [win_store_lo_k, win_store_hi_k); the one wave it runs beyond either end of that range is there for its
    // t+1 values only (pair_march_kernel).
    // (byte k of each word belongs to window k: words, not arrays, so that the kernel reads them with shifts)
    int windows;
    uint64_t win_first, win_count, win_store_lo, win_store_hi;
};

template <typename Real>
struct PairTile {
    using V = typename Vec16<Real>::type;
    static constexpr int VX = Vec16<Real>::N;
    int ny, nz, pitch;
    int64_t plane;
    int col;  // first of this lane's VX columns

    __device__ __forceinline__ V load(const Real* p, int y, int z) const {
        if (y < 0 || y >= ny || z < 0 || z >= nz) return (V)(Real(0));
        return *reinterpret_cast<const V*>(p + (int64_t)z * plane + (int64_t)y * pitch + col);
    }
    __device__ __forceinline__ V load_nt(const Real* p, int y, int z) const {
        if (y < 0 || y >= ny || z < 0 || z >= nz) return (V)(Real(0));
        return __builtin_nontemporal_load(reinterpret_cast<const V*>(p + (int64_t)z * plane + (int64_t)y * pitch + col));
    }
    __device__ __forceinline__ void store_cached(Real* p, int y, int z, V v) const {
        *reinterpret_cast<V*>(p + (int64_t)z * plane + (int64_t)y * pitch + col) = v;
    }
    __device__ __forceinline__ void store(Real* p, int y, int z, V v) const {
        __builtin_nontemporal_store(v, reinterpret_cast<V*>(p + (int64_t)z * plane + (int64_t)y * pitch + col));
    }
};

// The 7-point update of one lane's VX nodes of one row, in the reference's order:
// s = 0 + left; s += right; s += ym; s += yp; s += zm; s += zp; s = s / 3; s -= prev.
// edge_l / edge_r: the values just outside this wave's columns (lane 0 / lane 63 use them).
template <typename Real>
__device__ __forceinline__ typename Vec16<Real>::type pair_step_row(
        typename Vec16<Real>::type c0, typename Vec16<Real>::type ym, typename Vec16<Real>::type yp,
        typename Vec16<Real>::type zm, typename Vec16<Real>::type zp, typename Vec16<Real>::type pv, Real edge_l,
        Real edge_r) {
    constexpr int VX = Vec16<Real>::N;
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
        s = div3(s);
        s -= pv[j];
        out[j] = s;
    }
    return out;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory, which on
// gfx9 means `s_waitcnt vmcnt(0)`: every global load and store in flight would be drained at each of
// the two barriers per plane, and the march lives on keeping a plane of loads in flight across them.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// x-edge values between the waves of the workgroup, through LDS.  A wave publishes the first and last
// column of each row it owns (lanes 0 and 63); after a barrier every wave reads what it needs when
// it needs it (broadcast reads: all lanes, one address) -- el = last column of the wave to the left,
// er = first column of the wave to the right, 0 at the ends of the row (off the grid).  Two sets used
// alternately: a set is rewritten two planes later, and the barrier of the plane in between has seen
// every wave finish reading it, so one barrier per plane suffices.
template <typename Real, int K>
struct PairEdges {
    Real (*sl)[K][kPairMaxWaves];  // [set][row][wave]: first column
    Real (*sr)[K][kPairMaxWaves];  // last column
    int lane, wave, nw;

    __device__ __forceinline__ void publish(int set, int k, typename Vec16<Real>::type row) const {
        constexpr int VX = Vec16<Real>::N;
        if (lane == 0) sl[set][k][wave] = row[0];
        if (lane == 63) sr[set][k][wave] = row[VX - 1];
    }
    __device__ __forceinline__ Real left(int set, int k) const { return wave > 0 ? sr[set][k][wave - 1] : Real(0); }
    __device__ __forceinline__ Real right(int set, int k) const { return wave + 1 < nw ? sl[set][k][wave + 1] : Real(0); }
};

// Experiment switches (tools/pair_tune.hip only; the engine runs X = 0).  They change results or drop
// checks and exist to price a piece of the kernel.
// PX_NO_MEMORY: no global loads (values made up in registers) and no stores (behind a condition that never holds): what is
// left is the kernel's instruction stream -- the time below which no amount of temporal blocking can push a pass.
// PX_NO_COMPUTE: the loads and stores of the product, the arithmetic and the edge exchange replaced by a few additions that
// consume every loaded row: the memory side alone, with the product's access pattern and occupancy.
enum : int { PX_NO_MAP = 1, PX_NO_FLAGS = 2, PX_PREV_NT = 8, PX_STORE_CACHED = 16, PX_NO_MEMORY = 32, PX_NO_COMPUTE = 64 };

// NWC > 0: the row length is a compile-time constant, NWC waves = NWC * 64 * 16 B per row (address
// arithmetic folds; measured -6 % at 1024 doubles per row before div3, nothing since); 0: any row length.
// WIDE: rows longer than one workgroup can hold (kPairMaxWaves waves).  Several workgroups share a row, each a window
// of up to kPairMaxWaves waves that OVERLAP by two: a window's outermost wave on an interior side is a halo wave --
// it loads, exchanges edges and computes like any other, but stores nothing.  That is all it takes: with nothing
// to its outside, the halo wave's t+1 is wrong in its outermost column only (the missing neighbour counts as 0),
// which reaches no further than its own t+2; its innermost column's t+1 -- what the first storing wave needs --
// is right.  25 % (8 waves run for 6 stored) more arithmetic and L2 traffic for such rows, the same HBM bytes.
// (The body is a device function so that tools/pair_tune.hip can price it at other strip heights and occupancies; the
// product is pair_march_kernel below, RYT = kPairRows.)
template <typename Real, int X, int NWC, bool WIDE, int RYT>
__device__ __forceinline__ void pair_march_body(const PairArgs<Real>& a) {
    // (The arguments are never written to: a modified copy of the struct is no longer promoted to registers, its
    // pointers are re-read from scratch and lose their address space -- flat loads, a wait after each, three
    // times the run time.  That, not register pressure, was what made the NWC variant collapse in mid-round.)
    const int pitch = NWC > 0 ? NWC * 64 * Vec16<Real>::N : a.pitch;
    const int cls_pitch = NWC > 0 ? pitch / 4 : a.cls_pitch;
    using V = typename Vec16<Real>::type;
    constexpr int VX = Vec16<Real>::N;
    constexpr int RY = RYT;
    constexpr int K = 2 * RY + 2;  // edge rows per plane: RY+2 of `current`, RY of t+1
    __shared__ Real sl[2][K][kPairMaxWaves], sr[2][K][kPairMaxWaves];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD k (= blockIdx % 8, observed dispatch: used for locality only) takes a contiguous run of
    // strips, so that the ring rows two neighbouring strips both need meet in that XCD's L2
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    int row_waves = NWC > 0 ? NWC : a.nw;  // waves side by side in this workgroup ...
    int wave_first = 0;                      // ... the first of them counted along the row ...
    int store_lo = 0, store_hi = 1 << 30;    // ... and which waves of the row this workgroup stores
    if (WIDE && a.windows) {  // windows outermost: the workgroups of one window are the grid of a narrow mesh
        const int per_window = (int)(gridDim.x >> 3) / a.windows;
        const int win = j / per_window;
        j -= win * per_window;
        row_waves = (int)((a.win_count >> (8 * win)) & 0xFFu);
        wave_first = (int)((a.win_first >> (8 * win)) & 0xFFu);
        store_lo = (int)((a.win_store_lo >> (8 * win)) & 0xFFu);
        store_hi = (int)((a.win_store_hi >> (8 * win)) & 0xFFu);
    }
    int strip, chunk;
    if (a.unit_list) {
        const uint32_t first = a.list_start[xcd], count = a.list_start[xcd + 1] - first;
        if ((uint32_t)j >= count) return;  // whole workgroup
        const uint32_t u = a.unit_list[first + (uint32_t)j];
        strip = (int)(u & 0xFFFFu);
        chunk = (int)((u >> 16) & 0x1FFu);
        if (WIDE && !a.windows) {
            // rooms narrower than their rows: only the waves of the row between the first and the last one that holds
            // anything but `none` nodes for this unit (its rows +- a strip, its planes +- 2) -- what lies beyond is
            // zeros in every field, which is what a missing neighbour counts as
            wave_first = (int)((u >> 25) & 7u);
            row_waves = (int)(u >> 28) + 1;
        }
    } else {
        strip = xcd * a.strips_per_xcd + j % a.strips_per_xcd;
        chunk = j / a.strips_per_xcd;
    }
    if (WIDE && wave >= row_waves) return;  // (a finished wave does not hold up the barriers)
    const int wave_abs = wave_first + wave;
    const bool stores = !WIDE || (wave_abs >= store_lo && wave_abs < store_hi);
    if (strip >= a.strips || chunk >= a.chunks) return;  // whole workgroup
    const int y0 = strip * RY;
    const int zb = a.z_begin + chunk * a.zc, ze = min(zb + a.zc, a.z_end);
    if (zb >= ze) return;

    PairTile<Real> t;
    t.ny = a.ny;
    t.nz = a.nz;
    t.pitch = pitch;
    t.plane = (int64_t)pitch * a.ny;
    t.col = (wave_abs * 64 + lane) * VX;

    const PairEdges<Real, K> edges{sl, sr, lane, wave, row_waves};
    auto made_up = [&](int y, int z) -> V {  // (PX_NO_MEMORY)
        V v;
#pragma unroll
        for (int k = 0; k < VX; ++k) v[k] = Real(y) * Real(0.001) + Real(z + k + lane);
        return v;
    };
    auto load_b = [&](V(&dst)[RY + 4], int z) {
#pragma unroll
        for (int q = 0; q < RY + 4; ++q) dst[q] = (X & PX_NO_MEMORY) ? made_up(y0 - 2 + q, z) : t.load(a.cur, y0 - 2 + q, z);
    };
    // t+1 on plane z, rows y0-1 .. y0+RY, from current(z-1), current(z), current(z+1) and previous(z);
    // el / er: x edges of the current(z) rows y0-1 .. y0+RY.  Rows / planes off the grid are 0.
    // previous(z), rows y0-1 .. y0+RY: every row is also read by the strip above or below (as its ring
    // row), so no non-temporal hint -- the second reader should find it in L2
    auto load_p = [&](V(&dst)[RY + 2], int z) {
#pragma unroll
        for (int q = 0; q < RY + 2; ++q)
            dst[q] = (X & PX_NO_MEMORY) ? made_up(y0 - 1 + q, -z)
                                        : ((X & PX_PREV_NT) ? t.load_nt(a.prev, y0 - 1 + q, z) : t.load(a.prev, y0 - 1 + q, z));
    };
    // (x edges of the current(z) rows: edge rows 0 .. RY+1 of LDS set `set`)
    auto level1 = [&](V(&dst)[RY + 2], const V(&bm)[RY + 4], const V(&b0)[RY + 4], const V(&bp)[RY + 4],
                      const V(&pv)[RY + 2], int z, int set) {
#pragma unroll
        for (int q = 0; q < RY + 2; ++q) {
            const int y = y0 - 1 + q;
            const V v = pair_step_row<Real>(b0[q + 1], b0[q], b0[q + 2], bm[q + 1], bp[q + 1], pv[q], edges.left(set, q),
                                            edges.right(set, q));
            dst[q] = (y >= 0 && y < a.ny && z >= 0 && z < a.nz) ? v : (V)(Real(0));
        }
    };
    // 2-bit codes of this lane's VX nodes in the RY rows of the strip on plane z: one dword load
    auto codes_of = [&](int z) -> uint32_t {
        return reinterpret_cast<const uint32_t*>(a.pair_map)[cls_word_index(t.col, y0, z, a.ny, cls_pitch)];
    };
    auto row_codes = [&](uint32_t word, int r) -> uint32_t {
        const uint32_t byte = (word >> (r * 8)) & 0xFFu;
        return (VX == 4) ? byte : ((byte >> ((lane & 1) * 4)) & 0xFu);
    };

    // Register rings: three planes of `current` and three of t+1.  The loop below is unrolled by three
    // with the roles rotating through the rings, so that no plane is ever copied between registers.
    V bA[RY + 4], bB[RY + 4], bC[RY + 4];
    V tA[RY + 2], tB[RY + 2], tC[RY + 2];
    V pv[RY + 2];

    // ---- prologue: t+1 on planes zb-1 (-> tA) and zb (-> tB); current(zb) -> bB, current(zb+1) -> bC
    load_b(bC, zb - 2);
    load_b(bA, zb - 1);
    load_b(bB, zb);
    load_p(pv, zb - 1);
#pragma unroll
    for (int q = 0; q < RY + 2; ++q) edges.publish(0, q, bA[q + 1]);
    lds_barrier();
    level1(tA, bC, bA, bB, pv, zb - 1, 0);
    load_b(bC, zb + 1);
    load_p(pv, zb);
#pragma unroll
    for (int q = 0; q < RY + 2; ++q) edges.publish(1, q, bB[q + 1]);
    lds_barrier();
    level1(tB, bA, bB, bC, pv, zb, 1);

    bool nonfinite = false;
    // One plane: b_mid / b_hi = current(z) / current(z+1), b_nn receives current(z+2);
    // t_prev / t_cur = t+1(z-1) / t+1(z), t_next receives t+1(z+1).
    auto plane = [&](int z, int set, const V(&b_mid)[RY + 4], const V(&b_hi)[RY + 4], V(&b_nn)[RY + 4],
                     const V(&t_prev)[RY + 2], const V(&t_cur)[RY + 2], V(&t_next)[RY + 2]) {
        // everything planes z+1 / z+2 need from memory is requested first and stays in flight across
        // the edge exchange
        load_b(b_nn, z + 2);
        load_p(pv, z + 1);
        const uint32_t code_word = (X & PX_NO_MAP) ? 0x55555555u : codes_of(z);
        Real* const o1_base = (z >= a.out1_z0 && z < a.out1_z1) ? a.out1 : a.out2;
        if (X & PX_NO_COMPUTE) {  // (tools/pair_tune only)
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                if (y0 + r < a.ny && stores) {
                    const V o1 = b_nn[r] + b_nn[r + RY] + pv[r] + (r < 2 ? pv[r + RY] : b_mid[r]);
                    const V o2 = b_nn[r + RY] + b_hi[r + 1] + pv[(r + 2) % (RY + 2)];
                    t.store(a.out1, y0 + r, z, o1);
                    t.store(a.out2, y0 + r, z, o2);
                }
            }
            return;
        }
        // one exchange serves both levels: x edges of current(z+1) (for t+1 on z+1) and of t+1(z) (for t+2 on z)
#pragma unroll
        for (int q = 0; q < RY + 2; ++q) edges.publish(set, q, b_hi[q + 1]);
#pragma unroll
        for (int r = 0; r < RY; ++r) edges.publish(set, RY + 2 + r, t_cur[r + 1]);
        lds_barrier();
        level1(t_next, b_mid, b_hi, b_nn, pv, z + 1, set);
        // What is stored: 0 at "none" nodes (code 0), the computed value everywhere else -- at nodes this
        // kernel cannot finish (codes 2, 3) that value is a placeholder the boundary kernel / the fix-up
        // list overwrites.  keep bit of node k of row r: bit 2k of byte r of `keep`.
        const uint32_t keep = code_word | (code_word >> 1);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            if (y0 + r < a.ny && stores) {
                const V v2 = pair_step_row<Real>(t_cur[r + 1], t_cur[r], t_cur[r + 2], t_prev[r + 1], t_next[r + 1],
                                                 b_mid[r + 2], edges.left(set, RY + 2 + r), edges.right(set, RY + 2 + r));
                V o1 = t_cur[r + 1], o2 = v2;
                const uint32_t bits = row_codes(keep, r);
#pragma unroll
                for (int k = 0; k < VX; ++k) {
                    const bool live = (bits >> (2 * k)) & 1u;
                    o1[k] = live ? o1[k] : Real(0);
                    o2[k] = live ? o2[k] : Real(0);
                    // one class test per value; exact flags are worked out after the march, and only
                    // if anything non-finite was seen at all (never, in a healthy run)
                    if (!(X & PX_NO_FLAGS)) nonfinite |= (bool)((int)!is_finite(o1[k]) | (int)!is_finite(o2[k]));  // no short circuit: no branch
                }
                if (X & PX_NO_MEMORY) {
                    if (nonfinite) {  // (never: the made-up values stay finite)
                        t.store(a.out1, y0 + r, z, o1);
                        t.store(a.out2, y0 + r, z, o2);
                    }
                } else if (X & PX_STORE_CACHED) {
                    t.store_cached(a.out1, y0 + r, z, o1);
                    t.store_cached(a.out2, y0 + r, z, o2);
                } else {
                    // (a plane whose t+1 is not to be stored -- PairArgs::out1_z0 -- sends it to the t+2 field instead, where the
                    // store after it, same lane, same address, replaces it: a scalar select of the base pointer, where a branch
                    // around the store cost the double march 12 B of scratch)
                    t.store(o1_base, y0 + r, z, o1);
                    t.store(a.out2, y0 + r, z, o2);
                }
            }
        }
    };
    // (LDS sets: the prologue used 0 then 1; six planes per trip keep set = plane parity a constant)
    for (int z = zb; z < ze; z += 6) {
        plane(z, 0, bB, bC, bA, tA, tB, tC);
        if (z + 1 < ze) plane(z + 1, 1, bC, bA, bB, tB, tC, tA);
        if (z + 2 < ze) plane(z + 2, 0, bA, bB, bC, tC, tA, tB);
        if (z + 3 < ze) plane(z + 3, 1, bB, bC, bA, tA, tB, tC);
        if (z + 4 < ze) plane(z + 4, 0, bC, bA, bB, tB, tC, tA);
        if (z + 5 < ze) plane(z + 5, 1, bA, bB, bC, tC, tA, tB);
    }
    if (__any(nonfinite)) {
        // Slow path: something this workgroup stored is inf or nan.  Re-read what it stored and raise
        // the step's flag bits for exactly the nodes the reference would test: t+1 where the node takes
        // the 7-point update (codes 1, 3), t+2 where this kernel's value is final (code 1).  Non-finite
        // placeholders at other nodes are not errors (their owners test the real values).
        int bad1 = 0, bad2 = 0;
        for (int z = zb; z < ze; ++z) {
            const uint32_t code_word = (X & PX_NO_MAP) ? 0x55555555u : codes_of(z);
            for (int r = 0; r < RY; ++r) {
                if (y0 + r >= a.ny) break;
                const V o1 = t.load(a.out1, y0 + r, z), o2 = t.load(a.out2, y0 + r, z);
                const uint32_t codes = row_codes(code_word, r);
                for (int k = 0; k < VX; ++k) {
                    const uint32_t c = (codes >> (2 * k)) & 3u;
                    if (c & 1u) bad1 |= bad_bits(o1[k]);
                    if (c == 1u) bad2 |= bad_bits(o2[k]);
                }
            }
        }
        if (bad1) atomicOr(a.flag1, bad1);
        if (bad2) atomicOr(a.flag2, bad2);
    }
}

template <typename Real, int X = 0, int NWC = 0, bool WIDE = false>
__global__ void __launch_bounds__(64 * kPairMaxWaves) pair_march_kernel(const PairArgs<Real> a) {
    pair_march_body<Real, X, NWC, WIDE, kPairRows>(a);
}

// ---- t+2 of the nodes the march could not finish (code 3): plain 7-point update from the complete t+1
// field (boundary nodes and the source sample included by now)
template <typename Real>
struct PairFixupArgs {
    const uint32_t* nodes;  // stored indices
    uint32_t n;
    const Real* t1;    // t+1, complete
    const Real* cur;   // t
    Real* out2;        // t+2
    int* flag2;
    int nx, ny, nz, pitch;
};

template <typename Real>
__device__ __forceinline__ void pair_fixup_node(const PairFixupArgs<Real>& a, uint32_t i) {
    if (i >= a.n) return;
    const uint32_t idx = a.nodes[i];
    const int x = (int)(idx % (uint32_t)a.pitch);
    const uint32_t q = idx / (uint32_t)a.pitch;
    const int y = (int)(q % (uint32_t)a.ny);
    const int z = (int)(q / (uint32_t)a.ny);
    const int64_t plane = (int64_t)a.pitch * a.ny;
    Real s = 0;
    s += (x > 0) ? a.t1[idx - 1] : Real(0);
    s += (x + 1 < a.nx) ? a.t1[idx + 1] : Real(0);
    s += (y > 0) ? a.t1[idx - a.pitch] : Real(0);
    s += (y + 1 < a.ny) ? a.t1[idx + a.pitch] : Real(0);
    s += (z > 0) ? a.t1[idx - plane] : Real(0);
    s += (z + 1 < a.nz) ? a.t1[idx + plane] : Real(0);
    s = div3(s);
    s -= a.cur[idx];
    const int bad = bad_bits(s);
    if (bad) atomicOr(a.flag2, bad);
    a.out2[idx] = s;
}

template <typename Real>
__global__ void __launch_bounds__(256) pair_fixup_kernel(const PairFixupArgs<Real> a) {
    pair_fixup_node<Real>(a, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- the pair map and the fix-up list, once per (mesh, source node) --------------------------------
struct PairMapArgs {
    const uint8_t* cls;
    uint8_t* pair_map;
    uint32_t* list;        // null: count only.  Nodes of the marched planes first ...
    uint32_t* list_face;   // ... nodes of a slab's face planes here (the march does not produce those planes)
    uint32_t* counter;     // [3]: code-3 nodes of the marched planes / of the face planes (count pass), write cursors (fill
                           // pass); [2]: listed nodes of the marched planes with a boundary node for a neighbour
    uint64_t source_node;  // stored index of the source node, ~0 = none
    int nx, ny, nz, pitch, cls_pitch;
    int z_begin, z_end;          // planes this engine owns
    int march_begin, march_end;  // planes the march produces: all owned planes but a slab's face planes
    // non-zero: an inside node of the marched planes that a (1-D) boundary node of the marched planes faces
    // is finished by that node's entry in the second boundary launch (boundary_kernel<.., FIX = true>) and stays
    // off the list.  Only for meshes that passed pair_inner_check_kernel.
    int cover;
};

// "A boundary node has an inside node next to it exactly when it is one-dimensional, and then its one
// direction bit names that node" -- how set_node_boundary_type types nodes (mesh_setup_program.cpp:110-172:
// 1-D = one axial inside neighbour, 2-D / 3-D = none, only a diagonal one; two or more = re-entrant), true
// for every mesh the set-up chain produces, not promised by a caller's own node array.  1-D entries
// finish the node they face (above) only when it holds for all entries.
struct PairInnerCheckArgs {
    const uint32_t* bnode;
    const uint8_t* btype;
    const uint8_t* cls;
    uint32_t n_entries;
    int nx, ny, nz, pitch, cls_pitch;
    int* violated;
};

__global__ void __launch_bounds__(256) pair_inner_check_kernel(const PairInnerCheckArgs a) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_entries) return;
    const uint32_t idx = a.bnode[e];
    if (idx == INVALID_NODE) return;
    const uint32_t dirs = a.btype[e] & 0x3Fu;
    const int pos[3] = {(int)(idx % (uint32_t)a.pitch), (int)((idx / (uint32_t)a.pitch) % (uint32_t)a.ny),
                        (int)(idx / ((uint32_t)a.pitch * (uint32_t)a.ny))};
    const int lim[3] = {a.nx, a.ny, a.nz};
    const bool one_d = __popc(dirs) == 1;
    bool ok = true;
    for (int port = 0; port < 6; ++port) {
        int p[3] = {pos[0], pos[1], pos[2]};
        p[port >> 1] += (port & 1) ? 1 : -1;
        const bool in_grid = p[port >> 1] >= 0 && p[port >> 1] < lim[port >> 1];
        const uint32_t c = in_grid ? (a.cls[cls_byte_index(p[0], p[1], p[2], a.ny, a.cls_pitch)] >> ((p[0] & 3) * 2)) & 3u : CLS_NONE;
        ok = ok && ((c == CLS_INSIDE) == (one_d && ((dirs >> port) & 1u) != 0));
    }
    if (!ok) *a.violated = 1;
}

// one thread per class byte (4 nodes of one row)
__global__ void __launch_bounds__(256) pair_map_kernel(const PairMapArgs a) {
    const int64_t n_bytes = (int64_t)a.cls_pitch * a.ny * a.nz;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bytes) return;
    const int xb = (int)(i % a.cls_pitch);
    const int64_t row = i / a.cls_pitch;
    const int y = (int)(row % a.ny), z = (int)(row / a.ny);
    auto cls_at = [&](int x, int yy, int zz) -> uint32_t {
        if (x < 0 || x >= a.pitch || yy < 0 || yy >= a.ny || zz < 0 || zz >= a.nz) return 1u;  // off the grid: see below
        return (a.cls[cls_byte_index(x, yy, zz, a.ny, a.cls_pitch)] >> ((x & 3) * 2)) & 3u;
    };
    auto is_source = [&](int x, int yy, int zz) -> bool {
        if (x < 0 || x >= a.pitch || yy < 0 || yy >= a.ny || zz < 0 || zz >= a.nz) return false;
        return ((uint64_t)zz * a.ny + yy) * (uint64_t)a.pitch + x == a.source_node;
    };
    const uint32_t own = a.cls[cls_byte_index(xb * 4, y, z, a.ny, a.cls_pitch)];
    uint32_t out = 0;
    for (int k = 0; k < 4; ++k) {
        const int x = xb * 4 + k;
        const uint32_t c = (own >> (2 * k)) & 3u;
        uint32_t code = c == CLS_BOUNDARY ? 2u : 0u;
        if (c & 1u) {
            bool plain = true;
            const int nb[6][3] = {{x - 1, y, z}, {x + 1, y, z}, {x, y - 1, z}, {x, y + 1, z}, {x, y, z - 1}, {x, y, z + 1}};
            for (int p = 0; p < 6; ++p) {
                plain = plain && (cls_at(nb[p][0], nb[p][1], nb[p][2]) & 1u) && !is_source(nb[p][0], nb[p][1], nb[p][2]);
            }
            // (a neighbour off the stored grid contributes the same 0 to the march as to the reference's
            // update; a pad column, x in [nx, pitch), is class "none" and sends its neighbour to the list)
            // a slab's face planes wait for the neighbour's t+1 face: the fix-up list takes all of their nodes
            const bool marched = z >= a.march_begin && z < a.march_end;
            plain = plain && marched;
            code = plain ? 1u : 3u;
            bool faced = false;  // by a boundary node whose entry finishes this node (PairMapArgs::cover)
            if (a.cover && !plain && marched && c == CLS_INSIDE) {
                for (int p = 0; p < 6; ++p) {
                    const bool in_grid = nb[p][0] >= 0 && nb[p][0] < a.nx && nb[p][1] >= 0 && nb[p][1] < a.ny &&
                                         nb[p][2] >= a.march_begin && nb[p][2] < a.march_end;
                    faced = faced || (in_grid && cls_at(nb[p][0], nb[p][1], nb[p][2]) == CLS_BOUNDARY);
                }
            }
            if (!plain && !faced && z >= a.z_begin && z < a.z_end) {
                if (marched) {
                    bool wall = false;
                    for (int p = 0; p < 6; ++p) wall = wall || cls_at(nb[p][0], nb[p][1], nb[p][2]) == CLS_BOUNDARY;
                    if (wall) atomicAdd(a.counter + 2, 1u);
                }
                const uint32_t at = atomicAdd(a.counter + (marched ? 0 : 1), 1u);
                uint32_t* dst = marched ? a.list : a.list_face;
                if (dst) dst[at] = (uint32_t)(((int64_t)z * a.ny + y) * a.pitch + x);
            }
        }
        out |= code << (2 * k);
    }
    a.pair_map[cls_byte_index(xb * 4, y, z, a.ny, a.cls_pitch)] = (uint8_t)out;
}

// ---- which waves of which strips hold anything but `none` nodes at which plane (rooms narrower than their rows) ---
struct WaveActivityArgs {
    const uint8_t* cls;
    uint8_t* raw;  // [nz][strips]: bit w = wave w's column block (128 doubles / 256 floats) of the strip's rows at plane z
    int ny, nz, pitch, cls_pitch, strips, nw, wave_cols;
    uint16_t* raw16;  // the same with room for 16 waves (the three-step march's narrower waves); null: `raw` is written
};

__global__ void __launch_bounds__(256) pair_wave_activity_kernel(const WaveActivityArgs a) {
    const int64_t n = (int64_t)a.nz * a.strips;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int s = (int)(t % a.strips), z = (int)(t / a.strips);
    uint32_t bits = 0;
    for (int y = s * kPairRows; y < min((s + 1) * kPairRows, a.ny); ++y)
        for (int w = 0; w < a.nw; ++w) {
            uint32_t any = 0;
            for (int x = w * a.wave_cols; x < min((w + 1) * a.wave_cols, a.pitch); x += 4)
                any |= a.cls[cls_byte_index(x, y, z, a.ny, a.cls_pitch)];  // any class but CLS_NONE
            if (any) bits |= 1u << w;
        }
    if (a.raw16)
        a.raw16[t] = (uint16_t)bits;
    else
        a.raw[t] = (uint8_t)bits;
}

}  // namespace wv
