// triple_kernels.hip.h -- THREE time steps of the pressure update in one pass over the fields.
//
// Same arithmetic as stream_kernels.hip.h / pair_kernels.hip.h (`normal_waveguide_update`,
// src/waveguide/src/program.cpp:393-412, applied three times), bit for bit; what changes is the traffic again.
// The two-step pass (pair_kernels.hip.h) moves 32 B per node for two updates and runs at what the memory system
// delivers; a pass that produces t+2 AND t+3 from (t-1, t) moves the same 32 B for THREE updates (10.7 B per
// node-update) -- if its state fits on the chip and its instruction stream hides behind the memory time.  The round-3
// prototype (tools/triple_bench.hip: strips of TWO rows, twice the arithmetic) lost.  This is the form with strips of
// FOUR rows (18 row updates for 12 useful ones, 1.5 x), made to fit by these things:
//
//   * of the three planes each time level needs, only TWO live in registers: the plane the level is being computed
//     at ("mid") and the plane ahead ("new", arriving row by row).  The plane behind ("lo") is read exactly twice per
//     row -- as the z-1 neighbour of this level and as the "own old value" of the next level -- so it lives in LDS, in
//     18 wave-private slots (8 + 6 + 4 rows), rewritten in place by the rows of "mid" as they retire.  No barrier
//     guards them: a wave's LDS operations execute in order and nobody else touches its slots.
//   * a lane holds EIGHT bytes of a row (one double / two floats), a wave 512 bytes: half the state per wave of the
//     two-step march's 16-byte lanes, while what does not scale with the lane's width (addresses, masks, the sums
//     in flight) stays -- the 16-byte form needed ~300 vector registers of the 256 a wave can have at two waves per
//     SIMD; this one takes 166, three waves per SIMD: a workgroup is up to 12 waves = 768 doubles of a row, longer
//     rows are shared by windows with halo waves, as in the two-step march (1024 doubles: two windows of 8 + 1).
//   * memory goes through buffer descriptors, one per (field, plane): a row-vector's address is the descriptor's
//     base (scalar) + a scalar row offset + ONE per-lane offset register for the whole kernel; rows and planes off
//     the grid are left to the hardware's range check (zeros on loads, stores dropped), lane-masked stores aim out
//     of range instead of branching.  No 64-bit per-lane address arithmetic, no branch inside a trip (an exec-masked
//     store alone cost the kernel 85 spilled registers: it cuts the trip into scheduling regions).
//   * loads are issued about four rows ahead of their use, in the order of use.
//
// A workgroup owns a strip of 4 rows x its window of the row and marches through a chunk of planes.  Trip f (the front
// plane f arrives): t+1 on plane f-1 (8 rows), t+2 on plane f-2 (6 rows), t+3 on plane f-3 (4 rows); stored per trip: 4
// rows of t+2 and of t+3 (plane f-3), and t+1 where the map asks for it (below).  Rows are interleaved -- t+1 row i, then
// t+2 row i-1, then t+3 row i-2 -- so that register rows retire as fast as new ones appear.  x neighbours: DPP inside a
// wave, edge columns between waves through LDS, published for the NEXT trip at the end of this one (all three levels'
// centre planes are known by then): one LDS-only barrier per trip.
//
// What the march may finish is decided by a 2-bit map with the class map's layout (triple_map_kernel, engine_triple.hip.h):
//   0  outside the room ("none"): t+2 and t+3 store 0 (every field holds 0 there at all times)
//   1  deep: takes the 7-point update and nothing within two nodes of it is anything else (boundary node, outside
//      node, source node): t+2 and t+3 are final here, t+1 is NOT stored (nobody reads it)
//   3  shell: takes the 7-point update, within two nodes of something else, or a receiver node / the source node:
//      t+1 is stored (final), t+2 / t+3 are placeholders where the node is within one / two nodes (fix-up lists)
//   2  boundary node: every level belongs to the boundary kernel (placeholders)
#pragma once
#include <type_traits>

#include "pair_kernels.hip.h"

namespace wv {

constexpr int kTripleRows = 4;
constexpr int kTripleMaxWaves = 12;  // waves side by side in one workgroup (three per SIMD at <= 170 registers): 12 * 64 * 8 B of a row
constexpr int kTripleLoSlots = 18;   // rows of the three "lo" planes a wave keeps in LDS: 8 of `current`, 6 of t+1, 4 of t+2
constexpr int kTripleEdgeRows = 18;  // centre rows per trip that need x edges: 8 (t+1) + 6 (t+2) + 4 (t+3)
constexpr int kTripleLaneBytes = 8;
constexpr int kTriplePvSlots = 8;    // DMA form: the eight rows of `previous` a trip needs, landing in LDS a trip ahead
constexpr int kTripleMaxWavesDma = 11;  // ... whose slots leave room for 11 waves in a CU's 160 KB

// a lane's LB bytes of a row (8: one double or two floats; 16: two doubles or four floats)
template <typename Real, int LB>
struct VecL {
    static constexpr int N = LB / (int)sizeof(Real);
    Real v[N];
    __device__ __forceinline__ Real& operator[](int k) { return v[k]; }
    __device__ __forceinline__ const Real& operator[](int k) const { return v[k]; }
};
// waves side by side in one workgroup: 8-byte lanes take <= 170 registers, three waves per SIMD; 16-byte lanes all 256, two per SIMD
constexpr int triple_max_waves(int LB) { return LB == 8 ? 12 : 8; }

template <typename Real>
struct TripleArgs {
    const Real* prev;  // t-1
    const Real* cur;   // t
    Real* out1;        // t+1 (shell nodes only)
    Real* out2;        // t+2
    Real* out3;        // t+3
    const uint8_t* map;  // triple map (above)
    int* suspect;        // set when anything the march kept is inf or nan: triple_flags_kernel then works out the exact error bits
    int ny, nz, pitch, cls_pitch;
    int z_begin, z_end;  // planes to produce
    // z-slabs: t+2 is wanted one plane beyond either end of that range too (z2_lo / z2_hi: 1 or 0) -- the plane next to a face plane, whose
    // t+3 is a plain step's business (the face's t+2 waits for the neighbour's) but whose t+2 only this march can supply: its neighbours'
    // t+1 is stored at shell nodes alone.  The first chunk stores it from its last warm-up trip, the last chunk runs one trip more.
    int z2_lo, z2_hi;
    int nw;              // waves per workgroup
    int zc, chunks;      // planes per workgroup, workgroups along z
    int strips, strips_per_xcd;
    // rows longer than one workgroup holds: `windows` workgroups side by side.  Window k runs waves [win_first_k, win_first_k +
    // win_count_k) of the row and stores those in [win_store_lo_k, win_store_hi_k); a wave it runs beyond either end of that range is
    // a halo wave: it loads, exchanges edges and computes like any other and stores nothing -- with nothing to its outside its t+1 /
    // t+2 / t+3 are wrong in its outermost one / two / three columns only, 61 columns away from what the first storing wave needs.
    // (byte k of each word belongs to window k)
    int windows;
    uint64_t win_first, win_count, win_store_lo, win_store_hi;
    // optional work list (rooms that leave much of the mesh outside; no windows then): workgroup j of XCD k takes unit
    // unit_list[list_start[k] + j] = strip | chunk << 14 | first wave << 23 | waves - 1 << 27 -- the waves of the row between the first
    // and the last one that holds anything but `none` nodes in all the unit reads, produces or hands on (its rows +- a strip, its planes
    // +- 3); what lies beyond is zeros in every field, which is what a missing neighbour counts as.  Units without a node to update are
    // not listed: their outputs keep the zeros they hold.
    const uint32_t* unit_list;
    uint32_t list_start[9];
};

constexpr int kTripleMaxWindows = 8;

// a work-list entry's fields
__host__ __device__ inline uint32_t triple_unit_entry(uint32_t strip, uint32_t chunk, uint32_t wave_first, uint32_t waves) {
    return strip | (chunk << 14) | (wave_first << 23) | ((waves - 1u) << 27);
}

// How a row of `row_waves` waves is shared out: windows of at most kTripleMaxWaves waves, one halo wave on every interior side.
// win[0..3][k] = first wave run, waves run, first wave stored, end of the stored waves.  `full_first`: as many full workgroups (12 waves:
// three on every SIMD) as the row gives and one short one for the rest -- two short ones share a CU -- instead of equal shares (16
// waves: 12 + 6 instead of 9 + 9, whose 9 waves are 3 + 2 + 2 + 2 on a CU's SIMDs and as slow as 12).
// Returns the number of windows (0: the row is one workgroup), -1 if the row is too long; *widest = waves per workgroup.
inline int triple_windows(int row_waves, uint8_t win[4][kTripleMaxWindows], int* widest, bool full_first = true, int max_waves = kTripleMaxWaves) {
    *widest = row_waves;
    if (row_waves <= max_waves) return 0;
    int n = 0, at = 0;
    *widest = 0;
    if (full_first) {
        while (at < row_waves && n < kTripleMaxWindows) {
            const int lo_halo = at > 0 ? 1 : 0;
            int end = at + max_waves - lo_halo;  // storing [at, end) with no halo above ...
            if (end < row_waves) end -= 1;             // ... or one wave less and a halo wave
            end = end < row_waves ? end : row_waves;
            const int first = at - lo_halo, last = end + (end < row_waves ? 1 : 0);
            win[0][n] = (uint8_t)first;
            win[1][n] = (uint8_t)(last - first);
            win[2][n] = (uint8_t)at;
            win[3][n] = (uint8_t)end;
            if (last - first > *widest) *widest = last - first;
            ++n;
            at = end;
        }
        return at < row_waves ? -1 : n;
    }
    n = 2;  // the widest window stores ceil(row_waves / n) waves and has a halo wave on one side (n = 2) or two
    while (n <= kTripleMaxWindows && (row_waves + n - 1) / n + (n > 2 ? 2 : 1) > max_waves) ++n;
    if (n > kTripleMaxWindows) return -1;
    for (int k = 0; k < n; ++k) {
        const int lo = row_waves * k / n, hi = row_waves * (k + 1) / n;
        const int first = lo - (k > 0 ? 1 : 0), last = hi + (k + 1 < n ? 1 : 0);
        win[0][k] = (uint8_t)first;
        win[1][k] = (uint8_t)(last - first);
        win[2][k] = (uint8_t)lo;
        win[3][k] = (uint8_t)hi;
        if (last - first > *widest) *widest = last - first;
    }
    return n;
}

inline size_t triple_lds_bytes(int nw, bool dma = false, int LB = kTripleLaneBytes) {
    return (size_t)nw * (kTripleLoSlots + (dma ? kTriplePvSlots : 0)) * 64 * LB + (size_t)2 * kTripleEdgeRows * (triple_max_waves(LB) + 2) * 2 * LB;
}

// The 7-point update of one lane's columns of one row, in the reference's order (pair_step_row, for 8-byte lanes).
template <typename Real, int LB>
__device__ __forceinline__ VecL<Real, LB> triple_step_row(const VecL<Real, LB>& c0, const VecL<Real, LB>& ym, const VecL<Real, LB>& yp,
                                                          const VecL<Real, LB>& zm, const VecL<Real, LB>& zp, const VecL<Real, LB>& pv, Real edge_l,
                                                          Real edge_r) {
    constexpr int VX = VecL<Real, LB>::N;
    VecL<Real, LB> out;
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

// the top 32 bits of |v|: an inf or nan is what compares >= 0x7FF00000 (double) / 0x7F800000 (float)
__device__ __forceinline__ uint32_t abs_bits_hi(double v) { return (uint32_t)__double2hiint(v) & 0x7FFFFFFFu; }
__device__ __forceinline__ uint32_t abs_bits_hi(float v) { return __float_as_uint(v) & 0x7FFFFFFFu; }

// Experiment switches (tools/triple4_bench.hip only; the engine runs X = 0)
enum : int { TX_NO_MEMORY = 1, TX_STORE_CACHED = 2, TX_WRAP_Z = 4 /* every plane is plane z & 1: the traffic stays in the caches */ };

// EDGE: the strip's trips touch rows off the grid (the first strip, the last one or two): every row offset goes through the range
// check and rows of t+1 / t+2 off the grid are forced to zero.  Interior strips know their rows are there.
// Vector-memory instructions a trip issues in row step m (triple_march_body: the same conditions, in the same order) ...
constexpr int triple_vm_ops(int LA, bool dma, int m) {
    constexpr int R0 = kTripleRows + 6, R1 = kTripleRows + 4;
    int n = 0;
    n += (m + LA + 1 < R0) ? 1 : 0;               // b_new[m + LA + 1]
    n += (m + LA + 1 == R0) ? 1 : 0;              // b_new[0]
    n += (!dma && m + LA < R1) ? 1 : 0;           // pv[m + LA]
    n += (m >= R1 + 1 - LA) ? (dma ? 1 : 2) : 0;  // the next trip's first rows
    n += (m >= 2 && m - 2 < kTripleRows) ? 3 : 0;  // t+1 / t+2 / t+3 stores of a finished row
    n += (dma && m >= 2 && (m & 1) == 0) ? 1 : 0;  // the DMA of pair m/2 - 1
    return n;
}
// ... and so how many are issued, at the very least, between the DMA of pair p (first in row step 2p+2 of the trip before) and row step
// 2p, where its rows are first read: the rest of that trip, the map load at the top of this one, this trip's row steps before 2p.  Two
// less than counted, for safety: a smaller number only waits for more.
constexpr int triple_dma_wait(int LA, int p) {
    int n = triple_vm_ops(LA, true, 2 * p + 2) - 1;
    for (int m = 2 * p + 3; m <= kTripleRows + 4; ++m) n += triple_vm_ops(LA, true, m);
    n += 1;
    for (int m = 0; m < 2 * p; ++m) n += triple_vm_ops(LA, true, m);
    return n > 2 ? n - 2 : 0;
}

// vmcnt <= n, nothing else waited for (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// lds_barrier() for a kernel with LDS-DMA in flight: its "local" fence would make the compiler wait for every pending `buffer_load ...
// lds` (they are LDS writes to it) -- a memory latency at every barrier.  Here: the wave's own LDS instructions retired, then the barrier.
__device__ __forceinline__ void lds_barrier_keep_vm() {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
    __builtin_amdgcn_s_barrier();
}

// PVDMA: the rows of `previous` (each read once, as a node's own old value) do not pass through registers: `buffer_load ... lds` drops
// them into eight more wave-private LDS slots a whole trip ahead, two rows per instruction.  The compiler counts such a load in vmcnt but
// does not order LDS reads behind it; the waits are placed by hand (triple_dma_wait: vector-memory instructions are issued and retired
// in order, and the row steps are fenced scheduling regions, so how many are issued between a DMA and the row step that reads its
// rows is a compile-time number).
// DENSE: the kernel is built for four waves per SIMD (128 registers: 8-byte lanes of floats get there with a row less of lookahead) --
// two workgroups of eight waves share a CU and its 160 KB of LDS, and sixteen waves hide what eight cannot.
template <typename Real, int X, bool EDGE, bool PVDMA, int LB, bool DENSE>
__device__ __forceinline__ void triple_march_body(const TripleArgs<Real>& a) {
    static_assert(!PVDMA || LB == 8, "the DMA form is written for 8-byte lanes");
    using V = VecL<Real, LB>;
    constexpr int VX = V::N;
    constexpr int RY = kTripleRows;
    constexpr int R0 = RY + 6, R1 = RY + 4, R2 = RY + 2;
    constexpr int NE = kTripleEdgeRows, WS = triple_max_waves(LB) + 2;
    extern __shared__ __attribute__((aligned(16))) char triple_lds[];
    typedef uint32_t U2 __attribute__((ext_vector_type(2)));
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    typedef typename std::conditional<LB == 8, U2, U4>::type UL;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    int row_waves = a.nw, wave_first = 0, store_lo = 0, store_hi = 1 << 30;
    if (a.windows) {  // windows outermost: the workgroups of one window are the grid of a narrow mesh
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
        strip = (int)(u & 0x3FFFu);
        chunk = (int)((u >> 14) & 0x1FFu);
        wave_first = (int)((u >> 23) & 0xFu);
        row_waves = (int)(u >> 27) + 1;
    } else {
        strip = xcd * a.strips_per_xcd + j % a.strips_per_xcd;
        chunk = j / a.strips_per_xcd;
    }
    if (wave >= row_waves) return;  // (a finished wave does not hold up the barriers)
    if (strip >= a.strips || chunk >= a.chunks) return;  // whole workgroup
    const int wave_abs = wave_first + wave;
    const bool stores = wave_abs >= store_lo && wave_abs < store_hi;
    const int y0 = strip * RY;
    const int zb = a.z_begin + chunk * a.zc, ze = min(zb + a.zc, a.z_end);
    if (zb >= ze) return;
    const int zb2 = zb - (zb == a.z_begin ? a.z2_lo : 0), ze2 = ze + (ze == a.z_end ? a.z2_hi : 0);  // t+2 is stored on [zb2, ze2)
    const int pitch = a.pitch;
    const int64_t plane = (int64_t)pitch * a.ny;

    // LDS: the "lo" planes, slot k of this wave = lo[k * 64]; then the edge columns [wave slot][side][set][row], 8 B each:
    // wave w's first vector (lane 0) in slot w + 1 side 0, its last vector (lane 63) in side 1; slots 0 and row_waves + 1 stay zero
    // (what lies beyond the row's ends -- or beyond a window's halo wave: see TripleArgs).  Both sets and all rows of one (slot, side)
    // lie within 288 bytes, and what a wave reads -- slot w side 1, slot w + 2 side 0 -- within 1 KB: one address register for all of it.
    constexpr int SLOTS = kTripleLoSlots + (PVDMA ? kTriplePvSlots : 0);
    V* const lo = reinterpret_cast<V*>(triple_lds) + (size_t)wave * SLOTS * 64 + lane;
    V* const edge = reinterpret_cast<V*>(triple_lds + (size_t)a.nw * SLOTS * 64 * LB);
    auto edge_at = [&](int set, int row, int slot, int side) -> V* { return edge + ((slot * 2 + side) * 2 + set) * NE + row; };
    V zero;
#pragma unroll
    for (int k = 0; k < VX; ++k) zero[k] = Real(0);
    for (int k = threadIdx.x; k < 2 * NE * WS * 2; k += 64 * row_waves) edge[k] = zero;
#pragma unroll
    for (int k = 0; k < kTripleLoSlots; ++k) lo[k * 64] = zero;
    lds_barrier();

    const int row_bytes = pitch * (int)sizeof(Real);
    const uint32_t plane_bytes = (uint32_t)row_bytes * (uint32_t)a.ny;
    const uint32_t lane_off = (uint32_t)((wave_abs * 64 + lane) * LB);
    const uint32_t voff0 = lane_off + (uint32_t)((y0 - 3) * row_bytes);  // row y0-3 of a plane (may wrap: then it is off the grid)
    auto in_y = [&](int y) { return (unsigned)y < (unsigned)a.ny; };
    auto in_z = [&](int z) { return (unsigned)z < (unsigned)a.nz; };
    auto plane_of = [&](const Real* base, int z, bool wanted) {
        const bool ok = wanted && in_z(z);
        const Real* p = base + (int64_t)(ok ? ((X & TX_WRAP_Z) ? (z & 1) : z) : 0) * plane;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<Real*>(p), 0, ok ? plane_bytes : 0u, 0x00020000);
    };
    using Rsrc = decltype(plane_of(a.cur, 0, true));
    auto made_up = [&](int y, int z) -> V {  // (TX_NO_MEMORY)
        V v;
#pragma unroll
        for (int k = 0; k < VX; ++k) v[k] = Real(y) * Real(0.001) + Real(z + k + lane);
        return v;
    };
    // row q of the plane: y = y0 - 3 + q.  EDGE: the row may lie off the grid, its whole offset goes through the range check;
    // otherwise the row's offset is scalar and the lanes' offset is one register for every access of the kernel.
    auto ld = [&](Rsrc r, int q, int z_for_made_up) -> V {
        if (X & TX_NO_MEMORY) return made_up(q, z_for_made_up);
        const uint32_t voff = EDGE ? voff0 + (uint32_t)(q * row_bytes) : lane_off;
        const int soff = EDGE ? 0 : (y0 - 3 + q) * row_bytes;
        UL raw;
        if constexpr (LB == 8)
            raw = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        else
            raw = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        return __builtin_bit_cast(V, raw);
    };
    // `wanted`: a store only some lanes want -- the others aim beyond the plane and the range check drops them
    auto st = [&](Rsrc r, int q, const V& v, bool wanted) {
        const uint32_t voff = wanted ? (EDGE ? voff0 + (uint32_t)(q * row_bytes) : lane_off) : 0xFFFFFFF0u;
        const int soff = EDGE ? 0 : (y0 - 3 + q) * row_bytes;
        if constexpr (LB == 8)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(UL, v), r, voff, soff, (X & TX_STORE_CACHED) ? 0 : 2 /* nt */);
        else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(UL, v), r, voff, soff, (X & TX_STORE_CACHED) ? 0 : 2 /* nt */);
    };
    // (PVDMA) rows y0-2+2p and y0-1+2p of `previous` into pv slots 2p, 2p+1: lanes 0..31 fetch the first row, 16 bytes each, lanes 32..63
    // the second; the 1024 bytes land lane by lane, i.e. row by row
    typedef __attribute__((address_space(3))) char LdsChar;
    const uint32_t dma_voff = (uint32_t)(wave_abs * 64 * LB + (lane & 31) * 16) + (uint32_t)(lane >> 5) * (uint32_t)row_bytes;
    auto dma_pv = [&](Rsrc r, int p) {
        LdsChar* const dst = (LdsChar*)triple_lds + (wave * SLOTS + kTripleLoSlots + 2 * p) * 64 * LB;
        if (EDGE)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, dma_voff + (uint32_t)((y0 - 2 + 2 * p) * row_bytes), 0, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, dma_voff, (y0 - 2 + 2 * p) * row_bytes, 0, 0);
    };
    // 2-bit codes of this lane's columns in the RY rows of the strip on plane z: one dword load (byte r = row r, 4 columns), through a
    // descriptor of its own (the map of 2^32 nodes is 1 GB: within a descriptor's reach)
    const auto r_map = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.map), 0,
                                                         (uint32_t)(((int64_t)a.nz * ((a.ny + 3) >> 2) * a.cls_pitch) * 4), 0x00020000);
    auto codes_of = [&](int z) -> uint32_t {
        const uint32_t map_lane_off = (lane_off / (uint32_t)(sizeof(Real) * 4)) * 4u;  // the word of this lane's columns
        return __builtin_amdgcn_raw_buffer_load_b32(r_map, map_lane_off, (int)((z * ((a.ny + 3) >> 2) + (y0 >> 2)) * a.cls_pitch * 4), 0);
    };
    auto row_codes = [&](uint32_t word, int r) -> uint32_t {  // bits 2k, 2k+1: column k of this lane
        const uint32_t byte = (word >> (r * 8)) & 0xFFu;
        return VX == 4 ? byte : (VX == 2 ? ((byte >> ((lane & 1) * 4)) & 0xFu) : ((byte >> ((lane & 3) * 2)) & 0x3u));
    };
    // x edges: the last column of the wave to the left, the first column of the wave to the right (zeros beyond the row's ends)
    auto edge_l = [&](int set, int row) -> Real { return reinterpret_cast<const Real*>(edge_at(set, row, wave, 1))[VX - 1]; };
    auto edge_r = [&](int set, int row) -> Real { return reinterpret_cast<const Real*>(edge_at(set, row, wave + 2, 0))[0]; };

    // Register planes: "mid" and "new" of each level change places every trip (the loop is unrolled by two).
    V bA[R0], bB[R0], uA[R1], uB[R1], wA[R2], wB[R2];
    // inf / nan among the values kept: the largest |t+3 value|'s top 32 bits, one max per value; what it means is looked at once, after the
    // march.  t+3 alone will do: an inf / nan at t+1 or t+2 that the march is responsible for sits at a node whose six neighbours all
    // take the 7-point update, and their next level inherits it.  (A class test per kept value and an OR of lane masks held the kernel at
    // 32 spilled registers.)
    uint32_t top_exp = 0;

    // One trip.  b: `current`, u: t+1, w: t+2.  *_mid = the level's plane at f-1 / f-2 / f-3, *_new receives f / f-1 / f-2.
    // LDS slots: 0..7 current(f-2) rows y0-2.., 8..13 t+1(f-3) rows y0-1.., 14..17 t+2(f-4) rows y0..
    // A row of t+1 / t+2 whose plane lies off the grid (or, EDGE, the row itself) is forced to zero -- what a missing neighbour counts as.
    // Loads are issued in the order they are needed, about four rows ahead of their use, one row of each field per row step -- across
    // the trips' boundaries too: the first four rows of the NEXT trip are asked for in the last four row steps of this one, into the
    // registers of "mid" rows that have retired (all of a trip's loads at its top would be registers waiting for their turn, and a trip
    // that opens with its first loads opens with a memory latency, every wave of the workgroup at the same time).
    // rows of lookahead (the edge strips' offsets live in vector registers: they have fewer to spare; so have floats on 16-byte lanes, four
    // values per register row to shift and mask: 256 registers + 20 B of scratch with four rows, 249 and none with three)
    constexpr int LA = (EDGE ? 3 : 4) - (DENSE ? 1 : 0) - ((sizeof(Real) == 4 && LB == 16) ? 1 : 0);
    auto trip = [&](int f, int set, V(&b_mid)[R0], V(&b_new)[R0], V(&u_mid)[R1], V(&u_new)[R1], V(&w_mid)[R2], V(&w_new)[R2], V(&pv)[R1],
                    V(&pv_next)[R1]) {
        const int z = f - 3;
        const bool storing = z >= zb && z < ze && stores;  // (warm-up trips and halo waves produce nothing)
        const bool storing2 = z >= zb2 && stores;          // (t+2: a slab's extra plane at either end)
        const Rsrc r_cur = plane_of(a.cur, f, true), r_prev = plane_of(a.prev, f - 1, true);
        const Rsrc r_cur_n = plane_of(a.cur, f + 1, true), r_prev_n = plane_of(a.prev, f, true);
        const Rsrc r_o1 = plane_of(a.out1, z, storing), r_o2 = plane_of(a.out2, z, storing2), r_o3 = plane_of(a.out3, z, storing);
        const uint32_t code_word = codes_of(min(max(z, 0), a.nz - 1));
        const bool z1 = in_z(f - 1), z2 = in_z(f - 2);
        // the edge columns of this trip's centre planes, published by every wave at the end of the last trip
        if (PVDMA)
            lds_barrier_keep_vm();
        else
            lds_barrier();
#pragma unroll
        for (int m = 0; m <= R1; ++m) {
            __builtin_amdgcn_sched_barrier(0);  // (rows in the order written: the scheduler would hoist every LDS read of the trip to its top)
            if (m + LA + 1 < R0) b_new[m + LA + 1] = ld(r_cur, m + LA + 1, f);
            if (m + LA + 1 == R0) b_new[0] = ld(r_cur, 0, f);  // (row y0-3: the next trip's first y-1 neighbour)
            if (PVDMA && !(X & TX_NO_MEMORY)) {
                // rows m-2, m-1 of `previous` have been used: the next trip's take their slots.  (First in its region, and the
                // region after it fenced: what follows counts as issued behind it, whatever order the scheduler likes.)
                if (m >= 2 && (m & 1) == 0) {
                    dma_pv(r_prev_n, m / 2 - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // rows m, m+1 are about to be read: their DMA, issued a trip ago, has to have landed
                if (m < R1 && (m & 1) == 0) {
                    if (m == 0) wait_vmcnt<triple_dma_wait(LA, 0)>();
                    if (m == 2) wait_vmcnt<triple_dma_wait(LA, 1)>();
                    if (m == 4) wait_vmcnt<triple_dma_wait(LA, 2)>();
                    if (m == 6) wait_vmcnt<triple_dma_wait(LA, 3)>();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!PVDMA && m + LA < R1) pv[m + LA] = ld(r_prev, m + LA + 1, 1 - f);
            if (m >= R1 + 1 - LA) {  // the next trip's first rows: its b_new is this trip's b_mid, whose rows 1 .. LA have retired
                b_mid[m - (R1 - LA)] = ld(r_cur_n, m - (R1 - LA), f + 1);
                if (!PVDMA) pv_next[m - (R1 + 1 - LA)] = ld(r_prev_n, m - (R1 - LA), -f);
            }
            V zm_b = zero;
            if (m < R1) {  // t+1 on plane f-1, row y0-2+m
                const int i = m;
                zm_b = lo[i * 64];
                const V own = (PVDMA && !(X & TX_NO_MEMORY)) ? lo[(kTripleLoSlots + i) * 64] : ((PVDMA) ? made_up(i, 1 - f) : pv[i]);
                V v = triple_step_row<Real, LB>(b_mid[i + 1], b_mid[i], b_mid[i + 2], zm_b, b_new[i + 1], own, edge_l(set, i), edge_r(set, i));
                const bool ok = (!EDGE || in_y(y0 - 2 + i)) && z1;
#pragma unroll
                for (int k = 0; k < VX; ++k) v[k] = ok ? v[k] : Real(0);
                u_new[i] = v;
            }
            V zm_u = zero;
            if (m >= 1 && m - 1 < R2) {  // t+2 on plane f-2, row y0-1+r; own old value: current(f-2), slot r+1 -- read for the t+1 row above
                const int r = m - 1;
                zm_u = lo[(R1 + r) * 64];
                V v = triple_step_row<Real, LB>(u_mid[r + 1], u_mid[r], u_mid[r + 2], zm_u, u_new[r + 1], zm_b, edge_l(set, R1 + r), edge_r(set, R1 + r));
                const bool ok = (!EDGE || in_y(y0 - 1 + r)) && z2;
#pragma unroll
                for (int k = 0; k < VX; ++k) v[k] = ok ? v[k] : Real(0);
                w_new[r] = v;
            }
            if (m >= 2 && m - 2 < RY) {  // t+3 on plane z = f-3, row y0+s; own old value: t+1(f-3), slot r = s+1 -- read just above
                const int s = m - 2;
                const V zm_w = lo[(R1 + R2 + s) * 64];
                const V v3 = triple_step_row<Real, LB>(w_mid[s + 1], w_mid[s], w_mid[s + 2], zm_w, w_new[s + 1], zm_u, edge_l(set, R1 + R2 + s),
                                                   edge_r(set, R1 + R2 + s));
                V o1 = zm_u, o2 = w_mid[s + 1], o3 = v3;
                const uint32_t codes = row_codes(code_word, s);
                const uint32_t bits = codes | (codes >> 1);   // bit 2k: column k is not "none"
                const uint32_t want1 = codes & (codes >> 1);  // ... is a shell node (code 3): its t+1 is stored
#pragma unroll
                for (int k = 0; k < VX; ++k) {
                    const bool live = (bits >> (2 * k)) & 1u;
                    o1[k] = live ? o1[k] : Real(0);  // (a "none" column beside a shell one: every field holds 0 there)
                    o2[k] = live ? o2[k] : Real(0);
                    o3[k] = live ? o3[k] : Real(0);
                    top_exp = max(top_exp, abs_bits_hi(o3[k]));
                }
                if (X & TX_NO_MEMORY) {
                    if (top_exp == 0x7FFFFFFFu) {  // (never: the made-up values stay finite)
                        st(r_o2, 3 + s, o2, true);
                        st(r_o3, 3 + s, o3, true);
                    }
                } else {
                    st(r_o1, 3 + s, o1, (want1 & 0x55u) != 0);
                    st(r_o2, 3 + s, o2, true);
                    st(r_o3, 3 + s, o3, true);
                }
            }
            // rows of the "mid" planes nobody needs as a centre row any more take their place in the "lo" planes of the next trip
            if (m >= 1) lo[(m - 1) * 64] = b_mid[m];                              // current(f-1) row y0-2+(m-1)
            if (m >= 2 && m - 2 < R2) lo[(R1 + m - 2) * 64] = u_mid[m - 1];        // t+1(f-2) row y0-1+(m-2)
            if (m >= 3 && m - 3 < RY) lo[(R1 + R2 + m - 3) * 64] = w_mid[m - 2];   // t+2(f-3) row y0+(m-3)
        }
        __builtin_amdgcn_sched_barrier(0);
        // next trip's centre planes are complete: their edge columns, into the other set
        if (lane == 0 || lane == 63) {
            V* const p = edge_at(set ^ 1, 0, wave + 1, lane == 63 ? 1 : 0);
#pragma unroll
            for (int i = 0; i < R1; ++i) p[i] = b_new[i + 1];
#pragma unroll
            for (int r = 0; r < R2; ++r) p[R1 + r] = u_new[r + 1];
#pragma unroll
            for (int s = 0; s < RY; ++s) p[R1 + R2 + s] = w_new[s + 1];
        }
    };

    // ---- prologue: current(zb-3) in the lo slots, current(zb-2) = b_mid; the t+1 / t+2 planes hold zeros.  Four warm-up trips
    // (f = zb-1 .. zb+2) fill the pipeline; what they compute from those zeros is either right (planes off the grid) or overwritten
    // before anything is stored.
    {
        const Rsrc r2 = plane_of(a.cur, zb - 2, true), r3 = plane_of(a.cur, zb - 3, true);
#pragma unroll
        for (int q = 0; q < R0; ++q) bA[q] = ld(r2, q, zb - 2);
#pragma unroll
        for (int i = 0; i < R1; ++i) lo[i * 64] = ld(r3, i + 1, zb - 3);
    }
#pragma unroll
    for (int q = 0; q < R1; ++q) uA[q] = zero;
#pragma unroll
    for (int q = 0; q < R2; ++q) wA[q] = zero;
    if (lane == 0 || lane == 63) {
        V* const p = edge_at(0, 0, wave + 1, lane == 63 ? 1 : 0);
#pragma unroll
        for (int i = 0; i < R1; ++i) p[i] = bA[i + 1];
    }
    V pA[R1], pB[R1];
    {  // the first trip's first rows
        const Rsrc r_cur = plane_of(a.cur, zb - 1, true), r_prev = plane_of(a.prev, zb - 2, true);
#pragma unroll
        for (int q = 0; q < LA; ++q) {
            bB[q + 1] = ld(r_cur, q + 1, zb - 1);
            if (!PVDMA) pA[q] = ld(r_prev, q + 1, 2 - zb);
        }
        if (PVDMA && !(X & TX_NO_MEMORY)) {
#pragma unroll
            for (int p = 0; p < 4; ++p) dma_pv(r_prev, p);
            wait_vmcnt<0>();  // (the waits inside the trips count on a whole trip's instructions between a DMA and its use)
        }
    }
    for (int f = zb - 1; f <= ze2 + 2; f += 2) {
        trip(f, 0, bA, bB, uA, uB, wA, wB, pA, pB);
        if (f + 1 <= ze2 + 2) trip(f + 1, 1, bB, bA, uB, uA, wB, wA, pB, pA);
    }
    // inf / nan among the values this workgroup kept: say so, triple_flags_kernel (launched behind every march) works out the exact bits
    if (__any(top_exp >= (sizeof(Real) == 8 ? 0x7FF00000u : 0x7F800000u)) && lane == 0) atomicOr(a.suspect, 1);
}

template <typename Real, int X = 0, bool PVDMA = false, int LB = kTripleLaneBytes, bool DENSE = false>
__global__ void __launch_bounds__(DENSE ? 1024 : 64 * triple_max_waves(LB)) triple_march_kernel(const TripleArgs<Real> a) {
    // (which strip: as in the body)
    int j = (int)(blockIdx.x >> 3);
    if (a.windows) j %= (int)(gridDim.x >> 3) / a.windows;
    int strip = (int)(blockIdx.x & 7) * a.strips_per_xcd + j % a.strips_per_xcd;
    if (a.unit_list) {
        const uint32_t first = a.list_start[blockIdx.x & 7], count = a.list_start[(blockIdx.x & 7) + 1] - first;
        if ((uint32_t)j >= count) return;
        strip = (int)(a.unit_list[first + (uint32_t)j] & 0x3FFFu);
    }
    const int y0 = strip * kTripleRows;
    if (y0 < 4 || y0 + kTripleRows + 2 >= a.ny)
        triple_march_body<Real, X, true, PVDMA, LB, DENSE>(a);
    else
        triple_march_body<Real, X, false, PVDMA, LB, DENSE>(a);
}

// ---- the triple map and the third level's fix-up list, once per (mesh, source node, receiver set) ------------------------------
// From the two-step pass's map (pair_map_kernel: 3 = takes the 7-point update and has a neighbour that does not, or the source node for
// a neighbour): a node that takes the 7-point update is a SHELL node when it or one of its six neighbours has that code -- something
// other than a plain node lies within two nodes of it -- and deep otherwise.  Shell nodes are the third level's fix-up list (their t+3
// is recomputed from the finished t+2 field; those with pair code 3 are the second level's list, as in a two-step pass).  One thread
// per map byte (four nodes of a row); lists are written in map order, by a block scan (count pass, host scan over blocks, fill pass).
struct TripleMapArgs {
    const uint8_t* pair_map;
    uint8_t* map;
    uint32_t* block_count;   // count pass: [blocks] listed nodes per block; fill pass: exclusive offsets
    uint32_t* list;          // fill pass: stored indices of the shell nodes; null: count pass
    int ny, nz, pitch, cls_pitch;
    int z_begin, z_end;      // planes this engine owns (nodes outside them are not listed)
    const uint32_t* covered; // bit per stored node: finished at the third level by an x-facing wall's entry, not listed; null: none
};

__global__ void __launch_bounds__(256) triple_map_kernel(const TripleMapArgs a) {
    __shared__ uint32_t scan[256];
    const int64_t n_bytes = (int64_t)a.cls_pitch * a.ny * a.nz;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t out = 0, mine = 0, listed_bits = 0;
    int xb = 0, y = 0, z = 0;
    if (i < n_bytes) {
        xb = (int)(i % a.cls_pitch);
        const int64_t row = i / a.cls_pitch;
        y = (int)(row % a.ny);
        z = (int)(row / a.ny);
        auto code_at = [&](int x, int yy, int zz) -> uint32_t {  // off the grid: plain
            if (x < 0 || x >= a.pitch || yy < 0 || yy >= a.ny || zz < 0 || zz >= a.nz) return 1u;
            return (a.pair_map[cls_byte_index(x, yy, zz, a.ny, a.cls_pitch)] >> ((x & 3) * 2)) & 3u;
        };
        for (int k = 0; k < 4; ++k) {
            const int x = xb * 4 + k;
            const uint32_t pc = code_at(x, y, z);
            uint32_t code = pc;
            if (pc & 1u) {
                const bool shell = pc == 3u || code_at(x - 1, y, z) == 3u || code_at(x + 1, y, z) == 3u || code_at(x, y - 1, z) == 3u ||
                                   code_at(x, y + 1, z) == 3u || code_at(x, y, z - 1) == 3u || code_at(x, y, z + 1) == 3u;
                code = shell ? 3u : 1u;
                bool listed = shell && z >= a.z_begin && z < a.z_end;
                if (listed && a.covered) {
                    const uint32_t node = (uint32_t)(((int64_t)z * a.ny + y) * a.pitch + x);
                    listed = !((a.covered[node >> 5] >> (node & 31u)) & 1u);
                }
                if (listed) {
                    ++mine;
                    listed_bits |= 1u << k;
                }
            }
            out |= code << (2 * k);
        }
        if (!a.list) a.map[cls_byte_index(xb * 4, y, z, a.ny, a.cls_pitch)] = (uint8_t)out;
    }
    // exclusive scan of `mine` over the block
    scan[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)d ? scan[threadIdx.x - d] : 0u;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    if (!a.list) {
        if (threadIdx.x == 255) a.block_count[blockIdx.x] = scan[255];
        return;
    }
    uint32_t at = a.block_count[blockIdx.x] + scan[threadIdx.x] - mine;
    if (i < n_bytes)
        for (int k = 0; k < 4; ++k)
            if ((listed_bits >> k) & 1u) a.list[at++] = (uint32_t)(((int64_t)z * a.ny + y) * a.pitch + xb * 4 + k);
}

// receiver nodes and the source node that came out deep: shell all the same -- their t+1 has to be in the t+1 field (the source's
// sample goes into it, a receiver reads it).  After the list has been made: they need no fix-up.
struct TripleMarkArgs {
    uint8_t* map;
    const uint64_t* recv;  // stored indices, ~0 = not recorded
    uint32_t n_recv;
    uint64_t source_node;  // ~0 = none
    int ny, pitch, cls_pitch;
};

__global__ void __launch_bounds__(256) triple_mark_kernel(const TripleMarkArgs a) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > a.n_recv) return;
    const uint64_t idx = t == a.n_recv ? a.source_node : a.recv[t];
    if (idx == ~0ull) return;
    const int x = (int)(idx % (uint64_t)a.pitch);
    const uint64_t row = idx / (uint64_t)a.pitch;
    const int64_t at = cls_byte_index(x, (int)(row % (uint64_t)a.ny), (int)(row / (uint64_t)a.ny), a.ny, a.cls_pitch);
    uint32_t* const word = reinterpret_cast<uint32_t*>(a.map + (at & ~(int64_t)3));
    const uint32_t shift = (uint32_t)(at & 3) * 8u + (uint32_t)(x & 3) * 2u;
    // code 1 -> 3 (one more bit); every other code stays
    if (((*word >> shift) & 3u) == 1u) atomicOr(word, 2u << shift);
}

// ---- exact error bits of the values the march is the last to write, when it has seen an inf or a nan (never in a healthy run) -------
// Launched behind every march; leaves at once unless `suspect` is set.  t+1 of every node that takes the 7-point update (recomputed:
// a deep node's t+1 is not stored); t+2 where nothing but plain nodes lies within one node, t+3 within two (the placeholders' owners --
// fix-up lists, boundary launches -- test their own values).
template <typename Real>
struct TripleFlagsArgs {
    const Real *prev, *cur, *out2, *out3;
    const uint8_t* pair_map;  // 1: the node and its six neighbours are plain
    const int* suspect;
    int *flag1, *flag2, *flag3;
    uint64_t source_node;
    int nx, ny, nz, pitch, cls_pitch;
    int z_begin, z_end;
};

template <typename Real>
__device__ __forceinline__ void triple_flags_body(const TripleFlagsArgs<Real>& a) {
    if (*a.suspect == 0) return;
    const int64_t plane = (int64_t)a.pitch * a.ny;
    const int64_t n = plane * (a.z_end - a.z_begin);
    auto pc_at = [&](int x, int y, int z) -> uint32_t {
        if (x < 0 || x >= a.pitch || y < 0 || y >= a.ny || z < 0 || z >= a.nz) return 1u;
        return (a.pair_map[cls_byte_index(x, y, z, a.ny, a.cls_pitch)] >> ((x & 3) * 2)) & 3u;
    };
    auto at = [&](const Real* f, int x, int y, int z) -> Real {
        if (x < 0 || x >= a.nx || y < 0 || y >= a.ny || z < 0 || z >= a.nz) return Real(0);
        return f[(int64_t)z * plane + (int64_t)y * a.pitch + x];
    };
    int bad1 = 0, bad2 = 0, bad3 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.pitch);
        const int64_t q = i / a.pitch;
        const int y = (int)(q % a.ny), z = a.z_begin + (int)(q / a.ny);
        const uint32_t pc = pc_at(x, y, z);
        if (!(pc & 1u)) continue;
        Real s = Real(0) + at(a.cur, x - 1, y, z);
        s += at(a.cur, x + 1, y, z);
        s += at(a.cur, x, y - 1, z);
        s += at(a.cur, x, y + 1, z);
        s += at(a.cur, x, y, z - 1);
        s += at(a.cur, x, y, z + 1);
        s = div3(s);
        s -= at(a.prev, x, y, z);
        bad1 |= bad_bits(s);
        if (pc != 1u) continue;  // t+2 is somebody else's (a fix-up list, a boundary entry)
        bad2 |= bad_bits(at(a.out2, x, y, z));
        const bool deep = pc_at(x - 1, y, z) == 1u && pc_at(x + 1, y, z) == 1u && pc_at(x, y - 1, z) == 1u && pc_at(x, y + 1, z) == 1u &&
                          pc_at(x, y, z - 1) == 1u && pc_at(x, y, z + 1) == 1u;
        if (deep) bad3 |= bad_bits(at(a.out3, x, y, z));
    }
    if (bad1) atomicOr(a.flag1, bad1);
    if (bad2) atomicOr(a.flag2, bad2);
    if (bad3) atomicOr(a.flag3, bad3);
}

template <typename Real>
__global__ void __launch_bounds__(256) triple_flags_kernel(const TripleFlagsArgs<Real> a) {
    triple_flags_body<Real>(a);
}

// The third level's fix-up list and the exact-flags check in ONE launch (both want the fields t-1 and t of the pass intact and nothing
// else of each other; a launch is 5 us, which is something at 256^3).
template <typename Real>
__global__ void __launch_bounds__(256) triple_list_kernel(const PairFixupArgs<Real> f, const TripleFlagsArgs<Real> g) {
    pair_fixup_node<Real>(f, blockIdx.x * blockDim.x + threadIdx.x);
    triple_flags_body<Real>(g);
}

}  // namespace wv
