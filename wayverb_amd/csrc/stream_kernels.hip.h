// stream_kernels.hip.h -- the per-step pressure update for every non-boundary node.
//
// Replaces the `normal_waveguide_update` arm of `condensed_waveguide`
// (src/waveguide/src/program.cpp:393-412, :494-530) and its id_none arm (:485).  Boundary nodes
// (class 2) are left to boundary_kernels.hip.h; the two kernels write disjoint nodes of
// `prev` and read only `cur`, so they may run in either order or concurrently.
//
// Arithmetic per updated node, in the pressure type Real, no FMA contraction:
//     s = 0; s += nx; s += px; s += ny; s += py; s += nz; s += pz   (off-grid ports contribute +0,
//     s = s / 3;  next = s - prev                                    which equals skipping them)
//
// Bound: HBM.  Algorithmic traffic per node = read prev + read cur + write next = 3*sizeof(Real).
#pragma once
#include "device_common.hip.h"

namespace wv {

// ---------------------------------------------------------------------------------------------
// Variant 0: one node per lane, neighbours straight from global memory (caches do the reuse).
// Kept as the simple baseline the tuned kernel is checked and timed against.
// ---------------------------------------------------------------------------------------------
template <typename Real>
__global__ void __launch_bounds__(256) stream_naive_kernel(const StreamArgs<Real> a) {
    const int64_t plane = (int64_t)a.nx * a.ny;
    const int64_t first = (int64_t)a.z_begin * plane;
    const int64_t count = (int64_t)(a.z_end - a.z_begin) * plane;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = first + i;
        const int x = (int)(idx % a.nx);
        const int64_t q = idx / a.nx;
        const int y = (int)(q % a.ny);
        const int z = (int)(q / a.ny);
        const uint32_t c = (a.cls[(q * a.cls_pitch) + (x >> 2)] >> ((x & 3) * 2)) & 3u;
        if (c == CLS_BOUNDARY) continue;
        Real out = 0;
        if (c & 1u) {
            Real s = 0;
            s += (x > 0) ? a.cur[idx - 1] : Real(0);
            s += (x + 1 < a.nx) ? a.cur[idx + 1] : Real(0);
            s += (y > 0) ? a.cur[idx - a.nx] : Real(0);
            s += (y + 1 < a.ny) ? a.cur[idx + a.nx] : Real(0);
            s += (z > 0) ? a.cur[idx - plane] : Real(0);
            s += (z + 1 < a.nz) ? a.cur[idx + plane] : Real(0);
            s = s / Real(3);
            s -= a.prev[idx];
            out = s;
            bad |= bad_bits(out);
        }
        a.prev[idx] = out;
    }
    if (bad) atomicOr(a.flag, bad);
}

// ---------------------------------------------------------------------------------------------
// Variant 1: wave-autonomous 2.5-D march.
//
// One wave owns a tile of WX = 64*VX contiguous x (VX = 16 B / sizeof(Real) elements per lane,
// so every row access is one fully coalesced 1 KiB wave transaction) by RY rows of y, and
// marches it through `zc` planes of z.  The three z-planes of `cur` it needs live in registers
// and rotate as it advances, so each `cur` value is fetched from HBM once; +-x neighbours
// come from the adjacent lane through DPP wave shifts, +-y from the lane's own registers
// (rows y0-1 and y0+RY are loaded as halo rows), and the two x-edge halo values of each row
// arrive in one 2-address load (lanes 0-31 fetch the left one, 32-63 the right one).  No LDS, no
// barrier: waves of a workgroup are independent, which lets every wave keep a full plane of
// loads (z+2) in flight while it computes plane z.
//
// NW waves stack along y in a workgroup purely for locality (shared halo rows hit L1/L2), and
// the block->tile map keeps each XCD's workgroups on y-adjacent tiles so the halo rows that
// cross workgroups are served by that XCD's L2 instead of HBM.
// ---------------------------------------------------------------------------------------------
// Experiment switches (tools/stream_bench.hip only; the engine always uses 0).  The first three
// change results and exist to price a piece of the kernel; the nt ones are result-neutral.
enum : int { X_NO_EDGE = 1, X_NO_CLS = 2, X_MUL_THIRD = 4, X_NT_STORE = 8, X_NT_PREV = 16, X_NT_CUR = 32,
             X_NO_HALO_ROWS = 64, X_TX_FAST = 128, X_NT_BELOW = 256, X_NT_MID = 512 };
// what the engine runs: x-fastest tile order, non-temporal prev loads and next stores (both are
// touched exactly once per step, so they should not displace the re-used `cur` lines from L2)
constexpr int X_PRODUCT = X_TX_FAST | X_NT_STORE | X_NT_PREV;

template <typename Real, int RY, int NWX, int NWY, int X = X_PRODUCT>
__global__ void __launch_bounds__(64 * NWX * NWY) stream_march_kernel(const StreamArgs<Real> a) {
    using V = typename Vec16<Real>::type;
    constexpr int VX = Vec16<Real>::N;
    constexpr int WX = 64 * VX;

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
    if (y0 >= a.ny || x0 >= a.nx) return;
    const int zb = a.z_begin + cz * a.zc;
    const int ze = min(zb + a.zc, a.z_end);
    if (zb >= ze) return;

    const int xl = x0 + lane * VX;            // first x of this lane
    const bool full = (x0 + WX <= a.nx);      // wave-uniform: whole tile inside the row
    const int64_t plane = (int64_t)a.nx * a.ny;

    // a row of `cur` (zeros when the row or plane is off-grid; lanes past nx read zeros)
    auto load_cur = [&](int y, int z) -> V {
        V v = (V)(Real(0));
        if (y >= 0 && y < a.ny && z >= 0 && z < a.nz) {
            const Real* p = a.cur + (z * plane + (int64_t)y * a.nx + xl);
            if (full) {
                v = (X & X_NT_CUR) ? __builtin_nontemporal_load(reinterpret_cast<const V*>(p))
                                   : *reinterpret_cast<const V*>(p);
            } else {
#pragma unroll
                for (int j = 0; j < VX; ++j)
                    if (xl + j < a.nx) v[j] = p[j];
            }
        }
        return v;
    };
    // x-edge halo of all RY rows of a plane in ONE load: lane r (< RY) fetches cur[x0-1] of row
    // y0+r, lane 32+r fetches cur[x0+WX] of that row; read_lane hands them to lanes 0 / 63 later.
    auto load_edges = [&](int z) -> Real {
        Real e = 0;
        const int r = lane & 31;
        const int y = y0 + r;
        if (!(X & X_NO_EDGE) && r < RY && y < a.ny && z >= 0 && z < a.nz) {
            const int xe = (lane < 32) ? x0 - 1 : x0 + WX;
            if (xe >= 0 && xe < a.nx) e = a.cur[z * plane + (int64_t)y * a.nx + xe];
        }
        return e;
    };
    auto load_prev = [&](int y, int z) -> V {
        V v = (V)(Real(0));
        if (y < a.ny && z < ze) {
            const Real* p = a.prev + (z * plane + (int64_t)y * a.nx + xl);
            if (full) {
                v = (X & X_NT_PREV) ? __builtin_nontemporal_load(reinterpret_cast<const V*>(p))
                                    : *reinterpret_cast<const V*>(p);
            } else {
#pragma unroll
                for (int j = 0; j < VX; ++j)
                    if (xl + j < a.nx) v[j] = p[j];
            }
        }
        return v;
    };
    // 2 class bits per element of this lane (all "boundary" = never stored when off-grid)
    auto load_cls = [&](int y, int z) -> uint32_t {
        uint32_t c = 0xAAu;
        if (X & X_NO_CLS) return 0x55u;
        if (y < a.ny && z < ze && xl < a.nx) {
            const uint8_t byte = a.cls[((int64_t)z * a.ny + y) * a.cls_pitch + (xl >> 2)];
            c = (VX == 4) ? byte : ((byte >> ((lane & 1) * 4)) & 0xFu);
        }
        return c;
    };

    V below[RY];        // cur(z-1), rows y0 .. y0+RY-1
    V mid[RY + 2];      // cur(z),   rows y0-1 .. y0+RY
    Real mid_e;         // x-edge halos of cur(z): lanes r / 32+r hold row y0+r's left / right value
    V above[RY + 2];    // cur(z+1)
    Real above_e;
    V pv[RY];           // prev(z)
    uint32_t cl[RY];

#pragma unroll
    for (int r = 0; r < RY; ++r) below[r] = load_cur(y0 + r, zb - 1);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) mid[r] = ((X & X_NO_HALO_ROWS) && (r == 0 || r == RY + 1)) ? (V)(Real(0)) : load_cur(y0 - 1 + r, zb);
    mid_e = load_edges(zb);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) above[r] = ((X & X_NO_HALO_ROWS) && (r == 0 || r == RY + 1)) ? (V)(Real(0)) : load_cur(y0 - 1 + r, zb + 1);
    above_e = load_edges(zb + 1);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        pv[r] = load_prev(y0 + r, zb);
        cl[r] = load_cls(y0 + r, zb);
    }

    int bad = 0;
    for (int z = zb; z < ze; ++z) {
        // ---- issue the loads of the next iteration first: plane z+2 of cur, plane z+1 of prev
        V nxt[RY + 2];
        Real nxt_e;
        V pv_n[RY];
        uint32_t cl_n[RY];
#pragma unroll
        for (int r = 0; r < RY + 2; ++r) nxt[r] = ((X & X_NO_HALO_ROWS) && (r == 0 || r == RY + 1)) ? (V)(Real(0)) : load_cur(y0 - 1 + r, z + 2);
        nxt_e = load_edges(z + 2);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pv_n[r] = load_prev(y0 + r, z + 1);
            cl_n[r] = load_cls(y0 + r, z + 1);
        }

        // ---- update plane z
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int y = y0 + r;
            if (y < a.ny) {
                const V c0 = mid[r + 1];
                const Real edge_l = read_lane(mid_e, r), edge_r = read_lane(mid_e, 32 + r);
                V out;
                bool skip_any = false;
#pragma unroll
                for (int j = 0; j < VX; ++j) {
                    const Real left = (j == 0) ? lane_from_below(edge_l, c0[VX - 1]) : c0[j - 1];
                    const Real right = (j == VX - 1) ? lane_from_above(edge_r, c0[0]) : c0[j + 1];
                    Real s = Real(0) + left;
                    s += right;
                    s += mid[r][j];
                    s += mid[r + 2][j];
                    s += below[r][j];
                    s += above[r + 1][j];
                    s = (X & X_MUL_THIRD) ? s * (Real(1) / Real(3)) : s / Real(3);
                    s -= pv[r][j];
                    const uint32_t c = (cl[r] >> (2 * j)) & 3u;
                    const Real o = (c & 1u) ? s : Real(0);
                    bad |= bad_bits(o);
                    out[j] = o;
                    skip_any |= (c == CLS_BOUNDARY);
                }
                Real* q = a.prev + (z * plane + (int64_t)y * a.nx + xl);
                if (full && !__any(skip_any)) {
                    if (X & X_NT_STORE) __builtin_nontemporal_store(out, reinterpret_cast<V*>(q));
                    else *reinterpret_cast<V*>(q) = out;
                } else {
#pragma unroll
                    for (int j = 0; j < VX; ++j) {
                        const uint32_t c = (cl[r] >> (2 * j)) & 3u;
                        if (xl + j < a.nx && c != CLS_BOUNDARY) q[j] = out[j];
                    }
                }
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
// Variant 2: plane sweep with L2-resident z-reuse.
//
// Measured on MI355X (tools/stream_bench.hip, profiles/): HBM delivers ~6.5 TB/s to short-lived
// workgroups that are dispatched in address order (the chip-wide set of in-flight addresses is
// a few long contiguous runs), but only ~5.3 TB/s to thousands of long-lived waves that each
// own a private stream -- which is what the z-march is.  This variant keeps the march's lane
// layout (16 B per lane, DPP x-neighbours, one-load x-edges) but makes every wave short-lived:
// a wave updates its WX x RY tile of ONE plane and exits.  The z-reuse of `cur` then has to
// come from cache, so the work is laid out for the per-XCD L2 (4 MiB, private):
//   - XCD k (= blockIdx % 8) owns y-stripe s = pass*8 + k, `stripe_rows` rows tall, and sweeps it
//     through all planes; the three `cur` planes of a stripe (3 * stripe_rows * nx * 8 B) stay in
//     that XCD's L2 while prev/next stream through it with the non-temporal hint;
//   - all 8 XCDs advance through z together, so the chip-wide access front is 8 short runs
//     per stream inside one plane.
// `cur` is then fetched from HBM once per step (plus 2 halo rows per stripe).
// ---------------------------------------------------------------------------------------------
template <typename Real, int RY, int NWX, int NWY, int X = X_NT_STORE | X_NT_PREV>
__global__ void __launch_bounds__(64 * NWX * NWY) stream_sweep_kernel(const StreamArgs<Real> a) {
    using V = typename Vec16<Real>::type;
    constexpr int VX = Vec16<Real>::N;
    constexpr int WX = 64 * VX;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wx = wave % NWX, wy = wave / NWX;

    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int per_plane = a.tiles_x * a.tiles_y_stripe;
    const int tl = j % per_plane;
    j /= per_plane;
    const int nzr = a.z_end - a.z_begin;
    const int z = a.z_begin + j % nzr;
    const int pass = j / nzr;
    const int stripe = pass * 8 + xcd;
    const int tx = tl % a.tiles_x, tyl = tl / a.tiles_x;

    const int y_lo = stripe * a.stripe_rows;
    const int y_hi = min(y_lo + a.stripe_rows, a.ny);
    const int x0 = (tx * NWX + wx) * WX;
    const int y0 = y_lo + (tyl * NWY + wy) * RY;
    if (y0 >= y_hi || x0 >= a.nx) return;

    const int xl = x0 + lane * VX;
    const bool full = (x0 + WX <= a.nx);
    const int64_t plane = (int64_t)a.nx * a.ny;

    auto load_cur = [&](int y, int zz, bool nt) -> V {
        V v = (V)(Real(0));
        if (y >= 0 && y < a.ny && zz >= 0 && zz < a.nz) {
            const Real* p = a.cur + (zz * plane + (int64_t)y * a.nx + xl);
            if (full) {
                v = nt ? __builtin_nontemporal_load(reinterpret_cast<const V*>(p)) : *reinterpret_cast<const V*>(p);
            } else {
#pragma unroll
                for (int jx = 0; jx < VX; ++jx)
                    if (xl + jx < a.nx) v[jx] = p[jx];
            }
        }
        return v;
    };

    // ---- everything this tile needs, issued back to back
    V below[RY], mid[RY + 2], above[RY], pv[RY];
    uint32_t cl[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) above[r] = load_cur(y0 + r, z + 1, (X & X_NT_CUR) != 0);   // first touch: HBM
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        pv[r] = (V)(Real(0));
        cl[r] = 0xAAu;
        const int y = y0 + r;
        if (y < y_hi) {
            const Real* p = a.prev + (z * plane + (int64_t)y * a.nx + xl);
            if (full) {
                pv[r] = (X & X_NT_PREV) ? __builtin_nontemporal_load(reinterpret_cast<const V*>(p))
                                        : *reinterpret_cast<const V*>(p);
            } else {
#pragma unroll
                for (int jx = 0; jx < VX; ++jx)
                    if (xl + jx < a.nx) pv[r][jx] = p[jx];
            }
            if (xl < a.nx) {
                const uint8_t byte = a.cls[((int64_t)z * a.ny + y) * a.cls_pitch + (xl >> 2)];
                cl[r] = (VX == 4) ? byte : ((byte >> ((lane & 1) * 4)) & 0xFu);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) mid[r] = load_cur(y0 - 1 + r, z, (X & X_NT_MID) != 0);  // L2 (loaded as z+1 one plane ago)
    Real mid_e = 0;
    {
        const int r = lane & 31;
        const int y = y0 + r;
        if (r < RY && y < a.ny) {
            const int xe = (lane < 32) ? x0 - 1 : x0 + WX;
            if (xe >= 0 && xe < a.nx) mid_e = a.cur[z * plane + (int64_t)y * a.nx + xe];
        }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) below[r] = load_cur(y0 + r, z - 1, (X & X_NT_BELOW) != 0);  // L2 (two planes ago): last use

    int bad = 0;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const int y = y0 + r;
        if (y < y_hi) {
            const V c0 = mid[r + 1];
            const Real edge_l = read_lane(mid_e, r), edge_r = read_lane(mid_e, 32 + r);
            V out;
            bool skip_any = false;
#pragma unroll
            for (int jx = 0; jx < VX; ++jx) {
                const Real left = (jx == 0) ? lane_from_below(edge_l, c0[VX - 1]) : c0[jx - 1];
                const Real right = (jx == VX - 1) ? lane_from_above(edge_r, c0[0]) : c0[jx + 1];
                Real s = Real(0) + left;
                s += right;
                s += mid[r][jx];
                s += mid[r + 2][jx];
                s += below[r][jx];
                s += above[r][jx];
                s = (X & X_MUL_THIRD) ? s * (Real(1) / Real(3)) : s / Real(3);
                s -= pv[r][jx];
                const uint32_t c = (cl[r] >> (2 * jx)) & 3u;
                const Real o = (c & 1u) ? s : Real(0);
                bad |= bad_bits(o);
                out[jx] = o;
                skip_any |= (c == CLS_BOUNDARY);
            }
            Real* q = a.prev + (z * plane + (int64_t)y * a.nx + xl);
            if (full && !__any(skip_any)) {
                if (X & X_NT_STORE) __builtin_nontemporal_store(out, reinterpret_cast<V*>(q));
                else *reinterpret_cast<V*>(q) = out;
            } else {
#pragma unroll
                for (int jx = 0; jx < VX; ++jx) {
                    const uint32_t c = (cl[r] >> (2 * jx)) & 3u;
                    if (xl + jx < a.nx && c != CLS_BOUNDARY) q[jx] = out[jx];
                }
            }
        }
    }
    if (__any(bad != 0)) {
        if (bad) atomicOr(a.flag, bad);
    }
}

}  // namespace wv
