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
template <typename Real, int RY, int NW>
__global__ void __launch_bounds__(64 * NW) stream_march_kernel(const StreamArgs<Real> a) {
    using V = typename Vec16<Real>::type;
    constexpr int VX = Vec16<Real>::N;
    constexpr int WX = 64 * VX;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // XCD-aware tile map: workgroup b runs on XCD b % 8 (observed dispatch, used for locality only)
    const int b = blockIdx.x;
    const int slot = b >> 3;
    const int t = (b & 7) * a.tiles_per_xcd + slot;
    if (slot >= a.tiles_per_xcd || t >= a.total_tiles) return;
    const int ty = t % a.tiles_y;
    const int rem = t / a.tiles_y;
    const int tx = rem % a.tiles_x;
    const int cz = rem / a.tiles_x;

    const int x0 = tx * WX;
    const int y0 = (ty * NW + wave) * RY;
    if (y0 >= a.ny) return;
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
                v = *reinterpret_cast<const V*>(p);
            } else {
#pragma unroll
                for (int j = 0; j < VX; ++j)
                    if (xl + j < a.nx) v[j] = p[j];
            }
        }
        return v;
    };
    // x-edge halo of a row: lanes 0..31 fetch cur[x0-1], lanes 32..63 fetch cur[x0+WX]
    auto load_edge = [&](int y, int z) -> Real {
        Real e = 0;
        if (y >= 0 && y < a.ny && z >= 0 && z < a.nz) {
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
                v = *reinterpret_cast<const V*>(p);
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
        if (y < a.ny && z < ze && xl < a.nx) {
            const uint8_t byte = a.cls[((int64_t)z * a.ny + y) * a.cls_pitch + (xl >> 2)];
            c = (VX == 4) ? byte : ((byte >> ((lane & 1) * 4)) & 0xFu);
        }
        return c;
    };

    V below[RY];        // cur(z-1), rows y0 .. y0+RY-1
    V mid[RY + 2];      // cur(z),   rows y0-1 .. y0+RY
    Real mid_e[RY];     // x-edge halo of cur(z) rows y0 .. y0+RY-1
    V above[RY + 2];    // cur(z+1)
    Real above_e[RY];
    V pv[RY];           // prev(z)
    uint32_t cl[RY];

#pragma unroll
    for (int r = 0; r < RY; ++r) below[r] = load_cur(y0 + r, zb - 1);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) mid[r] = load_cur(y0 - 1 + r, zb);
#pragma unroll
    for (int r = 0; r < RY; ++r) mid_e[r] = load_edge(y0 + r, zb);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) above[r] = load_cur(y0 - 1 + r, zb + 1);
#pragma unroll
    for (int r = 0; r < RY; ++r) above_e[r] = load_edge(y0 + r, zb + 1);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        pv[r] = load_prev(y0 + r, zb);
        cl[r] = load_cls(y0 + r, zb);
    }

    int bad = 0;
    for (int z = zb; z < ze; ++z) {
        // ---- issue the loads of the next iteration first: plane z+2 of cur, plane z+1 of prev
        V nxt[RY + 2];
        Real nxt_e[RY];
        V pv_n[RY];
        uint32_t cl_n[RY];
#pragma unroll
        for (int r = 0; r < RY + 2; ++r) nxt[r] = load_cur(y0 - 1 + r, z + 2);
#pragma unroll
        for (int r = 0; r < RY; ++r) nxt_e[r] = load_edge(y0 + r, z + 2);
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
                V out;
                bool skip_any = false;
#pragma unroll
                for (int j = 0; j < VX; ++j) {
                    const Real left = (j == 0) ? lane_from_below(mid_e[r], c0[VX - 1]) : c0[j - 1];
                    const Real right = (j == VX - 1) ? lane_from_above(mid_e[r], c0[0]) : c0[j + 1];
                    Real s = Real(0) + left;
                    s += right;
                    s += mid[r][j];
                    s += mid[r + 2][j];
                    s += below[r][j];
                    s += above[r + 1][j];
                    s = s / Real(3);
                    s -= pv[r][j];
                    const uint32_t c = (cl[r] >> (2 * j)) & 3u;
                    const Real o = (c & 1u) ? s : Real(0);
                    bad |= bad_bits(o);
                    out[j] = o;
                    skip_any |= (c == CLS_BOUNDARY);
                }
                Real* q = a.prev + (z * plane + (int64_t)y * a.nx + xl);
                if (full && !__any(skip_any)) {
                    *reinterpret_cast<V*>(q) = out;
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
#pragma unroll
        for (int r = 0; r < RY; ++r) below[r] = mid[r + 1];
#pragma unroll
        for (int r = 0; r < RY + 2; ++r) {
            mid[r] = above[r];
            above[r] = nxt[r];
        }
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            mid_e[r] = above_e[r];
            above_e[r] = nxt_e[r];
            pv[r] = pv_n[r];
            cl[r] = cl_n[r];
        }
    }
    if (__any(bad != 0)) {
        if (bad) atomicOr(a.flag, bad);
    }
}

}  // namespace wv
