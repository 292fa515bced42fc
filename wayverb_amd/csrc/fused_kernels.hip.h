// fused_kernels.hip.h -- one launch per time step: the plane sweep of stream_kernels.hip.h with every
// workgroup finishing the BOUNDARY nodes of its own tile, injecting the next step's source sample and
// gathering this step's receivers.
//
// Why.  Below a few hundred MB a step is bound by launches and by the latency of small kernels, not by
// bytes: 64^3 = 13 us per step for 5.6 us of sweep, 256^3 = 93 us for 66 us of HBM time, the rest being
// the boundary launch (1 500 workgroups of dependent loads), the gaps between dependent launches and
// the 64-lane source / receiver launch.  A grid-wide barrier is no way out on this chip (27 us for 256
// workgroups, tools/grid_sync_bench.hip); what is free is that the workgroup that sweeps a tile has just
// pulled the very cache lines its boundary nodes need.  So:
//   * boundary entries are listed per sweep tile (engine.hip, build_tile_boundary_lists); after its
//     stores and a barrier the workgroup runs boundary_entry() -- the same code as boundary_kernel --
//     over its tile's list.  Only the tile's owner ever writes into the tile, so there is no ordering
//     between workgroups to keep; the sweep's "old value written back" at a boundary node is simply
//     overwritten by the same workgroup afterwards.
//   * the source sample of step s+1 is put in place by whoever produces the source node's value in
//     step s (after that value has been tested for inf / nan, which is what the reference's kernel tail
//     tests): hard source -> the sample, soft source -> value + sample.  Nothing reads that node between
//     the end of step s and the injection the reference does at the start of step s+1, so the fields
//     are the same.  The first step of a batch is served by pre_post_kernel, which also resets the flag
//     words of the whole batch.
//   * the receivers of step s read `current`, which is complete and read-only during the launch: the
//     first workgroup gathers them.
// Results are bit-identical to the separate launches (tests/test_gpu_parity.py runs every case both ways).
#pragma once
#include "boundary_kernels.hip.h"
#include "stream_kernels.hip.h"

namespace wv {

template <typename Real>
struct StepIO {
    const double* signal;     // device copy of the source signal
    uint64_t next_pos;        // sample index of the NEXT step (relative to *signal_base when set)
    const uint64_t* signal_base;
    uint64_t source_node;     // stored index, ~0 = none
    int source_kind;          // 0: nothing to inject for the next step; 1 hard; 2 soft
    const uint64_t* recv;     // this step's receivers: read from `cur`
    Real* recv_out;
    uint32_t n_recv;
};

template <typename Real>
struct FusedArgs {
    StreamArgs<Real> s;
    BoundaryArgs<Real> b;     // prev / next / cur as in s
    StepIO<Real> io;
    const uint32_t* tb_start;    // [tiles + 1] first entry of each tile in tb_entries; tile = (z * tiles_y_all + ty) * tiles_x + tx
    const uint32_t* tb_entries;  // entry ids (as boundary_entry takes them), grouped by tile
    int tiles_y_all;             // workgroup tiles along y over the whole mesh
};

template <typename Real>
__device__ __forceinline__ Real inject_next(const StepIO<Real>& io, Real value) {
    const Real s = (Real)io.signal[io.next_pos + (io.signal_base ? *io.signal_base : 0ull)];
    return io.source_kind == 1 ? s : (Real)(value + s);
}

template <typename Real, int RY, int NWX, int NWY>
__global__ void __launch_bounds__(64 * NWX * NWY) stream_fused_kernel(const FusedArgs<Real> f) {
    using V = typename Vec16<Real>::type;
    constexpr int VX = Vec16<Real>::N;
    constexpr int WX = TileIO<Real>::WX;
    constexpr int X = X_SWEEP;
    const StreamArgs<Real>& a = f.s;
    __shared__ V halo[NWY][NWX][2][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wx = wave % NWX, wy = wave / NWX;

    // this step's receivers: `cur` is complete and nobody writes it during this launch
    if (blockIdx.x == 0) {
        for (uint32_t r = threadIdx.x; r < f.io.n_recv; r += 64 * NWX * NWY) {
            const uint64_t node = f.io.recv[r];
            f.io.recv_out[r] = node != ~0ull ? a.cur[node] : Real(0);
        }
    }

    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    int tl, z, stripe;
    uint32_t mask = ~0u;
    bool have_tile = true;
    if (a.tile_list) {
        const uint32_t first = a.list_start[xcd], count = a.list_start[xcd + 1] - first;
        have_tile = (uint32_t)j < count;
        const uint64_t e = have_tile ? a.tile_list[first + (uint32_t)j] : 0ull;
        mask = (uint32_t)(e >> 40) & 0xFFu;
        tl = (int)(e & 0xFFFFFu);
        z = (int)((e >> 20) & 0xFFFFFu);
        stripe = (int)(e >> 48);
    } else {
        const int per_plane = a.tiles_x * a.tiles_y_stripe;
        tl = j % per_plane;
        j /= per_plane;
        const int nzr = a.z_end - a.z_begin;
        // planes in rotated order: the launch ends on planes from the middle of the mesh, not on a wall plane
        // whose tiles are all boundary nodes (consecutive workgroups still take consecutive planes: L2 reuse)
        z = a.z_begin + (j % nzr + nzr / 2) % nzr;
        stripe = (j / nzr) * 8 + xcd;
    }
    if (!have_tile) return;  // whole workgroup
    const int tx = tl % a.tiles_x, tyl = tl / a.tiles_x;

    const int y_lo = stripe * a.stripe_rows;
    const int y_hi = min(y_lo + a.stripe_rows, a.ny);
    // the arithmetic mapping also launches workgroups for stripes / tiles past the end of the mesh: they own no tile
    if (y_lo + tyl * NWY * RY >= y_hi) return;  // whole workgroup
    const int x0 = (tx * NWX + wx) * WX;
    const int y0 = y_lo + (tyl * NWY + wy) * RY;
    auto row_wave_active = [&](int wyy) {
        return y_lo + (tyl * NWY + wyy) * RY < y_hi && ((mask >> (wyy * NWX + wx)) & 1u);
    };
    const bool active = x0 < a.pitch && row_wave_active(wy);
    const bool from_lo = wy > 0 && row_wave_active(wy - 1);
    const bool from_hi = wy + 1 < NWY && row_wave_active(wy + 1);

    const TileIO<Real> io(a, lane, min(x0, a.pitch - WX));
    V below[RY], mid[RY + 2], above[RY], pv[RY];
    uint32_t cl[RY];
    Real mid_e = 0;
    if (active) {
#pragma unroll
        for (int r = 0; r < RY; ++r) mid[r + 1] = io.template cur_row<false>(y0 + r, z);
        if (!from_lo) mid[0] = io.template cur_row<false>(y0 - 1, z);
        if (!from_hi) mid[RY + 1] = io.template cur_row<false>(y0 + RY, z);
#pragma unroll
        for (int r = 0; r < RY; ++r) above[r] = io.template cur_row<false>(y0 + r, z + 1);
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
        for (int r = 0; r < RY; ++r) below[r] = io.template cur_row<false>(y0 + r, z - 1);
        halo[wy][wx][0][lane] = mid[1];
        halo[wy][wx][1][lane] = mid[RY];
    }
    __syncthreads();
    int bad = 0;
    if (active) {
        if (from_lo) mid[0] = halo[wy - 1][wx][1][lane];
        if (from_hi) mid[RY + 1] = halo[wy + 1][wx][0][lane];
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            if (y0 + r < y_hi) {
                bool skip = false;
                V out = update_row<Real, X>(mid[r + 1], mid[r], mid[r + 2], below[r], above[r], pv[r], mid_e, r, cl[r], bad, skip);
                if (f.io.source_kind) {
                    // the next step's sample, if this row holds the source node and this kernel (not the
                    // boundary part below) produces its value
                    const int64_t at = io.at(y0 + r, z);
                    const int64_t d = (int64_t)f.io.source_node - at;
                    if (d >= 0 && d < VX && ((cl[r] >> (2 * (int)d)) & 3u) != CLS_BOUNDARY) {
#pragma unroll
                        for (int k = 0; k < VX; ++k)
                            if (k == (int)d) out[k] = inject_next<Real>(f.io, out[k]);
                    }
                }
                store_row<Real, X>(a.next + io.at(y0 + r, z), out, cl[r], skip);
            }
        }
    }
    if (__any(bad != 0)) {
        if (bad) atomicOr(a.flag, bad);
    }

    // ---- the boundary nodes of this workgroup's tile (global ty: stripes are whole numbers of tiles)
    const int ty = (y_lo / (RY * NWY)) + tyl;
    const uint32_t tile = (uint32_t)((z * f.tiles_y_all + ty) * a.tiles_x + tx);
    const uint32_t first = f.tb_start[tile], last = f.tb_start[tile + 1];
    if (first == last) return;  // whole workgroup: most tiles hold no boundary node
    __syncthreads();            // (also orders the sweep's stores above before the stores below)
    int bbad = 0;
    for (uint32_t e = first + threadIdx.x; e < last; e += 64 * NWX * NWY) {
        const uint32_t entry = f.tb_entries[e];
        boundary_entry<Real>(f.b, f.b.coeffs, entry, bbad);
        if (f.io.source_kind && (uint64_t)f.b.bnode[entry] == f.io.source_node) {
            // the source sits on a boundary node: its new value was stored by boundary_entry just now
            Real* p = f.b.next + f.io.source_node;
            *p = inject_next<Real>(f.io, *p);
        }
    }
    if (bbad) atomicOr(f.b.flag, bbad);
}

}  // namespace wv
