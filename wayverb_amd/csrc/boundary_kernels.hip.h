// boundary_kernels.hip.h -- locally-reacting boundary nodes with order-6 IIR wall filters,
// source injection / receiver gather, and the one-off mesh set-up kernels.
//
// Replaces `boundary_1/2/3` and their helpers (src/waveguide/src/program.cpp:150-387) and
// `filter_step_canonical` (src/waveguide/src/cl/filters.cpp:17-36,39).
//
// Data layout (HBM): boundary nodes are compacted into one entry list, all 1-D nodes first,
// then 2-D, then 3-D; inside a class the engine orders them by 64x8x8 brick (engine.hip, init) so
// that consecutive lanes touch consecutive memory on the y- and z-walls and share cache lines
// on the x-walls.  Filter state is structure-of-arrays: fmem[j][slot], slot = base_D + i*n_D + k
// for filter i of node k, so the 6 state words of 64 neighbouring nodes are 6 coalesced 512-byte
// rows instead of 64 strided 56-byte structs (cl/structs.h:38-41).
//
// Arithmetic: exactly SURVEY.md Appendix A -- pressure-type (Real) sums, double (filt_real)
// quotients added into Real accumulators, no FMA contraction.
#pragma once
#include "device_common.hip.h"

namespace wv {

// Order-6 transposed direct form II with the reference's zero-coefficient guards
// (filters.cpp:26-35): a rigid wall (a0 == 0) yields a non-finite `out` that every guarded
// product then ignores.
__device__ __forceinline__ void filter_step_6(double in, double m[6], const double* __restrict__ cb,
                                              const double* __restrict__ ca) {
    const double out = (in * cb[0] + m[0]) / ca[0];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double b = cb[i + 1] == 0 ? 0 : cb[i + 1] * in;
        const double a = ca[i + 1] == 0 ? 0 : ca[i + 1] * out;
        m[i] = b - a + m[i + 1];
    }
    const double b = cb[6] == 0 ? 0 : cb[6] * in;
    const double a = ca[6] == 0 ? 0 : ca[6] * out;
    m[5] = b - a;
}

// `filter_test_2` (src/waveguide/src/cl/filters.cpp:66-75; tests/rectangular_kernel.cpp:180): one
// canonical filter per work-item, float in / float out; here all samples of a filter in one launch.
__global__ void __launch_bounds__(64) filter_test_2_kernel(const float* input, float* output, double* memory,
                                                           const double* coeffs, uint32_t n_filters,
                                                           uint32_t n_samples) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_filters) return;
    double m[6], cb[7], ca[7];
    for (int j = 0; j < 6; ++j) m[j] = memory[(size_t)f * 6 + j];
    for (int j = 0; j < 7; ++j) {
        cb[j] = coeffs[(size_t)f * 14 + j];
        ca[j] = coeffs[(size_t)f * 14 + 7 + j];
    }
    for (uint32_t s = 0; s < n_samples; ++s) {
        const double in = (double)input[(size_t)s * n_filters + f];
        // the step's output is needed here, so it is restated rather than taken from filter_step_6
        const double out = (in * cb[0] + m[0]) / ca[0];
        filter_step_6(in, m, cb, ca);
        output[(size_t)s * n_filters + f] = (float)out;
    }
    for (int j = 0; j < 6; ++j) memory[(size_t)f * 6 + j] = m[j];
}

// The arithmetic of one boundary node (program.cpp:331-387 with ghost_point_pressure_update, :150-174), from values
// already loaded: nb = `cur` at -/+ along x, y, z, off = which of those lie off the grid, prev = the node's own old
// value, m / cf = its filters' memories and coefficient sets.  Returns the node's new value; the memories are
// advanced in place.  Shared by every way of loading (boundary_node, xwall_node) so that they cannot differ by a bit.
template <typename Real, int D>
__device__ __forceinline__ Real boundary_value(Real courant, Real courant_sq, uint32_t dirs, const Real (&nb)[3][2],
                                               const bool (&off)[3][2], Real prev, double (&m)[D][6],
                                               const double* const (&cf)[D]) {
    // 2 * inner pressures, x before y before z (program.cpp:19-87, :268-276)
    Real sum = 0;
    bool inner_axis[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const bool has_n = (dirs >> (2 * ax)) & 1u, has_p = (dirs >> (2 * ax + 1)) & 1u;
        inner_axis[ax] = has_n || has_p;
        const Real p = has_p ? (off[ax][1] ? Real(0) : nb[ax][1]) : (off[ax][0] ? Real(0) : nb[ax][0]);
        const Real with = sum + 2 * p;
        sum = inner_axis[ax] ? with : sum;
    }
    // un-doubled in-plane / along-edge neighbours, lower axis first, n before p
    // (program.cpp:112-143, :178-227); off-grid ends the sum at 0 (statically flagged at create)
    Real surr = 0;
    if (D < 3) {
        bool ok = true;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bool take = !inner_axis[ax] && ok;
                const Real with = surr + nb[ax][s];
                surr = take ? (off[ax][s] ? Real(0) : with) : surr;
                ok = ok && !(take && off[ax][s]);
            }
        }
    }
    const Real csw = courant_sq * (sum + surr);

    Real facc = 0, cacc = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) facc = (Real)((double)facc + m[i][0] / cf[i][0]);
#pragma unroll
    for (int i = 0; i < D; ++i) cacc = (Real)((double)cacc + cf[i][7] / cf[i][0]);
    const Real fw = courant_sq * facc;
    const Real cw = cacc * courant;
    const Real pw = (cw - 1) * prev;
    const Real next = (csw + fw + pw) / (1 + cw);

#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double b0 = cf[i][0], a0 = cf[i][7];
        const double diff = (a0 * (double)(Real)(prev - next)) / (b0 * (double)courant) + (m[i][0] / b0);
        filter_step_6(-diff, m[i], cf[i], cf[i] + 7);
    }
    return next;
}

// The 7-point update of an inside node from six loaded neighbours (off-grid ones 0) and its own old value
// (program.cpp:393-412), in the reference's order.
template <typename Real>
__device__ __forceinline__ Real faced_value(const Real (&fnb)[3][2], Real fprev) {
    Real s = 0;
    s += fnb[0][0];
    s += fnb[0][1];
    s += fnb[1][0];
    s += fnb[1][1];
    s += fnb[2][0];
    s += fnb[2][1];
    s = div3(s);
    s -= fprev;
    return s;
}

// One boundary node.  Everything it needs from memory is requested before anything is used: all six
// neighbours of `cur` whatever the node's type (the type only decides which of them enter which sum), its
// own old value, its filters' memories -- and, in the second launch of a two-step pass, what the inside
// node it faces needs.  Selection is by v_cndmask, not by branch: with the loads behind per-lane branches
// (the first form of this kernel) every one of them was waited for on its own, and eight to ten memory
// round trips in a row, not bytes, set the kernel's time.
template <typename Real, int D, bool FIX>
__device__ __forceinline__ void boundary_node(const BoundaryArgs<Real>& a, const double* coeffs, uint32_t k, uint32_t entry,
                                              uint32_t slot_base, uint32_t n_d, int& bad) {
    const uint32_t idx = a.bnode[entry];
    if (idx == INVALID_NODE) return;
    const uint32_t dirs = a.btype[entry];  // bit p set: inner node through port p (nx,px,ny,py,nz,pz)
    const int x = (int)(idx % (uint32_t)a.pitch);
    const uint32_t q = idx / (uint32_t)a.pitch;
    const int y = (int)(q % (uint32_t)a.ny);
    const int z = (int)(q / (uint32_t)a.ny);
    if (z < a.z_begin || z >= a.z_end) return;

    const int64_t plane = (int64_t)a.pitch * a.ny;
    const int64_t stride[3] = {1, a.pitch, plane};
    const int pos[3] = {x, y, z};
    const int lim[3] = {a.nx, a.ny, a.nz};
    const Real* cur = a.cur;

    // ---- loads -----------------------------------------------------------------------------------
    Real nb[3][2];   // cur at -/+ along x, y, z (the node's own value where that is off the grid: unused)
    bool off[3][2];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = pos[ax] + (s ? 1 : -1);
            off[ax][s] = c < 0 || c >= lim[ax];
            nb[ax][s] = cur[(int64_t)idx + (off[ax][s] ? 0 : (s ? stride[ax] : -stride[ax]))];
        }
    }
    const Real prev = a.prev[idx];
    double m[D][6];
    const double* cf[D];
    uint32_t slot[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        slot[i] = slot_base + (uint32_t)i * n_d + k;
        cf[i] = coeffs + (size_t)a.cidx[slot[i]] * 14;
#pragma unroll
        // filter memories are read once and written once per step: keep them from displacing field lines in L2
        for (int j = 0; j < 6; ++j) m[i][j] = __builtin_nontemporal_load(a.fmem + (size_t)j * a.n_slots + slot[i]);
    }
    // two-step pass, second launch: the inside node a 1-D entry faces (boundary_kernel<.., FIX = true>) and the
    // seven values its update t+1 -> t+2 reads
    bool fix = false;
    int64_t fn = idx;
    Real fnb[3][2];
    Real fprev = 0;
    if (D == 1 && FIX) {
        int fp[3] = {x, y, z};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const int step = (int)((dirs >> (2 * ax + 1)) & 1u) - (int)((dirs >> (2 * ax)) & 1u);  // +1, -1 or 0
            fp[ax] += step;
            fn += step * stride[ax];
        }
        fix = fp[0] >= 0 && fp[0] < a.nx && fp[1] >= 0 && fp[1] < a.ny && fp[2] >= a.fix_z0 && fp[2] < a.fix_z1;
        if (!fix) fn = idx;  // (loads below stay inside the field)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c = (fix ? fp[ax] : pos[ax]) + (s ? 1 : -1);
                const bool o = c < 0 || c >= lim[ax];
                const Real v = cur[fn + (o ? 0 : (s ? stride[ax] : -stride[ax]))];
                fnb[ax][s] = o ? Real(0) : v;
            }
        }
        fprev = a.prev[fn];
    }

    // ---- the node's new value, its filters' new memories ---------------------------------------------
    const Real next = boundary_value<Real, D>(a.courant, a.courant_sq, dirs, nb, off, prev, m, cf);
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) __builtin_nontemporal_store(m[i][j], a.fmem + (size_t)j * a.n_slots + slot[i]);
    }
    bad |= bad_bits(next);
    a.next[idx] = next;

    // the faced node: 7-point update from the complete t+1 field (program.cpp:393-412, as pair_fixup_kernel
    // does it for the nodes nobody faces)
    if (D == 1 && FIX) {
        const Real s = faced_value<Real>(fnb, fprev);
        if (fix) {
            bad |= bad_bits(s);
            a.next[fn] = s;
        }
    }
}

// ---- walls that face along x: compact copies instead of field gathers -----------------------------------
// The mesh is x-fastest, so a wall node at (1, y, z) owns a 128-byte line of every field it touches and uses 8-24
// bytes of it; its in-wall neighbours (1, y+-1, z), (1, y, z+-1) each sit in a line of their own.  boundary_node's
// gathers cost such a node 2.4 whole lines in the first launch of a two-step pass and 6 in the second (PMC, round 2),
// where a node of a y- or z-facing wall moves little more than its algorithmic 168 B.  So in two-step passes the
// 1-D entries that face along x (in the marched planes; the first xw_n positions of the entry list, engine_setup)
// keep what they would gather in compact arrays indexed by entry position -- neighbours are then +-1 / +-8 positions
// away in brick order -- and touch the fields for exactly what must cross between wall and march:
//   level 1 (t-1, t -> t+1)  reads one line of the t+1 field: the faced node's value (the march's output) and the node
//                            behind it; writes its own t+1 value into that line
//   level 2 (t, t+1 -> t+2)  reads no field at all; writes its own and the faced node's t+2 values (one line)
// xw_a / xw_b: the wall node's own value at the odd / even time level (t-1 then t+1 / t then t+2, updated in
// place); xw_f: the faced node at the even level; xw_f1 / xw_g: the faced node and the node behind it at t+1, captured
// by level 1 for level 2.  xw_nbr[k][pos], k = y-, y+, z-, z+: position of that in-wall neighbour if it is one of these
// entries (bit 31: it faces the same way, so ITS faced node is the lateral neighbour of mine), else XW_FIELD: read the
// field as boundary_node would.  Same arithmetic (boundary_value / faced_value), same results.
constexpr uint32_t XW_FIELD = 0xFFFFFFFFu, XW_SAME_FACING = 0x80000000u;

template <typename Real, int LEVEL>
__device__ __forceinline__ void xwall_node(const BoundaryArgs<Real>& a, const double* coeffs, uint32_t pos_e, int& bad) {
    const uint32_t idx = a.bnode[pos_e];
    const uint32_t dirs = a.btype[pos_e];  // 1: inner node at x-1, 2: at x+1
    const int x = (int)(idx % (uint32_t)a.pitch);
    const uint32_t q = idx / (uint32_t)a.pitch;
    const int y = (int)(q % (uint32_t)a.ny);
    const int z = (int)(q / (uint32_t)a.ny);
    const int64_t plane = (int64_t)a.pitch * a.ny;
    const int64_t stride[3] = {1, a.pitch, plane};
    const int pos[3] = {x, y, z};
    const int lim[3] = {a.nx, a.ny, a.nz};
    const int step = (dirs & 2u) ? 1 : -1;
    const int64_t fn = (int64_t)idx + step;  // the faced node: in the grid (xwall_eligible_kernel)
    const bool far_off = x + 2 * step < 0 || x + 2 * step >= a.nx;

    // ---- loads -----------------------------------------------------------------------------------
    uint32_t ref[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ref[k] = a.xw_nbr[(size_t)k * a.xw_n + pos_e];
    const Real* level = LEVEL == 1 ? a.xw_b : a.xw_a;  // wall values at the time level the neighbours are read at
    Real nb[3][2];
    bool off[3][2];
    off[0][0] = x - 1 < 0;
    off[0][1] = x + 1 >= a.nx;
    nb[0][0] = nb[0][1] = LEVEL == 1 ? a.xw_f[pos_e] : a.xw_f1[pos_e];  // (only the inner side enters the sums)
    bool any_field = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ax = 1 + (k >> 1), s = k & 1;
        const int c = pos[ax] + (s ? 1 : -1);
        off[ax][s] = c < 0 || c >= lim[ax];
        const bool mirrored = ref[k] != XW_FIELD;
        nb[ax][s] = level[mirrored ? (ref[k] & ~XW_SAME_FACING) : pos_e];
        any_field = any_field || !mirrored;
    }
    const Real prev = LEVEL == 1 ? a.xw_a[pos_e] : a.xw_b[pos_e];
    double m[1][6];
    const double* cf[1];
    cf[0] = coeffs + (size_t)a.cidx[pos_e] * 14;  // (1-D entries: slot = position)
#pragma unroll
    for (int j = 0; j < 6; ++j) m[0][j] = __builtin_nontemporal_load(a.fmem + (size_t)j * a.n_slots + pos_e);
    // level 1: what level 2 will want of the t+1 field (a.next), both in this node's line
    Real f1 = 0, far1 = 0;
    // level 2: the faced node's update
    Real fnb[3][2], fprev = 0;
    bool any_lateral_field = false;
    if (LEVEL == 1) {
        f1 = a.next[fn];
        far1 = a.next[fn + (far_off ? 0 : step)];
        far1 = far_off ? Real(0) : far1;
    } else {
        const Real own1 = a.xw_a[pos_e], far = a.xw_g[pos_e];
        fnb[0][0] = step > 0 ? own1 : far;
        fnb[0][1] = step > 0 ? far : own1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ax = 1 + (k >> 1), s = k & 1;
            const bool regular = ref[k] != XW_FIELD && (ref[k] & XW_SAME_FACING);
            const Real v = a.xw_f1[regular ? (ref[k] & ~XW_SAME_FACING) : pos_e];
            fnb[ax][s] = off[ax][s] ? Real(0) : v;  // (the faced node's y, z are this node's)
            any_lateral_field = any_lateral_field || (!regular && !off[ax][s]);
        }
        fprev = a.xw_f[pos_e];
    }
    // the wall's rim, walls that meet other things than walls: those neighbours come from the fields
    if (any_field) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ax = 1 + (k >> 1), s = k & 1;
            if (ref[k] == XW_FIELD) nb[ax][s] = a.cur[(int64_t)idx + (off[ax][s] ? 0 : (s ? stride[ax] : -stride[ax]))];
        }
    }
    if (LEVEL == 2 && any_lateral_field) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ax = 1 + (k >> 1), s = k & 1;
            const bool regular = ref[k] != XW_FIELD && (ref[k] & XW_SAME_FACING);
            if (!regular && !off[ax][s]) fnb[ax][s] = a.cur[fn + (s ? stride[ax] : -stride[ax])];
        }
    }

    // ---- the node's new value, its filter's new memories; stores -----------------------------------------
    const Real next = boundary_value<Real, 1>(a.courant, a.courant_sq, dirs, nb, off, prev, m, cf);
#pragma unroll
    for (int j = 0; j < 6; ++j) __builtin_nontemporal_store(m[0][j], a.fmem + (size_t)j * a.n_slots + pos_e);
    bad |= bad_bits(next);
    a.next[idx] = next;
    if (LEVEL == 1) {
        a.xw_a[pos_e] = next;
        a.xw_f1[pos_e] = f1;
        a.xw_g[pos_e] = far1;
    } else {
        const Real s2 = faced_value<Real>(fnb, fprev);
        bad |= bad_bits(s2);
        a.next[fn] = s2;
        a.xw_b[pos_e] = next;
        a.xw_f[pos_e] = s2;
    }
}

// The same walls in a THREE-step pass (engine_triple.hip.h): three levels on the compact copies, the generations rotating through the
// arrays so that no lane overwrites what another lane of the same launch still reads.  With t the pass's `current`:
//   own value    xw_a = t-1, xw_b = t   --L1-->  xw_o2 = t+1   --L2-->  xw_a = t+2   --L3-->  xw_b = t+3        (the roles the next pass starts from)
//   faced node   xw_f = t                --L1-->  xw_f1 = t+1 (captured from the march's t+1 field, with xw_g = the node behind it)
//                                        --L2-->  xw_f2 = t+2 (computed here, as in a two-step pass)   --L3-->  xw_f = t+3 (computed here)
//   level 1 (t-1, t -> t+1)    reads one line of the t+1 field (the march's shell values), writes its own t+1 into it
//   level 2 (t, t+1 -> t+2)    reads no field; writes its own and the faced node's t+2 (one line)
//   level 3 (t+1, t+2 -> t+3)  reads no field; writes its own, the faced node's AND the t+3 of the node behind that one (one line) -- none
//                              of these nodes is on the third level's list, where each node two columns from an x-facing wall cost six lines
//                              (1.56 GB per launch of that list at 1024^3 for 0.3 GB of values)
// The node behind the faced one (xw_gok: a plain node whose six neighbours are plain, the one behind it neither a boundary node nor finished
// by one -- its t+2 is the march's or the second level's list's, final before level 2 runs): level 2 captures its t+2 and that of the node
// behind it from their line of the t+2 field (xw_g2 / xw_h2); level 3 gives it its update from the faced node's fresh t+2 (xw_f2), xw_h2,
// the lateral neighbours' xw_g2 and its own t+1 (xw_g).  Where xw_gok is 0 level 3 reads that node from the t+2 field as before and the
// list keeps it.
// a.prev / a.cur / a.next are the fields at the levels boundary_node would read and write; rim entries fall back to them as in xwall_node.
template <typename Real, int LEVEL>
__device__ __forceinline__ void xwall3_node(const BoundaryArgs<Real>& a, const double* coeffs, uint32_t pos_e, int& bad) {
    const uint32_t idx = a.bnode[pos_e];
    const uint32_t dirs = a.btype[pos_e];  // 1: inner node at x-1, 2: at x+1
    const int x = (int)(idx % (uint32_t)a.pitch);
    const uint32_t q = idx / (uint32_t)a.pitch;
    const int y = (int)(q % (uint32_t)a.ny);
    const int z = (int)(q / (uint32_t)a.ny);
    const int64_t plane = (int64_t)a.pitch * a.ny;
    const int64_t stride[3] = {1, a.pitch, plane};
    const int pos[3] = {x, y, z};
    const int lim[3] = {a.nx, a.ny, a.nz};
    const int step = (dirs & 2u) ? 1 : -1;
    const int64_t fn = (int64_t)idx + step;  // the faced node: in the grid (xwall_eligible_kernel)
    const bool far_off = x + 2 * step < 0 || x + 2 * step >= a.nx;
    // which array holds what at this level
    const Real* const own_nb = LEVEL == 1 ? a.xw_b : (LEVEL == 2 ? a.xw_o2 : a.xw_a);    // wall values at the level the neighbours are read at
    const Real* const own_prev = LEVEL == 1 ? a.xw_a : (LEVEL == 2 ? a.xw_b : a.xw_o2);  // ... one level back: this node's own old value
    Real* const own_out = LEVEL == 1 ? a.xw_o2 : (LEVEL == 2 ? a.xw_a : a.xw_b);
    const Real* const f_nb = LEVEL == 1 ? a.xw_f : (LEVEL == 2 ? a.xw_f1 : a.xw_f2);     // the faced node at the neighbours' level
    const int64_t gn = fn + (far_off ? 0 : step);  // the node behind the faced one (the faced one itself if that is off the grid)
    const bool h_off = far_off || x + 3 * step < 0 || x + 3 * step >= a.nx;
    const bool gok = LEVEL != 1 && a.xw_gok[pos_e] != 0;

    // ---- loads -----------------------------------------------------------------------------------
    uint32_t ref[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ref[k] = a.xw_nbr[(size_t)k * a.xw_n + pos_e];
    Real g2c = 0, h2c = 0;  // level 2: t+2 of that node and of the one behind it (a.next: nobody writes them in this launch where gok)
    if (LEVEL == 2) {
        g2c = a.next[gn];
        h2c = a.next[gn + (h_off ? 0 : step)];
    }
    Real nb[3][2];
    bool off[3][2];
    off[0][0] = x - 1 < 0;
    off[0][1] = x + 1 >= a.nx;
    nb[0][0] = nb[0][1] = f_nb[pos_e];  // (only the inner side enters the sums)
    bool any_field = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ax = 1 + (k >> 1), s = k & 1;
        const int c = pos[ax] + (s ? 1 : -1);
        off[ax][s] = c < 0 || c >= lim[ax];
        const bool mirrored = ref[k] != XW_FIELD;
        nb[ax][s] = own_nb[mirrored ? (ref[k] & ~XW_SAME_FACING) : pos_e];
        any_field = any_field || !mirrored;
    }
    const Real prev = own_prev[pos_e];
    double m[1][6];
    const double* cf[1];
    cf[0] = coeffs + (size_t)a.cidx[pos_e] * 14;  // (1-D entries: slot = position)
#pragma unroll
    for (int j = 0; j < 6; ++j) m[0][j] = __builtin_nontemporal_load(a.fmem + (size_t)j * a.n_slots + pos_e);
    // level 1: what the later levels will want of the t+1 field (a.next), both in this node's line
    Real f1 = 0, far1 = 0;
    // levels 2, 3: the faced node's update from the level the neighbours are read at
    Real fnb[3][2], fprev = 0;
    bool any_lateral_field = false;
    if (LEVEL == 1) {
        f1 = a.next[fn];
        far1 = a.next[fn + (far_off ? 0 : step)];
        far1 = far_off ? Real(0) : far1;
    } else {
        const Real own_at = own_nb[pos_e];  // this node's own value at the neighbours' level
        // (level 3: from level 2's capture where there is one -- no field line --, else from the t+2 field)
        Real far = LEVEL == 2 ? a.xw_g[pos_e] : *(gok ? a.xw_g2 + pos_e : a.cur + gn);
        far = (LEVEL == 3 && far_off) ? Real(0) : far;
        fnb[0][0] = step > 0 ? own_at : far;
        fnb[0][1] = step > 0 ? far : own_at;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ax = 1 + (k >> 1), s = k & 1;
            const bool regular = ref[k] != XW_FIELD && (ref[k] & XW_SAME_FACING);
            const Real v = f_nb[regular ? (ref[k] & ~XW_SAME_FACING) : pos_e];
            fnb[ax][s] = off[ax][s] ? Real(0) : v;  // (the faced node's y, z are this node's)
            any_lateral_field = any_lateral_field || (!regular && !off[ax][s]);
        }
        fprev = LEVEL == 2 ? a.xw_f[pos_e] : a.xw_f1[pos_e];
    }
    // level 3: the node behind the faced one, from copies alone where its lateral neighbours are such nodes of neighbouring entries
    Real gnb[3][2], gprev = 0;
    uint32_t g_from_field = 0;
    if (LEVEL == 3) {
        const Real f2 = nb[0][0], h2 = a.xw_h2[pos_e];
        gnb[0][0] = step > 0 ? f2 : h2;
        gnb[0][1] = step > 0 ? h2 : f2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ax = 1 + (k >> 1), s = k & 1;
            const bool regular = ref[k] != XW_FIELD && (ref[k] & XW_SAME_FACING);
            const uint32_t other = regular ? (ref[k] & ~XW_SAME_FACING) : pos_e;
            const bool copy = regular && a.xw_gok[other] != 0;
            const Real v = a.xw_g2[other];
            gnb[ax][s] = off[ax][s] ? Real(0) : v;
            if (!copy && !off[ax][s]) g_from_field |= 1u << k;
        }
        gprev = a.xw_g[pos_e];
        if (gok && g_from_field) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ax = 1 + (k >> 1), s = k & 1;
                if ((g_from_field >> k) & 1u) gnb[ax][s] = a.cur[gn + (s ? stride[ax] : -stride[ax])];
            }
        }
    }
    // the wall's rim, walls that meet other things than walls: those neighbours come from the fields
    if (any_field) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ax = 1 + (k >> 1), s = k & 1;
            if (ref[k] == XW_FIELD) nb[ax][s] = a.cur[(int64_t)idx + (off[ax][s] ? 0 : (s ? stride[ax] : -stride[ax]))];
        }
    }
    if (LEVEL != 1 && any_lateral_field) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ax = 1 + (k >> 1), s = k & 1;
            const bool regular = ref[k] != XW_FIELD && (ref[k] & XW_SAME_FACING);
            if (!regular && !off[ax][s]) fnb[ax][s] = a.cur[fn + (s ? stride[ax] : -stride[ax])];
        }
    }

    // ---- the node's new value, its filter's new memories; stores -----------------------------------------
    const Real next = boundary_value<Real, 1>(a.courant, a.courant_sq, dirs, nb, off, prev, m, cf);
#pragma unroll
    for (int j = 0; j < 6; ++j) __builtin_nontemporal_store(m[0][j], a.fmem + (size_t)j * a.n_slots + pos_e);
    bad |= bad_bits(next);
    a.next[idx] = next;
    own_out[pos_e] = next;
    if (LEVEL == 1) {
        a.xw_f1[pos_e] = f1;
        a.xw_g[pos_e] = far1;
    } else {
        const Real sf = faced_value<Real>(fnb, fprev);
        bad |= bad_bits(sf);
        a.next[fn] = sf;
        (LEVEL == 2 ? a.xw_f2 : a.xw_f)[pos_e] = sf;
        if (LEVEL == 2) {
            a.xw_g2[pos_e] = g2c;
            a.xw_h2[pos_e] = h_off ? Real(0) : h2c;
        } else {
            const Real sg = faced_value<Real>(gnb, gprev);
            if (gok) {
                bad |= bad_bits(sg);
                a.next[gn] = sg;
            }
        }
    }
}

// entry id -> dimensionality dispatch (entry order: all 1-D, all 2-D, all 3-D)
template <typename Real, bool FIX>
__device__ __forceinline__ void boundary_entry(const BoundaryArgs<Real>& a, const double* coeffs, uint32_t e, int& bad) {
    if (e < a.n1) {
        boundary_node<Real, 1, FIX>(a, coeffs, e, e, 0u, a.n1, bad);
    } else if (e < a.n1 + a.n2) {
        boundary_node<Real, 2, FIX>(a, coeffs, e - a.n1, e, a.n1, a.n2, bad);
    } else if (e < a.n1 + a.n2 + a.n3) {
        boundary_node<Real, 3, FIX>(a, coeffs, e - a.n1 - a.n2, e, a.n1 + 2u * a.n2, a.n3, bad);
    }
}

// LDSC: the coefficient table (14 doubles per surface) is staged in LDS by the workgroup first --
// a lane reads 14 coefficients per filter and step, which are 14 requests into L1/L2 when they
// come from global memory and none when they come from LDS (scenes have a handful of materials).
constexpr uint32_t kMaxLdsCoefficientSets = 256;  // 28 KiB

template <typename Real>
__device__ __forceinline__ void pre_post_body(const PrePostArgs<Real>& a, uint32_t t, uint32_t width);  // below

// `next`: when next.fused is set the last workgroup also does the NEXT step's source injection /
// receiver gather (on `prev`, which is that step's `current`).  Only legal when none of those nodes
// is a boundary node -- then they were final when the sweep before this launch ended (engine.hip).
// FIX: the second launch of a two-step pass, where 1-D entries also finish the inside node they face
// (BoundaryArgs::fix_z0 / fix_z1).
// (`block` of `blocks`: this workgroup's place among the boundary workgroups of the launch -- all of it for boundary_kernel, the
// tail of the grid for plane_step_kernel.)
// XW3: 0, or the level (1, 2, 3) of a three-step pass whose x-facing walls work on their compact copies (xwall3_node)
template <typename Real, bool LDSC, bool FIX, int XW3 = 0>
__device__ __forceinline__ void boundary_body(const BoundaryArgs<Real>& a, const PrePostArgs<Real>& next, uint32_t block, uint32_t blocks) {
    __shared__ double s_coeffs[LDSC ? kMaxLdsCoefficientSets * 14 : 1];
    if (LDSC) {
        for (uint32_t w = threadIdx.x; w < a.n_coeffs * 14u; w += 256) s_coeffs[w] = a.coeffs[w];
        __syncthreads();
    }
    const double* coeffs = LDSC ? s_coeffs : a.coeffs;
    const uint32_t t = block * blockDim.x + threadIdx.x;
    int bad = 0;
    if (a.xw_n) {
        // two-step pass: the first xw_n entries (x-facing walls) by their compact copies, whole workgroups of them
        // first; then everything else as ever (FIX = this is the pass's second launch = level 2)
        if (t < a.xw_pad) {
            if (t < a.xw_n) {
                if (XW3)
                    xwall3_node<Real, XW3 ? XW3 : 1>(a, coeffs, t, bad);
                else
                    xwall_node<Real, FIX ? 2 : 1>(a, coeffs, t, bad);
            }
        } else if (a.order) {
            if (t - a.xw_pad < a.n_order) boundary_entry<Real, FIX>(a, coeffs, a.order[t - a.xw_pad], bad);
        } else {
            boundary_entry<Real, FIX>(a, coeffs, a.xw_n + (t - a.xw_pad), bad);
        }
    } else if (a.order) {
        if (t < a.n_order) boundary_entry<Real, FIX>(a, coeffs, a.order[t], bad);
    } else {
        boundary_entry<Real, FIX>(a, coeffs, t, bad);
    }
    if (bad) atomicOr(a.flag, bad);
    if (next.fused && block == blocks - 1) pre_post_body<Real>(next, threadIdx.x, 256);
}

template <typename Real, bool LDSC, bool FIX = false, int XW3 = 0>
__global__ void __launch_bounds__(256) boundary_kernel(const BoundaryArgs<Real> a, const PrePostArgs<Real> next) {
    if (next.fused) {
        // the source / receiver work that rides here has a workgroup of its own, the grid's FIRST (launch_boundary adds it): a chain of
        // some eight dependent memory round trips -- sample, node, fence, the short list's neighbours -- that starts with the launch and
        // runs beside the boundary workgroups, instead of behind the last of them (10 us of a 25 us launch at 256^3)
        if (blockIdx.x == 0) {
            pre_post_body<Real>(next, threadIdx.x, 256);
            return;
        }
        PrePostArgs<Real> none{};
        boundary_body<Real, LDSC, FIX, XW3>(a, none, blockIdx.x - 1, gridDim.x - 1);
        return;
    }
    boundary_body<Real, LDSC, FIX, XW3>(a, next, blockIdx.x, gridDim.x);
}

// the inside nodes the x-facing walls' entries finish at the third level of a three-step pass (xwall3_node): a bit per stored node, for the
// third-level list to leave them out (triple_map_kernel)
struct XwCoverArgs {
    const uint32_t* bnode;
    const uint8_t* btype;
    uint32_t* covered;  // bitmap over stored node indices
    uint32_t xw_n;
    // the node behind the faced one: xwall3_node's xw_gok (see there) -- and covered too where that says yes
    const uint8_t* pair_map;
    uint8_t* gok;
    int nx, ny, nz, pitch, cls_pitch;
};
__global__ void __launch_bounds__(256) xwall_cover_kernel(const XwCoverArgs a) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.xw_n) return;
    const uint32_t idx = a.bnode[p];
    const int step = (a.btype[p] & 2u) ? 1 : -1;
    const uint32_t fn = idx + (uint32_t)step;
    atomicOr(a.covered + (fn >> 5), 1u << (fn & 31u));
    const int x = (int)(idx % (uint32_t)a.pitch);
    const uint32_t q = idx / (uint32_t)a.pitch;
    const int y = (int)(q % (uint32_t)a.ny), z = (int)(q / (uint32_t)a.ny);
    auto code_at = [&](int xx) -> uint32_t { return (a.pair_map[cls_byte_index(xx, y, z, a.ny, a.cls_pitch)] >> ((xx & 3) * 2)) & 3u; };
    const int xg = x + 2 * step, xh = x + 3 * step;
    bool ok = xg >= 0 && xg < a.nx && code_at(xg) == 1u;  // plain, and so are its six neighbours: its t+2 is the march's
    if (ok && xh >= 0 && xh < a.nx) ok = code_at(xh) != 2u && code_at(xh) != 3u;  // (3: its t+2 may be a boundary entry's, in level 2's own launch)
    a.gok[p] = ok ? 1 : 0;
    if (ok) {
        const uint32_t gn = idx + (uint32_t)(2 * step);
        atomicOr(a.covered + (gn >> 5), 1u << (gn & 31u));
    }
}

// Which 1-D entries may live on compact copies (xwall_node): facing along x, in the planes the march produces, the
// faced node an inside node of the grid, and the node behind it not a boundary node (level 1 reads its t+1 value
// while other lanes of the same launch are still writing theirs).  One thread per 1-D entry, before the engine
// settles the entries' processing order (engine_setup.hip.h).
struct XwEligibleArgs {
    const uint32_t* bnode;
    const uint8_t* btype;
    const uint8_t* cls;
    uint8_t* eligible;  // [n1]
    uint32_t n1;
    int nx, ny, nz, pitch, cls_pitch;
    int march_begin, march_end;
};

__global__ void __launch_bounds__(256) xwall_eligible_kernel(const XwEligibleArgs a) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n1) return;
    const uint32_t idx = a.bnode[e];
    const uint32_t dirs = a.btype[e] & 0x3Fu;
    bool ok = idx != INVALID_NODE && (dirs == 1u || dirs == 2u);
    if (ok) {
        const int x = (int)(idx % (uint32_t)a.pitch);
        const uint32_t q = idx / (uint32_t)a.pitch;
        const int y = (int)(q % (uint32_t)a.ny), z = (int)(q / (uint32_t)a.ny);
        const int step = dirs == 2u ? 1 : -1;
        auto cls_at = [&](int xx) -> uint32_t {
            return (a.cls[cls_byte_index(xx, y, z, a.ny, a.cls_pitch)] >> ((xx & 3) * 2)) & 3u;
        };
        ok = z >= a.march_begin && z < a.march_end && x + step >= 0 && x + step < a.nx && cls_at(x + step) == CLS_INSIDE;
        if (ok && x + 2 * step >= 0 && x + 2 * step < a.nx) ok = cls_at(x + 2 * step) != CLS_BOUNDARY;
    }
    a.eligible[e] = ok ? 1 : 0;
}

// (Re)fill the compact copies from the fields: a.prev / a.cur = fields t-1 / t.  After anything but a two-step pass has
// touched the fields (a caller's write, single steps), before the next pass.
template <typename Real>
__global__ void __launch_bounds__(256) xwall_gather_kernel(const BoundaryArgs<Real> a) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.xw_n) return;
    const uint32_t idx = a.bnode[p];
    const int64_t fn = (int64_t)idx + ((a.btype[p] & 2u) ? 1 : -1);
    a.xw_a[p] = a.prev[idx];
    a.xw_b[p] = a.cur[idx];
    a.xw_f[p] = a.cur[fn];
}

// ---- source injection + receiver gather: the pre/post callbacks, device resident -------------
// hard source:  current[node] = sample              (preprocessor/hard_source.h:21)
// soft source:  current[node] = current[node] + s   (preprocessor/soft_source.h:21-24)
// receivers read the same `current` afterwards      (waveguide.h:121)
// every thread of the workgroup calls this; `width` threads share the receivers
template <typename Real>
__device__ __forceinline__ void pre_post_body(const PrePostArgs<Real>& a, uint32_t t, uint32_t width) {
    if (t == 0 && a.flag) *a.flag = a.flag_init;  // waveguide.h:82 (write_value(error_flag, id_success)) + static bits
    if (t == 0 && a.flag2) *a.flag2 = a.flag_init;
    Real injected = 0;
    const bool has_source = a.source_kind != 0;
    if (has_source) {
        const Real s = (Real)a.signal[a.signal_pos + (a.signal_base ? *a.signal_base : 0ull)];
        injected = (a.source_kind == 1) ? s : (Real)(a.cur[a.source_node] + s);
    }
    for (uint32_t r = t; r < a.n_recv; r += width) {
        const uint64_t node = a.recv[r];
        Real v = 0;
        if (node != ~0ull) v = (has_source && node == a.source_node) ? injected : a.cur[node];
        a.recv_out[r] = v;
    }
    __syncthreads();
    if (t == 0 && has_source) a.cur[a.source_node] = injected;
    if (a.fix_n) {
        // (the sample above was stored by another lane, possibly another wave, of this workgroup: fence,
        // barrier, and loads that do not stop at this CU's L1)
        __threadfence();
        __syncthreads();
        const int64_t plane = (int64_t)a.pitch * a.ny;
        int bad = 0;
        for (uint32_t i = t; i < a.fix_n; i += width) {
            const uint32_t idx = a.fix_nodes[i];
            const int x = (int)(idx % (uint32_t)a.pitch);
            const uint32_t q = idx / (uint32_t)a.pitch;
            const int y = (int)(q % (uint32_t)a.ny);
            const int z = (int)(q / (uint32_t)a.ny);
            auto t1 = [&](int64_t at) { return __hip_atomic_load(a.cur + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            Real s = 0;
            s += (x > 0) ? t1((int64_t)idx - 1) : Real(0);
            s += (x + 1 < a.nx) ? t1((int64_t)idx + 1) : Real(0);
            s += (y > 0) ? t1((int64_t)idx - a.pitch) : Real(0);
            s += (y + 1 < a.ny) ? t1((int64_t)idx + a.pitch) : Real(0);
            s += (z > 0) ? t1((int64_t)idx - plane) : Real(0);
            s += (z + 1 < a.nz) ? t1((int64_t)idx + plane) : Real(0);
            s = div3(s);
            s -= a.fix_cur[idx];
            bad |= bad_bits(s);
            a.fix_out2[idx] = s;
        }
        if (bad) atomicOr(a.fix_flag, bad);
    }
}

template <typename Real>
__global__ void __launch_bounds__(64) pre_post_kernel(const PrePostArgs<Real> a) {
    pre_post_body<Real>(a, threadIdx.x, 64);
}

// ---- set-up (once per wv_create) ---------------------------------------------------------------
struct NodeRec {                 // condensed_node, cl/structs.h:19-22
    int32_t boundary_type;
    uint32_t boundary_index;
};

struct SetupArgs {
    const NodeRec* nodes;        // staged chunk (compact, nx per row): rows [first_row, first_row+rows)
    int64_t first_row, rows;
    int nx, ny;
    int pitch;                   // stored row length of the fields
    int cls_pitch;
    uint8_t* cls;
    uint32_t* bnode;
    uint8_t* btype;
    uint32_t n1, n2, n3;
    int* status;                 // bit0: invalid boundary type, bit1: boundary_index out of range
    int z_begin, z_end;          // planes this engine updates (ghost planes excluded)
};

__device__ __forceinline__ uint32_t classify(int32_t t, int* dim_out) {
    *dim_out = 0;
    if (t == 0) return CLS_NONE;
    const int bits = __popc((uint32_t)t);
    if (bits == 1 && (t & 1)) return CLS_INSIDE;
    if (bits == 1 && (t & 128)) return CLS_REENTRANT;
    *dim_out = bits;
    return CLS_BOUNDARY;
}

// one thread per class byte (4 nodes of one row)
__global__ void __launch_bounds__(256) setup_classify_kernel(const SetupArgs a) {
    const int64_t n_bytes = a.rows * a.cls_pitch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / a.cls_pitch;
        const int xb = (int)(i % a.cls_pitch);
        const int64_t grow = a.first_row + row;      // global row = z*ny + y
        const int z = (int)(grow / a.ny);
        uint32_t byte = 0;
        for (int j = 0; j < 4; ++j) {
            const int x = xb * 4 + j;
            if (x >= a.nx) break;                    // pad columns stay class "none"
            const NodeRec rec = a.nodes[row * a.nx + x];
            const int32_t t = rec.boundary_type;
            int dim;
            const uint32_t c = classify(t, &dim);
            byte |= c << (2 * j);
            if (c == CLS_BOUNDARY) {
                // valid boundary type: only direction bits, at most one per axis, 1..3 of them
                const uint32_t dirs = ((uint32_t)t >> 1) & 0x3Fu;
                const bool only_dirs = ((uint32_t)t & ~0x7Eu) == 0;
                const bool axes_ok = ((dirs & 3u) != 3u) && (((dirs >> 2) & 3u) != 3u) && (((dirs >> 4) & 3u) != 3u);
                if (!only_dirs || !axes_ok || dim < 1 || dim > 3) {
                    atomicOr(a.status, 1);
                    continue;
                }
                // ghost-plane boundary nodes belong to the neighbouring slab: classified (their
                // type matters to the static checks) but never entered in this engine's lists
                if (z < a.z_begin || z >= a.z_end) continue;
                const uint32_t n_d = dim == 1 ? a.n1 : (dim == 2 ? a.n2 : a.n3);
                const uint32_t off = dim == 1 ? 0u : (dim == 2 ? a.n1 : a.n1 + a.n2);
                const uint32_t k = rec.boundary_index;
                if (k >= n_d) {
                    atomicOr(a.status, 2);
                    continue;
                }
                a.bnode[off + k] = (uint32_t)(grow * a.pitch + x);
                a.btype[off + k] = (uint8_t)dirs;
            }
        }
        a.cls[cls_byte_index(xb * 4, (int)(grow % a.ny), z, a.ny, a.cls_pitch)] = (uint8_t)byte;
    }
}

// Static part of the reference's per-step checks (program.cpp:198-205, :245): an in-plane
// neighbour of a boundary node that is id_none / id_inside -> suspicious boundary; a neighbour
// off the grid -> outside mesh.  Mesh-invariant, so evaluated once and OR-ed into every step.
struct ValidateArgs {
    const uint32_t* bnode;
    const uint8_t* btype;
    const uint8_t* cls;
    uint32_t n_entries;
    int nx, ny, nz, pitch, cls_pitch;
    int* static_flag;
};

__global__ void __launch_bounds__(256) setup_validate_kernel(const ValidateArgs a) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_entries) return;
    const uint32_t idx = a.bnode[e];
    if (idx == INVALID_NODE) return;
    const uint32_t dirs = a.btype[e] & 0x3Fu;
    const int pos[3] = {(int)(idx % (uint32_t)a.pitch), (int)((idx / (uint32_t)a.pitch) % (uint32_t)a.ny),
                        (int)(idx / ((uint32_t)a.pitch * (uint32_t)a.ny))};
    const int lim[3] = {a.nx, a.ny, a.nz};
    int flag = 0;
    int dim = __popc(dirs);
    for (int ax = 0; ax < 3; ++ax) {
        const bool has_n = (dirs >> (2 * ax)) & 1u, has_p = (dirs >> (2 * ax + 1)) & 1u;
        if (has_n || has_p) {
            const int c = pos[ax] + (has_p ? 1 : -1);
            if (c < 0 || c >= lim[ax]) flag |= FLAG_OUTSIDE_MESH;
        } else if (dim < 3) {
            for (int s = -1; s <= 1; s += 2) {
                int p[3] = {pos[0], pos[1], pos[2]};
                p[ax] += s;
                if (p[ax] < 0 || p[ax] >= lim[ax]) {
                    flag |= FLAG_OUTSIDE_MESH;
                    break;  // the reference returns from the surrounding sum here
                }
                const uint8_t byte = a.cls[cls_byte_index(p[0], p[1], p[2], a.ny, a.cls_pitch)];
                const uint32_t c = (byte >> ((p[0] & 3) * 2)) & 3u;
                if (c == CLS_NONE || c == CLS_INSIDE) flag |= FLAG_SUSPICIOUS;
            }
            if (flag & FLAG_OUTSIDE_MESH) break;
        }
    }
    if (flag) atomicOr(a.static_flag, flag);
}

// Which workgroup tiles of the sweep hold at least one node the sweep updates (class inside or
// re-entrant): one thread per tile of `tile_rows` x `tile_cols` nodes of one plane.
struct TileActivityArgs {
    const uint8_t* cls;
    uint8_t* active;  // [nz][tiles_y][tiles_x]
    int ny, nz, pitch, cls_pitch;
    int tile_rows, tile_cols, tiles_x, tiles_y;
};

__global__ void __launch_bounds__(256) tile_activity_kernel(const TileActivityArgs a) {
    const int64_t n = (int64_t)a.nz * a.tiles_y * a.tiles_x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int tx = (int)(t % a.tiles_x);
    const int ty = (int)((t / a.tiles_x) % a.tiles_y);
    const int z = (int)(t / ((int64_t)a.tiles_x * a.tiles_y));
    uint32_t any = 0;
    for (int y = ty * a.tile_rows; y < min((ty + 1) * a.tile_rows, a.ny); ++y)
        for (int x = tx * a.tile_cols; x < min((tx + 1) * a.tile_cols, a.pitch); x += 4)
            any |= a.cls[cls_byte_index(x, y, z, a.ny, a.cls_pitch)] & 0x55u;  // bit 0 of a class: inside / re-entrant
    a.active[t] = any ? 1 : 0;
}

// ---- field I/O for wv_read_field / wv_write_field: compact host layout (nx per row) <-> stored
// layout (pitch per row), with f32 <-> f64 conversion when the element types differ
template <typename Dst, typename Src>
__global__ void __launch_bounds__(256) pack_rows_kernel(Dst* dst, int dst_pitch, const Src* src, int src_pitch, int nx,
                                                        int64_t rows) {
    const int64_t n = rows * nx;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / nx;
        const int x = (int)(i % nx);
        dst[row * dst_pitch + x] = (Dst)src[row * src_pitch + x];
    }
}

// AoS <-> SoA transposition of the boundary filter state (cl/structs.h:38-58)
struct BoundaryDataArgs {
    double* fmem;
    uint32_t* cidx;
    uint32_t n_slots, slot_base, n_d;
    int dim;
    uint64_t* aos;  // boundary_data_array<dim>[n_d] as 7 x 8-byte words per filter
    uint32_t entry_off;          // first entry of this dimensionality class
    const uint32_t* ref_to_pos;  // [n_entries] caller's boundary_index (+ entry_off) -> processing position
};

__global__ void __launch_bounds__(256) boundary_data_scatter_kernel(const BoundaryDataArgs a, int to_device) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_d * (uint32_t)a.dim) return;
    const uint32_t k_ref = t / (uint32_t)a.dim, i = t % (uint32_t)a.dim;
    const uint32_t k = a.ref_to_pos[a.entry_off + k_ref] - a.entry_off;
    const uint32_t slot = a.slot_base + i * a.n_d + k;
    uint64_t* rec = a.aos + (size_t)t * 7;
    if (to_device) {
        for (int j = 0; j < 6; ++j) a.fmem[(size_t)j * a.n_slots + slot] = __longlong_as_double((long long)rec[j]);
        a.cidx[slot] = (uint32_t)(rec[6] & 0xFFFFFFFFull);
    } else {
        for (int j = 0; j < 6; ++j) rec[j] = (uint64_t)__double_as_longlong(a.fmem[(size_t)j * a.n_slots + slot]);
        rec[6] = (uint64_t)a.cidx[slot];
    }
}

}  // namespace wv
