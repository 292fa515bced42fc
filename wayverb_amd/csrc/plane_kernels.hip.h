// plane_kernels.hip.h -- a few whole planes one time step on, in ONE launch: the sweep's workgroups and the workgroups of those
// planes' boundary entries side by side.
//
// The step of a plane is two independent pieces of work that read the same inputs and write disjoint nodes: the 7-point update
// of its inside / re-entrant nodes (stream_sweep_body, stream_kernels.hip.h) and its boundary nodes (boundary_body,
// boundary_kernels.hip.h; program.cpp:331-387).  The product sweep buys 4 % by storing whole vectors -- a boundary node gets its
// old value written back -- which is what forces "sweep, then boundary kernel" on one stream.  For the two or four planes a
// z-slab steps around its halo exchanges (engine_single.hip.h, launch_faces) the 4 % are nothing and a launch is ~6 us: here the
// sweep runs with masked stores (it leaves boundary nodes alone) and the boundary entries ride in the same grid, behind the
// sweep's workgroups.  Same arithmetic, same bits.
#pragma once
#include "boundary_kernels.hip.h"
#include "stream_kernels.hip.h"

namespace wv {

template <typename Real, bool LDSC>
__global__ void __launch_bounds__(256) plane_step_kernel(const StreamArgs<Real> s, const BoundaryArgs<Real> b, const uint32_t sweep_blocks) {
    if (blockIdx.x < sweep_blocks) {
        stream_sweep_body<Real, 4, 1, 4, (X_SWEEP & ~X_STORE_ALL)>(s, blockIdx.x);
    } else {
        PrePostArgs<Real> none{};  // (fused == 0: nothing rides -- the sweep's half of the step may still be running)
        boundary_body<Real, LDSC, false>(b, none, blockIdx.x - sweep_blocks, gridDim.x - sweep_blocks);
    }
}

// A whole step of a small mesh in ONE launch: the same two halves side by side over every owned plane, and the NEXT step's source
// sample, receiver row and flag word served on the way (StepDuties: by the tile that produces the node's value, from its registers;
// legal when every such node is an inside node -- the engine checks, engine_single.hip.h).  Below about 160^3 a step is two dependent
// launches of 5-6 us each, each of them a chain of dependent loads however little it moves (a 32^3 sweep: 5 us); here the two chains
// run beside each other and the stream sees one launch per step.  Same arithmetic, same bits.
template <typename Real, bool LDSC>
__global__ void __launch_bounds__(256) whole_step_kernel(const StreamArgs<Real> s, const BoundaryArgs<Real> b, const StepDuties<Real> d,
                                                         const uint32_t boundary_blocks) {
    if (blockIdx.x == 0 && threadIdx.x < 64) {  // what nobody else in this launch touches: the next step's flag word, columns of unrecorded receivers
        if (threadIdx.x == 0 && d.next_flag) *d.next_flag = d.flag_init;
        for (uint32_t r = threadIdx.x; r < d.n_recv; r += 64)
            if (d.recv[r] == ~0ull) d.recv_out[r] = Real(0);
    }
    // The boundary entries' workgroups come FIRST in the grid: a boundary node is ~150 dependent instructions behind a chain of loads,
    // the longest-lived workgroups of the step -- dispatched first they run under the sweep instead of after it (measured: 5-13 % of
    // a step between 64^3 and 160^3 against "sweep first").  `boundary_blocks` is a multiple of 8, so that sweep workgroup j still
    // lands on XCD j % 8.
    if (blockIdx.x >= boundary_blocks) {
        stream_sweep_body<Real, 4, 1, 4, ((X_SWEEP & ~X_STORE_ALL) | X_DUTIES)>(s, blockIdx.x - boundary_blocks, &d);
    } else {
        PrePostArgs<Real> none{};
        boundary_body<Real, LDSC, false>(b, none, blockIdx.x, boundary_blocks);
    }
}

}  // namespace wv
