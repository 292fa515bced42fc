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

}  // namespace wv
