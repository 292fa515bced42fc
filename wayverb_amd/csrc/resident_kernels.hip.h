// resident_kernels.hip.h -- a whole batch of single steps of a SMALL mesh in ONE launch: persistent workgroups on ONE XCD, each step's
// work cut into units that wait for the units around them only.
//
// Below about 160^3 a step is not bound by bytes but by launches: two dependent kernels of ~6 us each per step (sweep, boundary
// nodes), however little they move (32^3: 11 us per step for 0.5 MB of field).  Rounds 2-4 priced the obvious ways
// out and all lost: a persistent kernel with a GRID-WIDE barrier per step (27 us for 256 workgroups: the arrivals serialise on one
// counter), every sweep workgroup finishing the boundary nodes of its tile (a boundary node triples the life of its workgroup),
// a second stream (22 us per fork / join), hipGraph replays (+6 %: the kernels' own dispatch latency stays).
//
// Two things make this form work where those did not (both measured first: tools/neighbour_sync_bench.hip, tools/xcd_sync_bench.hip,
// profiles/r05/):
//  * no barrier across the grid, because the stencil needs none: the step of a piece of the mesh needs the pieces AROUND it one step
//    back, nothing else.  A step's work is cut into UNITS -- the workgroup tiles of the plane sweep (stream_sweep_body with masked
//    stores: inside / outside nodes) and the 256-entry blocks of the boundary list (boundary_entries) -- the very device code the
//    per-step launches run, so the arithmetic cannot differ by a bit.  Unit u of step s waits until every unit that writes a node u
//    reads, or reads a node u writes, has finished step s - 1 (a counter per unit; the lists are made once per mesh on the host:
//    engine_resident.hip.h), does its work, waits for its stores to be acknowledged and publishes its counter;
//  * every workgroup that takes part sits on ONE XCD.  Between XCDs a hand-over of data costs the consumer an invalidate of its
//    L2 (buffer_inv sc1), which serialises per XCD at 0.26 us per workgroup -- 10-24 us per step with a few hundred workgroups,
//    more than the launches it was to save (and no kind of device memory, uncached or fine-grained, does without it).  The 32 CUs
//    of one XCD share their L2: a store the L2 has acknowledged is what the next load of any of them sees, and a hand-over costs
//    1.4 us including the data.  Workgroups are dealt to the XCDs round-robin, so the launch has 8 K workgroups of which the K that
//    find themselves on XCD 0 (hardware register XCC_ID) stay; they count themselves in and start when all K are there (fewer:
//    the launch gives up at once and the engine goes back to per-step launches for good).  A mesh of up to ~64^3 lives in that
//    XCD's 4 MB of L2 for the whole batch.
// Workgroup w of the K takes the units w, w + K, w + 2 K ... of every step, in that order, which makes the scheme deadlock-free:
// the earliest unfinished (step, unit) never waits for a later one.
//
// Source and receivers (waveguide.h:80-123: `pre` injects into `current`, `post` observes it) ride with the units that own their
// nodes: after unit u has stored the values of level s + 1 it puts the sample of step s + 1 into the source node if that is its
// node, and records the receivers it owns for step s + 1, before it publishes.  Step 0 of a batch is served by the usual
// pre_post_kernel in front of the launch.
#pragma once
#include "boundary_kernels.hip.h"
#include "stream_kernels.hip.h"

namespace wv {

struct ResidentIo {  // one source / receiver duty of a unit
    uint64_t node;   // stored index
    uint32_t col;    // receiver: column of the step's row; source: unused
    uint32_t kind;   // 0 receiver, 1 hard source, 2 soft source
};

template <typename Real>
struct ResidentArgs {
    StreamArgs<Real> s;       // the full sweep over the owned planes (arithmetic tile mapping), fields filled in per step
    BoundaryArgs<Real> b;     // every boundary entry in list order, fields filled in per step
    Real* field[2];           // [0]: `current` of step 0, [1]: `previous` of step 0 (the steps alternate, in place)
    int* flags;               // [steps] error_code words, already reset to the mesh-static bits
    uint32_t steps;
    uint32_t n_sweep, n_units;         // units 0 .. n_sweep-1: sweep tiles; the rest: boundary blocks
    const uint32_t* sweep_block;       // [n_sweep] block index of the sweep launch each sweep unit stands for
    uint32_t boundary_blocks;          // = n_units - n_sweep
    const uint32_t* dep_start;         // [n_units + 1]
    const uint32_t* dep;               // units each unit waits for
    uint32_t* counter;                 // [n_units] steps finished, counted from `base`
    uint32_t base;
    uint32_t workgroups;               // K: how many workgroups take part (the launch has 8 K)
    uint32_t* arrived;                 // zeroed before the launch: the participants count themselves in
    const uint32_t* io_start;          // [n_units + 1]
    const ResidentIo* io;
    const double* signal;
    uint64_t signal_pos;               // sample of step 0 of this launch
    Real* recv_out;                    // [steps][n_recv]
    uint32_t n_recv;
    int* gave_up;                      // set when a wait ran into its bound (a bug, not a state: the launch then ends anyhow)
};

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xFu;
}

// How many workgroups of a launch of this size land on XCD 0?  (Once per engine, before the form is first taken: the K the real
// launches wait for must be what the dispatcher delivers.)
__global__ void __launch_bounds__(256) resident_probe_kernel(uint32_t* arrived) {
    if (xcc_id() == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The argument block lives in device memory and is read where it is needed (the pointer is made opaque once per unit): held in
// registers across the whole loop nest, two launches' worth of wave-uniform arguments leave the compiler nowhere but scratch for them.
template <typename Real, bool LDSC>
__global__ void __launch_bounds__(256) resident_kernel(const ResidentArgs<Real>* __restrict__ rp) {
    const uint32_t t = threadIdx.x;
    if (xcc_id() != 0) return;  // (seven of eight workgroups: the launch exists to put K of them on one XCD)
    __shared__ uint32_t s_rank;
    __shared__ int s_abort;
    const uint32_t K = rp->workgroups;
    if (t == 0) {
        s_abort = 0;
        s_rank = __hip_atomic_fetch_add(rp->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // everybody has to be there before anybody waits for a unit somebody else is to run
        uint32_t spins = 0;
        while (__hip_atomic_load(rp->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < K) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 16) || __hip_atomic_load(rp->gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(rp->gave_up, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (2: not a single step was taken)
                s_abort = 2;
                break;
            }
        }
    }
    __syncthreads();
    const uint32_t w = s_rank;
    if (s_abort == 2 || w >= K) return;
    const uint32_t steps = rp->steps, n_units = rp->n_units, n_sweep = rp->n_sweep, base = rp->base;
    for (uint32_t step = 0; step < steps; ++step) {
        const uint32_t need = base + step;  // every unit around has finished the step before this one
        for (uint32_t u = w; u < n_units; u += K) {
            const ResidentArgs<Real>* r = rp;
            asm volatile("" : "+s"(r));  // (opaque: what is read through it below is read now, not kept from an earlier unit)
            // ---- wait for the units around (one lane per unit waited for)
            const uint32_t d0 = r->dep_start[u], d1 = r->dep_start[u + 1];
            {
                const uint32_t* counter = r->counter;
                const uint32_t* dep = r->dep;
                int* gave_up = r->gave_up;
                for (uint32_t d = d0 + t; d < d1 && !s_abort; d += 256) {
                    const uint32_t* c = counter + dep[d];
                    uint32_t spins = 0;
                    while ((int32_t)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) < 0) {
                        __builtin_amdgcn_s_sleep(1);
                        // (a second or two: never in a correct run.  Whoever runs into the bound says so for the whole grid, and every
                        // other wait ends within a thousand polls: a broken launch must not hold the GPU)
                        if (++spins > (1u << 21) || ((spins & 1023u) == 0 && __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                            __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s_abort = 1;
                            break;
                        }
                    }
                }
            }
            __syncthreads();
            asm volatile("buffer_inv sc0" ::: "memory");  // this CU's L1 may hold lines of a step ago; the XCD's L2 is what everybody shares
            Real* cur = (step & 1u) ? r->field[1] : r->field[0];
            Real* nxt = (step & 1u) ? r->field[0] : r->field[1];
            // ---- the unit's share of the step: the per-step launches' own device code
            if (u < n_sweep) {
                StreamArgs<Real> s = r->s;
                s.cur = cur;
                s.prev = nxt;
                s.next = nxt;
                s.flag = r->flags + step;
                stream_sweep_body<Real, 4, 1, 4, X_NO_LIST>(s, r->sweep_block[u]);  // (masked stores; no streaming hints: the L2 is home)
            } else {
                BoundaryArgs<Real> b = r->b;
                b.cur = cur;
                b.prev = nxt;
                b.next = nxt;
                b.flag = r->flags + step;
                boundary_entries<Real, LDSC, false, false>(b, u - n_sweep);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have been acknowledged by the L2
            __syncthreads();                                  // ... and every wave's (and the bodies' LDS is free again)
            // ---- the next step's source sample / receiver samples at the nodes this unit has just finished
            const uint32_t i0 = r->io_start[u], i1 = r->io_start[u + 1];
            if (i0 != i1 && step + 1 < steps && t == 0) {
                asm volatile("buffer_inv sc0" ::: "memory");
                for (uint32_t i = i0; i < i1; ++i) {  // (a unit's source duty comes before its receiver duties)
                    const ResidentIo io = r->io[i];
                    // (the field through agent-scope atomics: a plain load at a wave-uniform address may be a scalar load, whose cache
                    // nobody invalidates here)
                    if (io.kind) {
                        const Real sample = (Real)r->signal[r->signal_pos + step + 1];
                        const Real old = __hip_atomic_load(nxt + io.node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(nxt + io.node, io.kind == 1 ? sample : (Real)(old + sample), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else {
                        r->recv_out[(size_t)(step + 1) * r->n_recv + io.col] = __hip_atomic_load(nxt + io.node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (t == 0) __hip_atomic_store(r->counter + u, need + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace wv
