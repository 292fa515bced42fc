/* wayverb_amd.h -- C ABI of the MI355X-native waveguide engine.
 *
 * Drop-in boundary for wayverb's `waveguide::run` hot path.  Every entry point below names
 * the reference interface it replaces (paths relative to the reference repository root).
 * The C++ mirror of `waveguide::run<pre,post>` (include/wayverb_amd/waveguide.h) and the
 * ctypes binding (wayverb_amd/engine.py) sit on top of exactly this ABI.
 *
 * Conventions
 *   - every function returns WV_OK (0) or a negative WV_E_* status; no exception crosses the ABI;
 *   - wv_last_error() returns the message for the calling thread's most recent failure;
 *   - plain pointers + sizes only; host pointers unless a parameter says "device";
 *   - node index = x + y*nx + z*nx*ny  (src/waveguide/src/cl/utils.cpp:33-36);
 *   - an engine handle is used from one thread at a time (as `run` is,
 *     src/waveguide/include/waveguide/waveguide.h:36-41).
 */
#ifndef WAYVERB_AMD_H
#define WAYVERB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------ */
enum {
    WV_OK = 0,
    WV_E_INVALID_ARGUMENT = -1,
    WV_E_INVALID_MESH = -2, /* node types / indices inconsistent with the data contract */
    WV_E_HIP = -3,          /* a HIP runtime call failed (message has the HIP error string) */
    WV_E_NO_DEVICE = -4,    /* no gfx950-class device visible: there is NO CPU fallback */
    WV_E_COMM = -5,         /* RCCL failure in the halo exchange */
    WV_E_STATE = -6         /* call not valid in the engine's current state */
};

/* ---- data contract: the reference's device structs, byte for byte -------------------------- */

/* boundary_type bits -- src/waveguide/include/waveguide/cl/utils.h:11-21 */
enum {
    WV_ID_NONE = 0,
    WV_ID_INSIDE = 1 << 0,
    WV_ID_NX = 1 << 1,
    WV_ID_PX = 1 << 2,
    WV_ID_NY = 1 << 3,
    WV_ID_PY = 1 << 4,
    WV_ID_NZ = 1 << 5,
    WV_ID_PZ = 1 << 6,
    WV_ID_REENTRANT = 1 << 7
};

/* error_code bits -- src/waveguide/include/waveguide/cl/structs.h:8-15 */
enum {
    WV_FLAG_SUCCESS = 0,
    WV_FLAG_INF = 1 << 0,
    WV_FLAG_NAN = 1 << 1,
    WV_FLAG_OUTSIDE_RANGE = 1 << 2,
    WV_FLAG_OUTSIDE_MESH = 1 << 3,
    WV_FLAG_SUSPICIOUS_BOUNDARY = 1 << 4
};

/* condensed_node -- cl/structs.h:19-22 (8 bytes) */
typedef struct wv_condensed_node {
    int32_t boundary_type;
    uint32_t boundary_index;
} wv_condensed_node;

/* coefficients_canonical -- cl/filter_structs.h:39-44,65-66 (112 bytes) */
typedef struct wv_coefficients_canonical {
    double b[7];
    double a[7];
} wv_coefficients_canonical;

/* boundary_data -- cl/structs.h:38-41 (56 bytes); boundary_data_array<D> is D of these */
typedef struct wv_boundary_data {
    double filter_memory[6];
    uint32_t coefficient_index;
    uint32_t reserved_;
} wv_boundary_data;

/* `waveguide::mesh` as `run` reads it (mesh.h:12-26, setup.h:27-48,
 * cl/boundary_index_array.h:8-11).  All arrays are borrowed for the duration of wv_create. */
typedef struct wv_mesh {
    int32_t nx, ny, nz;                            /* mesh_descriptor::dimensions */
    const wv_condensed_node* nodes;                /* [nx*ny*nz] */
    const wv_coefficients_canonical* coefficients; /* [num_coefficients] */
    uint32_t num_coefficients;
    const uint32_t* boundary_indices_1; /* [num_boundary_1][1] surface index per filter */
    const uint32_t* boundary_indices_2; /* [num_boundary_2][2] */
    const uint32_t* boundary_indices_3; /* [num_boundary_3][3] */
    uint64_t num_boundary_1, num_boundary_2, num_boundary_3;
} wv_mesh;

/* ---- engine options -------------------------------------------------------------------------- */
enum { WV_PRECISION_F32 = 0, /* pressures as the reference stores them (cl_float) */
       WV_PRECISION_F64 = 1  /* pressures in double: BASELINE.json north star */ };

typedef struct wv_tuning {
    int32_t pair;             /* two-step passes (pair_kernels.hip.h): -1 the engine decides by mesh size, 1 / 0 force on / off */
    int32_t pair_chunks;      /* workgroups along z of the march; 0 = fill whole rounds of workgroup slots */
    int32_t pair_inner_fix;   /* 1: 1-D boundary entries finish the inside node they face; 0: all such nodes go to the fix-up list */
    int32_t pair_wide;        /* 1: rows of more than 8 waves are shared by several workgroups; 0: such meshes keep single steps */
    int32_t pair_unit_waves;  /* sparse rooms: 1 = a listed unit runs only the live waves of its row */
    int32_t pair_unit_planes; /* sparse rooms: planes per work-list unit of the march (default 32) */
    int32_t pair_units_by_chunk; /* sparse rooms: 1 = an XCD takes its units chunk by chunk (neighbouring strips run together); 0 = strip by strip */
    int32_t tile_lists;       /* 1: rooms that leave much of the mesh outside visit live tiles / units only (see all_tiles) */
    int32_t fuse_pre_post;    /* 1: the next step's source / receiver work rides in this step's boundary launch where legal */
    int32_t graph;            /* 1: batches of single steps on small meshes are replayed as a hipGraph */
    int32_t boundary_lds;     /* 1: boundary workgroups stage the coefficient sets in LDS (<= 256 sets) */
    int32_t boundary_order;   /* 1: boundary entries processed in 64x8x8-brick order; 0: in the caller's order */
    int32_t boundary_xwall;   /* 1: in two- and three-step passes the wall nodes that face along x work on compact copies of what they would gather
                               * from the fields; 2: in two-step passes only (measurement); 0: nowhere */
    int32_t stream_ry, stream_nwx, stream_nwy, stream_zchunks; /* sweep tile shape as wv_set_stream_tuning; 0 = automatic */
    int32_t slab_early;       /* z-slabs, two-step passes: 1 = the faces AND the planes next to them are stepped ahead of the march, so that both
                               * halo exchanges of a pass (and the faces' second step, on the halo stream) run under it; 0 = the second exchange
                               * follows the march (the form of rounds 2 and 3); -1 (default) = 1 where a neighbour lives on another GPU (RCCL,
                               * or a slab of this process on another device), 0 between slabs that share a device.  Read once, at
                               * wv_create (the x-facing walls' compact copies leave out the planes an early pass steps ahead) */
    int32_t pair_split_rows;  /* 1: rows of 3..8 waves are marched as two overlapping windows (two smaller workgroups per CU); measurement only */
    int32_t fuse_planes;      /* z-slabs: 1 = the planes stepped around the halo exchanges take ONE launch (sweep + their boundary entries side by side) */
    int32_t whole_step;       /* single steps as ONE launch each (sweep workgroups and boundary workgroups side by side, the next step's source /
                               * receiver work served by the tiles that own those nodes): -1 the engine decides by mesh size (small meshes are
                               * bound by launches, not bytes), 1 / 0 force on / off.  Only where it is legal (one domain, the source and the
                               * receivers -- at most 63 -- on inside nodes); otherwise two launches per step as ever */
    int32_t triple;           /* three-step passes (triple_kernels.hip.h: 10.7 B per node-update where a two-step pass moves 16): -1 the engine
                               * decides by mesh size, 1 / 0 force on / off.  Wherever two-step passes run: one domain, a z-slab of a chain
                               * (three halo exchanges per pass; every rank's consent, like two-step passes), a room that leaves much of its
                               * mesh outside (a work list of live pieces).  A batch takes them wherever it has three steps left, then a
                               * two-step pass or a single step */
    int32_t triple_chunks;    /* workgroups along z of the three-step march; 0 = fill whole rounds of workgroup slots */
    int32_t triple_lanes;     /* bytes of a row per lane of the three-step march: 0 the engine decides (by row length, whichever ran faster on
                               * boxes of that size), 8 / 16 force */
} wv_tuning;

typedef struct wv_options {
    int32_t struct_size; /* = sizeof(wv_options); lets the struct grow */
    int32_t precision;   /* WV_PRECISION_* */
    int32_t device;      /* HIP device ordinal; -1 = the calling thread's current device */
    /* z-slab decomposition: plane z=0 (ghost_lo) / z=nz-1 (ghost_hi) of this mesh is a ghost
     * copy of the neighbouring rank's face plane; it is read, never updated, by this engine. */
    int32_t ghost_lo, ghost_hi;
    /* the error flag is brought to the host every `flag_interval` steps of wv_run
     * (1 = after every step, like waveguide.h:100-101; 0 = once per wv_run call) */
    int32_t flag_interval;
    int32_t stream_variant; /* 2 = plane sweep, y halos through LDS (default); kept for measurement:
                             * 3 = plane sweep without LDS, 0 = register z-march, 1 = naive */
    /* 0 (default): when a room leaves part of the mesh outside, the sweep visits only the tiles
     * that hold inside nodes (outside nodes are 0 and stay 0; wv_write_field / wv_write_value
     * re-enable the full sweep until they are 0 again).  1: always visit every tile. */
    int32_t all_tiles;
    /* 1: wv_mesh::nodes is a device pointer on `device` (what wv_scene_mesh_create_engine passes);
     * the boundary index and coefficient arrays are host arrays either way */
    int32_t nodes_on_device;
    /* z-slabs over RCCL: seconds a rank waits for a batch of steps (or for the ranks' agreement before one) before it gives up on
     * its peers -- WV_E_COMM, wv_last_error naming the rank, its neighbours and the stream that had not drained; the communicator
     * is aborted and the engine is good for wv_destroy only.  0 = 180 s, < 0 = wait for ever (plain hipStreamSynchronize). */
    int32_t comm_timeout_s;
    /* z-slabs, one rank per process (wv_comm_init): how the face planes reach the neighbouring ranks.
     *   WV_TRANSPORT_RCCL (default)  grouped ncclSend / ncclRecv on the halo stream;
     *   WV_TRANSPORT_IPC             copies into the neighbours' own fields, mapped with hipIpcOpenMemHandle (the handles travel
     *                                through the communicator at wv_comm_init; a copy engine moves the planes, no send / receive
     *                                kernel has to find CUs beside the march); the ranks' agreements and the flag OR stay on RCCL.
     *                                All ranks of a chain choose the same.  The engine then holds its four fields from wv_comm_init on. */
    int32_t transport;
    int32_t reserved_[5];
    /* HOW the engine does its work -- never what it computes: every setting gives bit-identical results
     * (tests/test_gpu_parity.py, test_gpu_pair.py run the golden cases under each).  wv_default_options
     * fills in the product's choices; the fields exist for measurement and for the tests.  The library
     * reads no environment variables (built with -DWV_DEBUG_ENV, WV_<FIELD> overrides a field: tools only). */
    wv_tuning tuning;
} wv_options;

enum { WV_TRANSPORT_RCCL = 0, WV_TRANSPORT_IPC = 1 };

typedef struct wv_engine wv_engine;

/* ---- life cycle ------------------------------------------------------------------------------ */

/* Replaces the set-up half of `run` (waveguide.h:43-76): zeroed previous/current fields, node,
 * coefficient and boundary-state buffers (get_boundary_data<N>, setup.h:68-85). */
int wv_create(const wv_mesh* mesh, const wv_options* options, wv_engine** out);
void wv_destroy(wv_engine* e);
const char* wv_last_error(void);
/* Fills `options` with defaults: F64 pressures (the default of every layer above this ABI too; F32 is
 * the reference's cl_float storage, bit for bit), the calling thread's current device, no ghosts,
 * flag_interval 0 (the flag words of a wv_run batch are read back once per batch; the step that
 * raised a flag is still reported exactly, see wv_run). */
void wv_default_options(wv_options* options);

/* ---- buffer access used by step pre/post-processors ----------------------------------------- */
enum { WV_BUF_CURRENT = 0, WV_BUF_PREVIOUS = 1 };

/* core::read_value / core::write_value on the pressure buffer
 * (src/core/include/core/cl/common.h:42-57); value converted to/from the engine precision. */
int wv_read_value(wv_engine* e, int buffer, uint64_t index, double* value);
int wv_write_value(wv_engine* e, int buffer, uint64_t index, double value);
/* core::read_from_buffer / cl::copy of the whole field (common.h:34-40;
 * preprocessor/gaussian.cpp:50).  elem_size 4 -> float[n], 8 -> double[n]. */
int wv_read_field(wv_engine* e, int buffer, void* dst, int elem_size);
int wv_write_field(wv_engine* e, int buffer, const void* src, int elem_size);
/* The same for planes [z_begin, z_begin + z_count) only: dst / src hold z_count*ny*nx elements.
 * (A 1024^3 field is 8.6 GB: a visualiser slice or a slab hand-over should not have to move it all.) */
int wv_read_planes(wv_engine* e, int buffer, int32_t z_begin, int32_t z_count, void* dst, int elem_size);
int wv_write_planes(wv_engine* e, int buffer, int32_t z_begin, int32_t z_count, const void* src, int elem_size);
/* Read back / restore boundary filter state in the reference layout boundary_data_array<D>[n_D]. */
int wv_read_boundary_data(wv_engine* e, int dimensionality, wv_boundary_data* dst);
int wv_write_boundary_data(wv_engine* e, int dimensionality, const wv_boundary_data* src);
/* mesh::set_coefficients (src/waveguide/src/setup.cpp:38-50): n must equal num_coefficients */
int wv_set_coefficients(wv_engine* e, const wv_coefficients_canonical* c, uint32_t n);
/* Page-locks `bytes` of the caller's host memory at `p` for the device (hipHostRegister) / releases it: reads and writes of fields,
 * planes and boundary data whose host side is registered memory go by DMA at the link's rate instead of through the runtime's staging
 * buffers (what cl_mirror.h does with the staging area of its cl::Buffer mirror).  Optional; WV_E_HIP when the runtime refuses. */
int wv_host_register(void* p, uint64_t bytes);
int wv_host_unregister(void* p);
/* Device addresses of the fields (element type per precision), for zero-copy wrappers.  Once a
 * pointer has been handed out the engine stops assuming that outside nodes hold zeros (see
 * wv_options::all_tiles) and keeps to one time step per pass over two fields, so that the pointers stay
 * the two fields (an engine that takes two-step passes rotates four). */
int wv_device_buffer(wv_engine* e, int buffer, void** device_ptr);

/* The state `run` carries from one loop iteration to the next (waveguide.h:80-123: `previous`, `current` and the boundary filter
 * memories), with the step count, the position in the source signal and the recorded receiver rows, copied aside ON THE DEVICE
 * (wv_checkpoint: two more fields of memory, allocated by the first call; WV_E_HIP when there is no room, the engine untouched) and
 * put back (wv_rollback: the engine continues from the checkpoint and, being deterministic, reproduces the abandoned steps bit for
 * bit).  For callers that run batches of steps ahead of per-step observers: `canonical`'s pressure callback may look at the field
 * of ANY step (canonical.h:66-69), so the C++ mirror runs batches speculatively and re-runs up to the step an observer looks at.
 * The source and the receivers must be the ones in place at the checkpoint (WV_E_STATE otherwise); wv_drop_checkpoint frees the copy.
 * One domain only: a slab of a chain (ghost planes, or a communicator of more than one rank) answers WV_E_STATE -- its neighbours'
 * planes and the transport's counters would have to go back with it. */
int wv_checkpoint(wv_engine* e);
int wv_rollback(wv_engine* e);
int wv_drop_checkpoint(wv_engine* e);

/* ---- stepping: the generic path ------------------------------------------------------------- */

/* One loop body of waveguide.h:82-119 without the callbacks: clear flag, launch the update
 * (previous <- next in place), read the flag back.  *flag receives the error_code bits. */
int wv_step(wv_engine* e, int32_t* flag);
/* std::swap(previous, current), waveguide.h:123 */
int wv_swap(wv_engine* e);

/* ---- stepping: the device-resident fast path ------------------------------------------------- */
enum { WV_SOURCE_NONE = 0,
       WV_SOURCE_HARD = 1, /* preprocessor::hard_source, preprocessor/hard_source.h:17-23 */
       WV_SOURCE_SOFT = 2  /* preprocessor::soft_source, preprocessor/soft_source.h:17-25 */ };

/* Signal injected at `node`, one sample per step, starting at the engine's current step. */
int wv_set_source(wv_engine* e, int kind, uint64_t node, const double* signal, uint64_t n);
/* Nodes whose pre-update `current` pressure is recorded every step
 * (postprocessor::node, src/postprocessor/node.cpp:14-18; the 7 reads of
 * directional_receiver.cpp:33-47).  node == UINT64_MAX records 0. */
int wv_set_receivers(wv_engine* e, const uint64_t* nodes, uint32_t n);
/* Run up to n_steps loop iterations on the device.  Stops early at the first step whose flag
 * is non-zero: *steps_done = completed steps (that step excluded), *flag = its error bits. */
int wv_run(wv_engine* e, uint64_t n_steps, uint64_t* steps_done, int32_t* flag);
/* Receiver samples of steps [first, first+n) as double[n][num_receivers]; steps driven by wv_step / wv_swap
 * record nothing (their rows are NaN). */
int wv_fetch_receivers(wv_engine* e, uint64_t first, uint64_t n, double* dst);
/* Number of loop iterations completed since creation. */
int wv_step_count(wv_engine* e, uint64_t* steps);

/* ---- timing hooks (bench.py) ------------------------------------------------------------------ */
/* Mean duration in ms of the dominant (pressure update) kernel over the launches since the
 * last call, measured with HIP events on the engine's own stream; 0 launches -> 0. */
int wv_kernel_time_ms(wv_engine* e, double* mean_ms, uint64_t* launches);
int wv_enable_kernel_timing(wv_engine* e, int enable);
/* The same plus the number of time steps the timed launches covered: on meshes big enough to be bound
 * by HBM bytes the engine advances TWO steps per pass over the fields (pair_kernels.hip.h; results are
 * bit-identical to single steps), so a launch of the dominant kernel may stand for two steps. */
int wv_kernel_time_detail(wv_engine* e, double* mean_ms, uint64_t* launches, uint64_t* steps);
/* What the engine is doing, for tests and tools (never needed to use it): *value receives
 *   WV_QUERY_PASSES          two-step passes taken since creation
 *   WV_QUERY_XWALL_ENTRIES   wall nodes that work on compact copies in two-step passes right now (0: none / not in use)
 *   WV_QUERY_FIELDS          pressure fields allocated (2, or 4 once two-step passes have been taken)
 *   WV_QUERY_MARCH_LIVE_PERMILLE   rooms that leave part of the mesh outside: the share of the mesh (in wave-sized
 *                            pieces of rows, per 1000) that the two-step march visits; 1000 when it visits everything
 *   WV_QUERY_SWEEP_LIVE_PERMILLE   the same for the single-step sweep's tiles
 *   WV_QUERY_MARCH_ROUNDS    how many times over the two-step march's workgroups fill the chip's workgroup slots (0 before the
 *                            first pass).  A slab with a neighbour marches in two rounds at least where that costs little, so
 *                            that the exchange of its t+1 faces gets a CU before the march ends
 *   WV_QUERY_HALO_WAIT_NS, WV_QUERY_HALO_WAITS   z-slabs with kernel timing on: total time the compute stream stood waiting for
 *                            ghost planes (the part of the halo exchange the interior work did not hide), over that many timed
 *                            waits (every fourth); both reset by wv_kernel_time
 *   WV_QUERY_HALO_EXCHANGES, WV_QUERY_HALO_BYTES_SENT   exchanges issued and bytes handed to neighbours since creation
 *   WV_QUERY_EARLY_PASSES    two-step passes of a slab that ran both exchanges under the march (wv_tuning::slab_early)
 *   WV_QUERY_TRIPLE_PASSES   three-step passes taken since creation (wv_tuning::triple) */
enum { WV_QUERY_PASSES = 0, WV_QUERY_XWALL_ENTRIES = 1, WV_QUERY_FIELDS = 2, WV_QUERY_MARCH_LIVE_PERMILLE = 3,
       WV_QUERY_SWEEP_LIVE_PERMILLE = 4, WV_QUERY_MARCH_ROUNDS = 5, WV_QUERY_HALO_WAIT_NS = 6, WV_QUERY_HALO_WAITS = 7,
       WV_QUERY_HALO_EXCHANGES = 8, WV_QUERY_HALO_BYTES_SENT = 9, WV_QUERY_EARLY_PASSES = 10,
       /* kernel timing on: total time of the two boundary launches of every eighth two-step pass whose march was timed (nodes to t+1 /
        * to t+2), over that many passes; reset by wv_kernel_time */
       WV_QUERY_BOUNDARY1_NS = 11, WV_QUERY_BOUNDARY2_NS = 12, WV_QUERY_BOUNDARY_TIMED = 13,
       WV_QUERY_WHOLE_STEPS = 14 /* single steps taken as one launch each (wv_tuning::whole_step) */,
       WV_QUERY_TRIPLE_PASSES = 15,
       /* kernel timing of the three-step passes (wv_enable_kernel_timing; reset by wv_kernel_time_ms like the rest): total time and count
        * of the timed three-step marches (kept apart from wv_kernel_time_ms's account, which is the two-step march's or the sweep's); of
        * the passes whose parts were timed (every eighth timed pass): their third boundary launch and their third-level fix-up list
        * (WV_QUERY_BOUNDARY1_NS / 2_NS count the first two boundary launches of either kind of pass) */
       WV_QUERY_TRIPLE_MARCH_NS = 16, WV_QUERY_TRIPLE_MARCH_TIMED = 17, WV_QUERY_BOUNDARY3_NS = 18, WV_QUERY_FIXUP3_NS = 19,
       WV_QUERY_TRIPLE_PARTS_TIMED = 20 };
int wv_query(wv_engine* e, int what, uint64_t* value);
/* hipStreamSynchronize on every engine stream. */
int wv_synchronize(wv_engine* e);
/* Tuning hook for the streaming kernel.  variant 2 = plane sweep, 0 = register z-march, 1 = naive.
 * rows_per_wave in {2,4}; a workgroup is waves_x by waves_y waves; `knob` = rows per XCD stripe
 * (variant 2) or workgroups along z (variant 0); 0 = automatic everywhere. */
int wv_set_stream_tuning(wv_engine* e, int variant, int rows_per_wave, int waves_x, int waves_y, int knob);

/* ---- z-slab halo exchange over RCCL (multi-GPU) ---------------------------- */
#define WV_UNIQUE_ID_BYTES 128
/* The shared library the RCCL entry points are resolved in (dlopen of exactly this path) instead of the librccl the
 * process finds by name; NULL / "" = by name again.  Process-wide, before the first communicator call (WV_E_STATE
 * afterwards).  For deployments with RCCL outside the loader's path -- and for the test stand-ins under tests/mock_rccl. */
int wv_comm_use_library(const char* path);
/* rank 0 creates the id; the caller distributes the bytes (e.g. torch.distributed broadcast) */
int wv_comm_unique_id(void* id_bytes /* [WV_UNIQUE_ID_BYTES] */);
/* Joins a communicator: this engine is slab `rank` of `nranks`, neighbours rank-1 / rank+1. */
int wv_comm_init(wv_engine* e, const void* id_bytes, int rank, int nranks);
int wv_comm_destroy(wv_engine* e);
/* On a chain of nranks > 1 every rank calls wv_run with the same n_steps.  Before each batch of steps the ranks
 * agree (one small all-reduce) on its length -- the ranks that hold the source plane know where the signal ends, and
 * the run ends there on every rank with the same *steps_done, as it does on one device (hard_source.h:18-20) -- and
 * on the form of its steps (two-step passes need every rank's consent); at the end of every batch the per-step flag
 * words are OR-ed over the ranks (one small all-reduce), so a NaN / Inf / bad-boundary flag raised
 * on one slab stops all of them at the same step -- the multi-device form of waveguide.h:100-119.
 *
 * The same chain inside ONE process (several engines on one GPU, or one per GPU of a node driven
 * from one thread): engines[r] is slab r, created with ghost_lo = (r > 0), ghost_hi = (r < n - 1);
 * face planes travel by device-to-device copies instead of RCCL, everything else in a step is the
 * same code.  Slabs joined this way are stepped together with wv_run_group (same contract as wv_run:
 * *steps_done and *flag are those of the chain; receivers are fetched per engine as usual). */
int wv_comm_init_local(wv_engine* const* engines, int32_t n);
int wv_run_group(wv_engine* const* engines, int32_t n, uint64_t n_steps, uint64_t* steps_done, int32_t* flag);

/* Measured device triad a[i] = b[i] + s*c[i] over n_doubles doubles per array (2 reads + 1 write, the
 * stencil's byte mix), mean of `iters` launches: the bandwidth yardstick of SURVEY.md 8(d). */
int wv_measure_triad(int32_t device, uint64_t n_doubles, int32_t iters, double* gb_per_s);

/* ---- unit kernel of the boundary IIR step ------------------------------------------------------- */
/* The reference's `filter_test_2` test kernel (src/waveguide/src/cl/filters.cpp:66-75, launched by
 * tests/rectangular_kernel.cpp:170-190): n_filters independent order-6 filters, each fed
 * input[s][f] for s = 0..n_samples-1; output[s][f] = filter output as float; memory[f][6] is
 * read, advanced and written back.  Runs the same device code as the boundary kernel. */
int wv_filter_test_2(const float* input, float* output, double* memory,
                     const wv_coefficients_canonical* coeffs, uint32_t n_filters, uint32_t n_samples);
/* Row length (in elements) of the stored pressure fields returned by wv_device_buffer:
 * nx rounded up to the wave tile (128 doubles / 256 floats); element (x,y,z) is at
 * (z*ny + y)*pitch + x and the pad columns are zero. */
int wv_field_pitch(wv_engine* e, uint64_t* pitch_elements);

/* ---- host helpers ------------------------------------------------------------------------------ */
/* Synthetic box mesh of SURVEY.md 8(d): planes [z_begin, z_begin+z_count) of a global
 * nx*ny*nz_global box; boundary_index numbered per dimensionality in increasing node index
 * over planes [number_from, number_to) (set_boundary_index,
 * src/waveguide/src/boundary_coefficient_finder.cpp:11-19); nodes outside that range get index 0.
 * counts[3] receives the number of 1D/2D/3D boundary nodes numbered. */
int wv_make_box_nodes(int32_t nx, int32_t ny, int32_t nz_global, int32_t z_begin, int32_t z_count,
                      int32_t number_from, int32_t number_to, wv_condensed_node* nodes,
                      uint64_t counts[3]);

/* ---- mesh set-up (first slice) -------------------------------------------- */
/* From per-node inside flags (what `set_node_inside` yields, mesh_setup_program.cpp:110-140) to
 * the `condensed_node` array: the reference's `set_node_boundary_type` kernel
 * (src/waveguide/src/mesh_setup_program.cpp:66-108,142-172) on the GPU, then `set_boundary_index`
 * as compute_boundary_index_data applies it (boundary_coefficient_finder.cpp:11-19,44-54):
 * counts[0] = 1-D boundary OR re-entrant nodes, counts[1] = 2-D, counts[2] = 3-D.
 * inside: uint8[nx*ny*nz], non-zero = inside the model. */
int wv_classify_nodes(int32_t nx, int32_t ny, int32_t nz, const uint8_t* inside, wv_condensed_node* nodes,
                      uint64_t counts[3]);

/* Second slice: the inside flags themselves, for triangle-soup scenes.
 * vertices: float[n][4] (cl_float3); triangles: uint32[m][4] = {surface, v0, v1, v2}
 * (src/core/include/core/cl/triangle.h:9-14).
 *
 * wv_voxelise (host): the flattened voxel -> triangle-list array of `get_flattened`
 * (src/core/src/spatial_division/voxel_collection.cpp:9-37) over a side^3 grid on [aabb_min,
 * aabb_max]; a triangle is listed in every voxel whose box, padded by 0.001, it overlaps
 * (src/core/include/core/spatial_division/voxelised_scene_data.h:28-44).  Two-call protocol:
 * *needed always receives the word count; nothing is written unless capacity >= *needed.
 *
 * wv_nodes_inside (GPU): the reference's `set_node_inside` kernel
 * (src/waveguide/src/mesh_setup_program.cpp:110-140; voxel ray-parity test
 * src/core/src/cl/voxel.cpp:98-225) for every node of the mesh (nx, ny, nz, min_corner, spacing):
 * inside[i] = 1 / 0. */
int wv_voxelise(const float* vertices, uint32_t n_vertices, const uint32_t* triangles, uint32_t n_triangles,
                const float aabb_min[3], const float aabb_max[3], uint32_t side, uint32_t* out, uint64_t capacity,
                uint64_t* needed);
int wv_nodes_inside(int32_t nx, int32_t ny, int32_t nz, const float min_corner[3], float spacing,
                    const uint32_t* voxel_index, uint64_t n_voxel_words, const float aabb_min[3],
                    const float aabb_max[3], uint32_t side, const uint32_t* triangles, uint32_t n_triangles,
                    const float* vertices, uint32_t n_vertices, uint8_t* inside);

/* Third slice: which scene surface each boundary filter takes -- compute_boundary_index_data
 * (src/waveguide/src/boundary_coefficient_finder.cpp:38-131) and its kernels
 * boundary_coefficient_finder_1d/_2d/_3d (src/waveguide/src/boundary_coefficient_program.cpp:
 * 310-338, 356-413, 429-484; nearest triangle by exact point-triangle distance, :16-143,218-235).
 * nodes: in, boundary_type as wv_classify_nodes leaves it (boundary_index is ignored);
 *        out, boundary_index as `run` wants it (1-D numbering without the re-entrant nodes).
 * b1 [counts[0]][1], b2 [counts[1]][2], b3 [counts[2]][3]: surface index per filter, i.e. the
 * wv_mesh::boundary_indices_* arrays.  counts[] is always written; with b1 = b2 = b3 = NULL the call
 * is a size query.  WV_E_INVALID_ARGUMENT when a capacity (in rows) is too small or when the
 * mesh lacks 1-D, 2-D or 3-D boundary nodes ("No boundaries.", boundary_coefficient_finder.cpp:30-33).
 * Entry 0 of the 1-D array is written by its owner only (the reference lets every inside node race
 * for it). */
int wv_boundary_index_data(int32_t nx, int32_t ny, int32_t nz, const float min_corner[3], float spacing,
                           wv_condensed_node* nodes, const uint32_t* triangles, uint32_t n_triangles,
                           const float* vertices, uint32_t n_vertices, uint32_t* b1, uint64_t capacity_1,
                           uint32_t* b2, uint64_t capacity_2, uint32_t* b3, uint64_t capacity_3,
                           uint64_t counts[3]);

/* The same three stages chained on the device (compute_mesh, src/waveguide/src/mesh.cpp:54-141, as
 * one unit): nothing but the scene goes up and nothing comes down unless asked for.
 *   wv_scene_mesh_create         inside flags -> node types -> numbering -> surfaces per filter, all in
 *                                HBM on `device` (-1 = current); counts[] = rows of the 1-D/2-D/3-D
 *                                boundary arrays; "No boundaries." like the reference when one is 0
 *   wv_scene_mesh_fetch          host copies (any pointer may be NULL): nodes [nx*ny*nz], b1 [c0][1],
 *                                b2 [c1][2], b3 [c2][3] -- identical to the three-call path above
 *   wv_scene_mesh_create_engine  wv_create on the device-resident nodes (options->device is
 *                                overridden by the scene mesh's device)
 */
typedef struct wv_scene_mesh wv_scene_mesh;
int wv_scene_mesh_create(int32_t nx, int32_t ny, int32_t nz, const float min_corner[3], float spacing,
                         const uint32_t* voxel_index, uint64_t n_voxel_words, const float aabb_min[3],
                         const float aabb_max[3], uint32_t side, const uint32_t* triangles, uint32_t n_triangles,
                         const float* vertices, uint32_t n_vertices, int32_t device, wv_scene_mesh** out,
                         uint64_t counts[3]);
int wv_scene_mesh_fetch(const wv_scene_mesh* sm, wv_condensed_node* nodes, uint32_t* b1, uint32_t* b2, uint32_t* b3);
int wv_scene_mesh_create_engine(const wv_scene_mesh* sm, const wv_coefficients_canonical* coefficients,
                                uint32_t num_coefficients, const wv_options* options, wv_engine** out);
void wv_scene_mesh_destroy(wv_scene_mesh* sm);

/* ---- boundary filter design, host side --------------------------------- */
/* arbitrary_magnitude_filter<6> (src/waveguide/include/waveguide/arbitrary_magnitude_filter.h:63-95):
 * (frequency 0..1 = DC..Nyquist, amplitude) points in any order -> order-6 IIR b/a whose magnitude
 * approximates them.  Points outside [0, 1] are dropped, (0,0) and (1,0) are added, the envelope is
 * resampled to 256 points by linear interpolation and fitted by the modified Yule-Walker method
 * (the reference calls itpp::yulewalk; see wayverb_amd/csrc/filter_design.cpp for what is restated). */
int wv_arbitrary_magnitude_filter(const double* frequency, const double* amplitude, uint32_t n_points,
                                  double b[7], double a[7]);
/* is_stable (src/waveguide/include/waveguide/stable.h:43-50) on ascending-power denominator
 * coefficients a[0..n-1] */
int wv_is_stable(const double* a, uint32_t n, int32_t* stable);
/* hrtf_data::hrtf_band_centres (src/hrtf/lib/include/hrtf/multiband.h:13-20): centres of the 8
 * simulation bands over 20 Hz..20 kHz, divided by sample_rate */
int wv_band_centres(double sample_rate, double centres[8]);
/* compute_reflectance_filter_coefficients (fitted_boundary.h:79-104): 8 band absorptions ->
 * pressure reflectance sqrt(1 - absorption) at 2*centre/sample_rate -> the filter above; fails with
 * "Unable to generate stable boundary filter." when the denominator is not stable */
int wv_reflectance_filter(const double absorption[8], double sample_rate, wv_coefficients_canonical* out);
/* to_impedance_coefficients (fitted_boundary.h:21-48): b' = a + b, a' = a - b, scaled by 1/a'[0]
 * when that is non-zero: what wv_mesh::coefficients holds for a surface (mesh.cpp:126-138) */
int wv_impedance_coefficients(const wv_coefficients_canonical* reflectance, wv_coefficients_canonical* impedance);

/* ---- receiver traces -> audio, host side --------------------------------- */
/* postprocessor::directional_receiver::output (src/waveguide/include/waveguide/postprocessor/
 * directional_receiver.h:22-25): what `canonical` collects per step */
typedef struct wv_directional_output {
    float intensity[3];
    float pressure;
} wv_directional_output;
/* bandpass_band (src/waveguide/include/waveguide/bandpass_band.h:11-20) */
typedef struct wv_waveguide_band {
    const wv_directional_output* directional;
    uint64_t n;
    double sample_rate;
    double valid_hz_min, valid_hz_max;
} wv_waveguide_band;
enum { WV_ATTENUATOR_NULL = 0,       /* core::attenuator::null: the pressure itself */
       WV_ATTENUATOR_MICROPHONE = 1  /* core::attenuator::microphone (pointing, shape) */
       /* core::attenuator::hrtf: 8 bands per sample, see the wv_*_hrtf entry points below */ };
enum { WV_FILTER_LOPASS = 0, WV_FILTER_HIPASS = 1, WV_FILTER_BANDPASS = 2 };

/* attenuate / make_attenuate_mapper (src/waveguide/include/waveguide/attenuator.h:13-49;
 * microphone: src/core/src/attenuator/microphone.cpp:18-25), float arithmetic as there.
 * Fails with "Acoustic impedance outside expected range." unless 300 <= Z < 500. */
int wv_attenuate(int32_t method, const float pointing[3], float shape, float acoustic_impedance,
                 const wv_directional_output* in, uint64_t n, float* out);
/* adjust_sampling_rate (src/waveguide/src/config.cpp:29-56): (size_t)(out/in * n) samples of
 * band-limited interpolation scaled by in/out.  *n_out always receives the length; nothing is
 * written unless capacity suffices.  (The reference calls libsamplerate; see postprocess.cpp.) */
int wv_adjust_sampling_rate(const float* in, uint64_t n, double in_sample_rate, double out_sample_rate, float* out,
                            uint64_t capacity, uint64_t* n_out);
/* frequency_domain::filter{best_fft_length(n) << 2}.run with compute_lopass / hipass /
 * bandpass_magnitude as the per-bin gain (src/frequency_domain/src/filter.cpp:22-47,
 * envelope.cpp:60-112); edges relative to the sample rate, in place */
int wv_frequency_domain_filter(float* signal, uint64_t n, int32_t kind, double edge_lo, double edge_hi,
                               double width_factor, uint32_t steepness);
/* waveguide::postprocess(bandpass_bands, method, Z, output_sample_rate)
 * (src/waveguide/include/waveguide/postprocess.h:74-126): attenuate, resample, band-pass each
 * band at its valid range, sum, 10 Hz DC block.  Size-query protocol as above. */
int wv_postprocess_waveguide(const wv_waveguide_band* bands, uint32_t n_bands, int32_t method, const float pointing[3],
                             float shape, float acoustic_impedance, double output_sample_rate, float* out,
                             uint64_t capacity, uint64_t* n_out);

/* ---- HRTF receiver capsules (core::attenuator::hrtf, src/core/include/core/attenuator/hrtf.h) ---------
 * The reference looks the 8 band energies of a direction up in a table it generates AT BUILD TIME from
 * measured head-related impulse responses (src/hrtf/cmd/main.cpp writes hrtf_entries.h; neither that file
 * nor the measurements are in the reference tree), so the table is the caller's to supply:
 *   energy[az][el][channel][band], az in [0, az_num): azimuth az * 360 / az_num degrees,
 *   el in [0, el_num): elevation (el + 1) * 180 / (el_num + 1) - 90 degrees (el_num odd), channel 0 = left,
 *   nearest-entry lookup exactly as vector_look_up_table.h:50-110 (azimuth = atan2(x, -z) negated, elevation =
 *   asin(y), in the head's frame: pointing = -z, up = +y; orientation.cpp:21-43). */
typedef struct wv_hrtf_table {
    const double* energy;
    uint32_t az_num, el_num;
} wv_hrtf_table;
/* attenuation(hrtf, incident) (src/core/src/attenuator/hrtf.cpp:121-133) */
int wv_hrtf_attenuation(const wv_hrtf_table* table, const float pointing[3], const float up[3], int32_t channel,
                        const float incident[3], float bands[8]);
/* get_ear_position (hrtf.cpp:135-141); fails with "Hrtf radius outside reasonable range." unless 0 <= radius <= 1 */
int wv_hrtf_ear_position(const float pointing[3], const float up[3], int32_t channel, float radius,
                         const float base_position[3], float ear[3]);
/* attenuate / make_attenuate_mapper with an hrtf method (attenuator.h:13-49): out[n][8] */
int wv_attenuate_hrtf(const wv_hrtf_table* table, const float pointing[3], const float up[3], int32_t channel,
                      float acoustic_impedance, const wv_directional_output* in, uint64_t n, float* out);
/* core::multiband_filter_and_mixdown (src/core/include/core/mixdown.h:17-26; hrtf_data::multiband_filter,
 * src/hrtf/lib/include/hrtf/multiband.h:38-44): bands[n][8] is filtered in place, out[n] = sum of the 8 bands */
int wv_multiband_filter_and_mixdown(float* bands, uint64_t n, double sample_rate, float* out);
/* waveguide::postprocess with an hrtf method (postprocess.h:57-126).  Size-query protocol as above. */
int wv_postprocess_waveguide_hrtf(const wv_waveguide_band* bands, uint32_t n_bands, const wv_hrtf_table* table,
                                  const float pointing[3], const float up[3], int32_t channel, float acoustic_impedance,
                                  double output_sample_rate, float* out, uint64_t capacity, uint64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* WAYVERB_AMD_H */
