// engine.hip.h -- `Engine<Real>`: one `waveguide::run` worth of device state and the host logic that steps it
// (src/waveguide/include/waveguide/waveguide.h:36-126 is what it replaces).  Real = float (the reference's cl_float
// fields) or double (BASELINE.json's fp64 engine).
//
// The member functions are defined in:
//   engine_setup.hip.h   wv_create: buffers, class map, boundary entry lists and their processing order, sweep plan,
//                        work lists for rooms that leave much of the mesh outside; release
//   engine_single.hip.h  one time step per pass: sweep + boundary launches, source / receiver launch, the slab form
//                        (faces first, exchange, interior), hipGraph replay for small meshes
//   engine_pair.hip.h    two time steps per pass: eligibility, pair map + fix-up lists, march geometry, parts A / B
//   engine_triple.hip.h  three time steps per pass: eligibility, triple map + third-level list, march geometry, the pass
//   engine_batch.hip.h   wv_step / wv_run: batches of steps, flag words, kernel timing
//   engine_io.hip.h      everything a caller reads or writes: values, fields, planes, filter memories, source,
//                        receivers
//   engine_slab.hip.h    z-slab chains: communicators, the in-process group (wv_comm_init_local / wv_run_group)
// There is no CPU path: without a HIP device every entry point fails.
#pragma once
#include "engine_base.h"

#include "boundary_kernels.hip.h"
#include "pair_kernels.hip.h"
#include "stream_kernels.hip.h"
#include "plane_kernels.hip.h"
#include "triple_kernels.hip.h"

namespace wv {

template <typename Real>
class Engine final : public wv_engine {
public:
    ~Engine() override { release(); }
    // ---- engine_setup.hip.h
    int init(const wv_mesh& m, const wv_options& opt) override;
    int set_tuning(int variant, int ry, int nwx, int nwy, int zchunks) override;
    int build_plane_order();
    void plan_stream();
    // ---- engine_single.hip.h
    template <int RY, int NWX, int NWY>
    void launch_shape(const wv::StreamArgs<Real>& a, unsigned grid);
    template <int RY>
    void launch_ry(const wv::StreamArgs<Real>& a, unsigned grid);
    // ---- engine_setup.hip.h
    int build_tile_lists(int z0, int z1);
    // ---- engine_single.hip.h
    // (`plan_only`: the launch's arguments and grid are handed back instead of launched -- launch_faces puts them into one launch
    // with the planes' boundary entries)
    struct StreamLaunch {
        wv::StreamArgs<Real> args;
        unsigned grid = 0;
    };
    int launch_stream(Real* prev, const Real* cur, int* flag, int z0, int z1, bool timed, Real* out = nullptr, int zb0 = 0, int zb1 = 0,
                      StreamLaunch* plan_only = nullptr);
    // (`planes`: how many owned planes next to each neighbour -- 1: the face planes, 2: the faces and the planes next to them)
    int launch_faces(Real* prev, const Real* cur, int* flag, Real* out, int planes = 1);
    wv::BoundaryArgs<Real> boundary_args(Real* prev, const Real* cur, int* flag) const;
    // (z0 = z1 = -1: the boundary nodes of a slab's face planes)
    // (z0 = z1 = -2: of the two planes next to each neighbour; `levels`: a two-step pass's launch over the bulk of the mesh,
    // in which the x-facing walls go by position on their compact copies)
    struct BoundaryLaunch {
        wv::BoundaryArgs<Real> args;
        unsigned blocks = 0;
        bool lds = false;
    };
    int launch_boundary(Real* prev, const Real* cur, int* flag, int z0, int z1, const wv::PrePostArgs<Real>* next = nullptr, Real* out = nullptr, bool fix_inner = false,
                        bool levels = false, BoundaryLaunch* plan_only = nullptr, int xw3 = 0);
    wv::PrePostArgs<Real> pre_post_args(Real* cur, int slot, bool with_pre_post, uint64_t signal_pos, bool source_live) const;
    int enqueue_step(int slot, bool with_pre_post, uint64_t signal_pos, bool source_live, int fuse_next = 0);
    // one-launch steps (plane_kernels.hip.h, whole_step_kernel): may this engine take them (synchronises once per source / receiver
    // set: not inside a capture), and the launch itself
    bool whole_step_ready();
    bool whole_step_sized() const;  // small enough for the form to beat two-step passes too (the automatic choice)
    int launch_whole_step(Real* prev, const Real* cur, int slot, uint64_t signal_pos, bool source_live, bool serve_next);
    // ---- engine_pair.hip.h
    bool pair_eligible();
    int ensure_pair();
    int build_pair_units(int owned);
    static void parallel_sort(std::vector<uint64_t>& v);
    int enqueue_pair_a(int slot, uint64_t signal_pos, bool source_live, bool fuse_mid);
    int launch_fixup(uint32_t first, uint32_t n, const Real* t1, const Real* cur, Real* out2, int* flag2);
    int build_xwall();
    void xwall_args(wv::BoundaryArgs<Real>& b) const;
    int enqueue_pair_b(int slot, uint64_t signal_pos, bool source_live, int fuse_next);
    bool slab_early_now() const;
    int begin_halo_wait_timing();
    int end_halo_wait_timing(int token);
    int begin_part_timing(int part, bool always = false);
    int end_part_timing(int part, int token);
    int enqueue_batch_pair(uint64_t i, int part, int next_kind) override;
    int batch_pair_eligible(int* eligible) override;
    int batch_pair_prepare(int* ready, int* singles_first) override;
    int batch_pair_vetoed() override;
    int batch_triple_prepare(int* ready) override;
    int enqueue_batch_triple(uint64_t i, int part) override;
    uint64_t role_signature() const override {
        return (uint64_t)cur_ | (uint64_t)prv_ << 2 | (uint64_t)spare_[0] << 4 | (uint64_t)spare_[1] << 6 | steps_done << 8;
    }
    // ---- engine_triple.hip.h
    static constexpr int kWideLaneBytes = 16;  // the wider form of the three-step march's lanes (triple_kernels.hip.h)
    int triple_lane_bytes() const;
    bool triple_eligible();
    int ensure_triple();
    int build_triple_units();
    int enqueue_triple(int slot, uint64_t signal_pos, bool source_live, int fuse_next);
    int enqueue_triple_slab(int slot, int part, uint64_t signal_pos, bool source_live);
    int launch_triple_march(int slot, const Real* A, const Real* B, Real* O1, Real* O2, Real* O3);
    int launch_triple_list(int slot, const Real* A, const Real* B, const Real* O1, const Real* O2, Real* O3, bool source_live);
    // ---- engine_batch.hip.h
    bool time_this_launch();
    int drain_timing();
    int step(int32_t* flag) override;
    int swap() override;
    // ---- engine_single.hip.h
    int replay_batch(uint64_t batch, bool source_live, bool can_fuse);
    // ---- engine_batch.hip.h
    uint64_t plan_batch(uint64_t remaining) override;
    int enqueue_batch_step(uint64_t i, uint64_t batch, int next_kind) override;
    int collect_batch(uint64_t batch) override;
    const int* batch_flags() const override { return flags_host_; }
    int commit_batch(uint64_t batch, const int* flags, uint64_t* good_out, int32_t* flag_out) override;
    int run(uint64_t n_steps, uint64_t* done, int32_t* flag_out) override;
    // ---- engine_io.hip.h
    int set_source(int kind, uint64_t node, const double* signal, uint64_t n) override;
    int set_receivers(const uint64_t* nodes, uint32_t n) override;
    bool io_nodes_plain();
    bool io_nodes(std::vector<uint64_t>* stored);
    bool io_nodes_unfaced();
    bool io_nodes_clear_of_x_walls();
    int fetch_receivers(uint64_t first, uint64_t n, double* dst) override;
    Real* buffer(int which) { return which == WV_BUF_CURRENT ? field_[cur_] : field_[prv_]; }
    hipError_t class_of(uint64_t x, uint64_t row, uint32_t* cls);
    uint64_t stored_index(uint64_t node) const;
    int read_value(int buffer_id, uint64_t index, double* v) override;
    int write_value(int buffer_id, uint64_t index, double v) override;
    template <typename Other>
    int copy_field(Real* stored, void* host, bool to_device, int z0, int planes);
    int read_field(int buffer_id, void* dst, int elem_size) override { return read_planes(buffer_id, 0, nz_, dst, elem_size); }
    int write_field(int buffer_id, const void* src, int elem_size) override {
        return write_planes(buffer_id, 0, nz_, src, elem_size);
    }
    int read_planes(int buffer_id, int z0, int planes, void* dst, int elem_size) override;
    int write_planes(int buffer_id, int z0, int planes, const void* src, int elem_size) override;
    int boundary_data(int dim, wv_boundary_data* host, bool to_device) override;
    int set_coefficients(const wv_coefficients_canonical* c, uint32_t n) override;
    int device_buffer(int buffer_id, void** p) override;
    int checkpoint(int op) override;
    // ---- engine_batch.hip.h
    int kernel_time(double* mean_ms, uint64_t* launches, uint64_t* steps) override;
    int synchronize() override;
    int query(int what, uint64_t* value) override;
    // ---- engine_slab.hip.h
    int comm_init(const void* id, int rank, int nranks) override;
    int comm_init_local(int rank, int nranks) override;
    int adopt_comm(std::unique_ptr<wv::SlabComm> c);
    wv::SlabComm* comm() override { return comm_.get(); }
    uint64_t field_pitch() const override { return (uint64_t)pitch_; }
    int comm_destroy() override;

private:
    void release();  // engine_setup.hip.h
    // rooms that leave much of the mesh outside: visit live tiles / units only (wv_options::all_tiles, wv_tuning::tile_lists)
    bool use_work_lists() const { return !opt_.all_tiles && opt_.tuning.tile_lists != 0; }

    wv_options opt_{};
    int nx_ = 0, ny_ = 0, nz_ = 0, z_begin_ = 0, z_end_ = 0, device_ = -1;
    uint64_t n_nodes_ = 0, stored_nodes_ = 0, field_bytes_ = 0;
    int pitch_ = 0;
    Real* field_[4] = {nullptr, nullptr, nullptr, nullptr};
    int cur_ = 1, prv_ = 0, spare_[2] = {2, 3};  // which field_ holds which role
    uint8_t* cls_ = nullptr;
    int cls_pitch_ = 0;
    uint32_t n1_ = 0, n2_ = 0, n3_ = 0, n_entries_ = 0, n_slots_ = 0, n_coeffs_ = 0;
    uint32_t* bnode_ = nullptr;
    uint64_t* tile_list_ = nullptr;   // sweep work list (build_tile_lists), null = arithmetic mapping
    uint32_t list_start_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t list_longest_ = 0;
    bool lists_built_ = false;
    int lists_z0_ = 0, lists_z1_ = 0;  // plane range the lists were built for
    struct GraphKey {
        uint64_t batch;
        int cur;
        bool source_live, can_fuse;
        uint32_t n_recv;
        uint64_t source_node;
        int source_kind;
        uint64_t signal_ptr, recv_ptr;
        bool lists;
        uint64_t cur_ptr, prv_ptr;
        uint64_t io_generation;  // set_source / set_receivers calls so far: the one-launch steps' duty count and legality are baked into the capture
        bool operator==(const GraphKey& o) const {
            return batch == o.batch && cur == o.cur && source_live == o.source_live && can_fuse == o.can_fuse &&
                   n_recv == o.n_recv && source_node == o.source_node && source_kind == o.source_kind &&
                   signal_ptr == o.signal_ptr && recv_ptr == o.recv_ptr && lists == o.lists && cur_ptr == o.cur_ptr &&
                   prv_ptr == o.prv_ptr && io_generation == o.io_generation;
        }
    };
    hipGraphExec_t graph_exec_ = nullptr;
    GraphKey graph_key_{};
    uint64_t* signal_base_dev_ = nullptr;
    bool graph_capturing_ = false;
    uint64_t graph_max_nodes_ = 64ull << 20;
    bool batch_flags_reset_ = false;  // the flag words of the batch being enqueued hold the mesh-static bits already
    bool pre_post_done_ = false;      // this step's pre/post work was done by the previous boundary launch
    // two-step passes
    int pair_inner_ok_ = -1;  // boundary entries finish the inside nodes they face (ensure_pair): -1 not checked yet
    uint64_t pair_min_nodes_ = 4ull << 20;      // stored nodes: between 128^3 (single steps win) and 160^3 (passes win)
    bool pair_failed_ = false;
    uint8_t* pair_map_ = nullptr;
    uint32_t* pair_list_ = nullptr;
    uint32_t* pair_counter_ = nullptr;
    uint32_t* pair_units_ = nullptr;               // march work list (build_pair_units), null = every unit
    uint32_t pair_unit_start_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t pair_units_longest_ = 0;
    bool pair_sparse_ok_ = true;                   // sparse room: the march's live units cost less than the sweep's live tiles
    double tile_active_frac_ = 1.0;
    double pair_live_frac_ = 1.0;                  // live waves of listed units / all waves of all units (build_pair_units)
    uint32_t pair_list_n_ = 0;  // fix-up nodes of the marched planes
    int pair_z0_ = 0, pair_z1_ = 0;                // planes the march produces
    uint64_t pair_source_ = 0;
    int pair_nw_ = 1, pair_strips_ = 0, pair_zc_ = 0, pair_chunks_ = 1;
    uint64_t timed_steps_ = 0;
    bool batch_can_fuse_ = false, batch_source_live_ = false;  // plan_batch's decisions for the batch being enqueued
    bool io_plain_known_ = false, io_plain_ = false;
    bool io_unfaced_known_ = false, io_unfaced_ = false;
    bool io_xclear_known_ = false, io_xclear_ = false;  // io_nodes_clear_of_x_walls
    bool duties_known_ = false, duties_ok_ = false;  // whole_step_ready
    wv::StepDuty* duties_ = nullptr;                  // [n_duties_] the source first, then the recorded receivers
    uint32_t n_duties_ = 0;
    uint64_t whole_steps_ = 0;                        // steps taken as one launch (WV_QUERY_WHOLE_STEPS)
    uint64_t graph_whole_steps_ = 0;                  // ... by one replay of the captured batch
    uint64_t io_generation_ = 0;                      // bumped by set_source / set_receivers (GraphKey)
    bool pair_list_early_ok_ = false;             // ensure_pair
    bool pair_unit_waves_ = false;                // the unit list carries each unit's live waves (build_pair_units)
    int pair_windows_ = 0;                        // WIDE march: workgroups side by side per row (0: one)
    uint8_t pair_win_[4][wv::kPairMaxWindows] = {};  // first wave, waves, first storing wave, end of the storing waves
    bool pair_mid_done_ = false, pair_list_done_ = false;  // part A of the pass in flight has served t+1's source / receivers, the list
    int outside_dirty_ = 0;           // steps until the outside nodes are known to be 0 in both fields again
    uint32_t* ref_to_pos_ = nullptr;  // [n_entries] caller's (class offset + boundary_index) -> processing position
    uint8_t* btype_ = nullptr;
    double* fmem_ = nullptr;
    uint32_t* cidx_ = nullptr;
    // boundary entries by plane (build_plane_order; slab path only); `_rest`: without the first n_xw_ entries
    uint32_t* zorder_ = nullptr;
    uint32_t* zorder_rest_ = nullptr;  // (same allocation as zorder_)
    uint32_t* face_order_ = nullptr;   // (same allocation) the entries of a slab's face plane(s)
    uint32_t face_n_ = 0;
    uint32_t* early_order_ = nullptr;  // (same allocation) the entries of the two planes next to each neighbour
    uint32_t early_n_ = 0;
    // a slab's two-step pass with both exchanges under the march (engine_pair.hip.h): planes whose t+1 the march stores
    int pair_s0_ = 0, pair_s1_ = 0;
    bool pair_early_ = false;          // the pass in flight is one (decided in part A, read by part B)
    // time the compute stream spends waiting for ghost planes (wv_enable_kernel_timing on a slab): event pairs around wait_ghosts
    std::vector<hipEvent_t> halo_events_;
    int halo_ev_used_ = 0;
    unsigned halo_timing_calls_ = 0;
    double halo_wait_ms_ = 0;
    uint64_t halo_wait_n_ = 0, early_passes_ = 0;
    // kernel timing of a two-step pass's two boundary launches (part 0: nodes to t+1, part 1: to t+2), in the passes whose march is timed
    // (three-step passes: [2] boundary nodes to t+3, [3] the third level's fix-up list, [4] the three-step march itself -- kept apart
    // from the two-step march's account, a batch may take both kinds of pass)
    static constexpr int kParts = 5;
    std::vector<hipEvent_t> part_events_[kParts];
    int part_ev_used_[kParts] = {0, 0, 0, 0, 0};
    double part_ms_[kParts] = {0, 0, 0, 0, 0};
    uint64_t part_n_[kParts] = {0, 0, 0, 0, 0};
    bool pass_timed_ = false;
    unsigned part_timing_calls_ = 0;
    std::vector<uint32_t> plane_start_, plane_start_rest_;
    // x-facing walls on compact copies in two-step passes (boundary_kernels.hip.h, xwall_node; engine_pair.hip.h)
    uint32_t n_xw_ = 0;            // the first n_xw_ entries qualify (settled with the entry order in init)
    uint32_t* xw_nbr_ = nullptr;   // [4][n_xw_] in-wall neighbours by entry position
    Real* xw_val_ = nullptr;       // [9][n_xw_]: own value at the odd / even level, faced node, level 1's captures; a three-step pass's third generations and level 2's captures
    uint8_t* xw_gok_ = nullptr;    // [n_xw_]: the entry finishes the node behind the one it faces at a three-step pass's third level (xwall_cover_kernel)
    bool xw_built_ = false;        // table and copies allocated (first ensure_pair that may use them)
    bool xw_active_ = false;       // this (mesh, source) runs its passes on them
    bool xw_valid_ = false;        // the copies hold what the fields hold
    uint64_t passes_taken_ = 0;
    // three-step passes (engine_triple.hip.h)
    Real* field1_ = nullptr;           // t+1 at shell and boundary nodes of the pass in flight, zeros at outside nodes
    uint8_t* triple_map_ = nullptr;
    uint32_t* triple_list_ = nullptr;  // third level's fix-up list: every shell node
    uint32_t triple_list_n_ = 0;
    int* suspect_ = nullptr;           // [kRing] per step slot: the march saw an inf / nan
    uint64_t triple_source_ = 0, triple_io_generation_ = ~0ull;
    bool triple_failed_ = false, triple_ready_ = false, triple_attr_set_ = false;
    uint32_t* triple_units_ = nullptr; // sparse rooms: the three-step march's work list (build_triple_units), XCD k's run at triple_unit_start_[k]
    uint32_t triple_unit_start_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t triple_units_longest_ = 0;
    double triple_live_frac_ = 1.0;    // live waves of listed units / all waves of all units
    bool triple_xw_ = false;           // the passes' three levels take the x-facing walls on their compact copies (and the third-level list leaves the nodes they face out)
    // stored nodes from which the engine takes three-step passes by itself (tools/pass_forms_by_size.py, profiles/r06/pass_forms_by_size_*.txt:
    // Gnode-updates/s two-step / three-step at the end of round 6, fp64: 224^3 187 / 200, 256^3 213 / 243, 320^3 214 / 255, 384^3 267 / 321,
    // 512^3 286 / 362, 768^3 326 / 416, 1024^3 338 / 443; fp32: 384^3 352 / 421, 512^3 487 / 587, 640^3 460 / 484, 768^3 557 / 590,
    // 896^3 536 / 649, 1024^3 626 / 765 -- wherever two-step passes run at all in fp64; in fp32 from the first size measured)
    uint64_t triple_min_nodes_ = sizeof(Real) == 8 ? (12ull << 20) : (24ull << 20);
    int triple_z0_ = 0, triple_z1_ = 0;  // the planes the march produces: the owned ones, less the face plane and the plane next to it where a neighbour follows
    int triple_nw_ = 1, triple_strips_ = 0, triple_zc_ = 0, triple_chunks_ = 1, triple_windows_ = 0;
    uint8_t triple_win_[4][wv::kTripleMaxWindows] = {};
    int triple_lb_ = 8;                // bytes of a row per lane of the march as set up (triple_lane_bytes)
    // stored row length (elements) from which doubles march on 16-byte lanes (profiles/r06/lane_width_by_size.txt: Gnode-updates/s with
    // 8- / 16-byte lanes 256^3 234 / 226, 320^3 220 / 234, 384^3 281 / 309, 512^3 346 / 340, 768^3 360 / 378, 1024^3 343 / 403)
    int triple_wide_from_ = 320;
    uint64_t triples_taken_ = 0;
    int* status_ = nullptr;
    int* static_flag_dev_ = nullptr;
    int static_flag_ = 0;
    double* coeffs_ = nullptr;
    int* flags_ = nullptr;
    int* flags_host_ = nullptr;
    void* scratch_ = nullptr;
    Real courant_ = 0, courant_sq_ = 0;
    hipStream_t stream_ = nullptr, comm_stream_ = nullptr;
    hipStream_t on_ = nullptr;  // the launch helpers' stream when it is not the compute stream (a slab's face work on its halo stream)
    hipStream_t st() const { return on_ ? on_ : stream_; }
    StreamPlan plan_;
    int tune_variant_ = -1, tune_ry_ = 0, tune_nwx_ = 0, tune_nwy_ = 0, tune_zchunks_ = 0;
    std::vector<hipEvent_t> events_;
    unsigned timing_launches_ = 0;
    int ev_used_ = 0;
    double time_ms_ = 0;
    uint64_t time_n_ = 0;
    // source / receivers
    int source_kind_ = WV_SOURCE_NONE;
    uint64_t source_node_ = 0, signal_len_ = 0, signal_pos_ = 0;
    double* signal_ = nullptr;
    uint64_t* recv_nodes_ = nullptr;
    Real* recv_out_ = nullptr;
    uint32_t n_recv_ = 0;
    uint64_t recv_first_step_ = 0;
    Real* recv_stage_ = nullptr;  // pinned, kRing rows: a copy into pageable memory would make hipMemcpyAsync wait for the stream on the host
    std::vector<double> recv_log_;
    std::unique_ptr<wv::SlabComm> comm_;
    // wv_checkpoint / wv_rollback (engine_io.hip.h): device copies of the two live fields and the filter memories, and the
    // host-side position that goes with them
    struct Checkpoint {
        Real* field[2] = {nullptr, nullptr};  // [0] current, [1] previous at the time of the save
        double* fmem = nullptr;
        bool valid = false;
        uint64_t steps_done = 0, signal_pos = 0, recv_first_step = 0;
        size_t recv_log_size = 0;
        uint32_t n_recv = 0;
        int outside_dirty = 0;
    } ckpt_;
};

}  // namespace wv
