// engine_batch.hip.h -- wv_step / wv_swap / wv_run: batches of steps, the flag protocol of waveguide.h:82-119, kernel timing.
//
// Part of the engine behind the C ABI of include/wayverb_amd.h (engine.hip is the translation unit; see engine.hip.h for
// the class and the map of which file holds what).
#pragma once
#include "engine.hip.h"

namespace wv {

// Kernel timing (wv_enable_kernel_timing): a pair of events around the dominant kernel.  The two records cost
// about 11 us of stream time (measured at 256^3: 6 % of a pass; the launches without them follow each other
// within a microsecond), so below 512^3 only every eighth launch is timed.
template <typename Real>
bool Engine<Real>::time_this_launch() {
    if (!timing || ev_used_ + 2 > (int)events_.size()) return false;
    const unsigned stride = stored_nodes_ < (128ull << 20) ? 8u : 1u;
    return (timing_launches_++ % stride) == 0;
}

template <typename Real>
int Engine<Real>::drain_timing() {
    for (int i = 0; i + 1 < ev_used_; i += 2) {
        float ms = 0;
        WV_HIP(hipEventElapsedTime(&ms, events_[i], events_[i + 1]));
        time_ms_ += ms;
        ++time_n_;
    }
    ev_used_ = 0;
    for (int i = 0; i + 1 < halo_ev_used_; i += 2) {
        float ms = 0;
        WV_HIP(hipEventElapsedTime(&ms, halo_events_[i], halo_events_[i + 1]));
        halo_wait_ms_ += ms;
        ++halo_wait_n_;
    }
    halo_ev_used_ = 0;
    for (int p = 0; p < kParts; ++p) {
        for (int i = 0; i + 1 < part_ev_used_[p]; i += 2) {
            float ms = 0;
            WV_HIP(hipEventElapsedTime(&ms, part_events_[p][i], part_events_[p][i + 1]));
            part_ms_[p] += ms;
            ++part_n_[p];
        }
        part_ev_used_[p] = 0;
    }
    return WV_OK;
}

// Kernel timing of the boundary launches of a two-step pass whose march is timed (bench.py's roofline.boundary: what stands between
// the dominant kernel's rate and the whole step's).
template <typename Real>
int Engine<Real>::begin_part_timing(int part, bool always) {
    if (!timing || !(pass_timed_ || always)) return -1;
    auto& ev = part_events_[part];
    if (ev.empty()) {
        ev.resize(2 * 512);
        for (auto& e : ev)
            if (hipEventCreate(&e) != hipSuccess) {
                (void)hipGetLastError();
                e = nullptr;
            }
    }
    const int at = part_ev_used_[part];
    if (at + 2 > (int)ev.size() || !ev[at] || !ev[at + 1]) return -1;
    if (hipEventRecord(ev[at], stream_) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return at;
}

template <typename Real>
int Engine<Real>::end_part_timing(int part, int token) {
    if (token < 0) return WV_OK;
    WV_HIP(hipEventRecord(part_events_[part][token + 1], stream_));
    part_ev_used_[part] = token + 2;
    return WV_OK;
}

// Kernel timing on a slab: how long the compute stream stands at "ghost planes in place" -- the part of the halo exchange that
// the interior work did not hide -- between a pair of events around every fourth such wait (the pair itself costs stream time).
template <typename Real>
int Engine<Real>::begin_halo_wait_timing() {
    if (!timing || !comm_ || (halo_timing_calls_++ & 3u) != 0) return -1;
    if (halo_events_.empty()) {
        halo_events_.resize(2 * 160);
        for (auto& e : halo_events_)
            if (hipEventCreate(&e) != hipSuccess) {
                (void)hipGetLastError();
                e = nullptr;
            }
    }
    if (halo_ev_used_ + 2 > (int)halo_events_.size() || !halo_events_[halo_ev_used_] || !halo_events_[halo_ev_used_ + 1]) return -1;
    if (hipEventRecord(halo_events_[halo_ev_used_], stream_) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return halo_ev_used_;
}

template <typename Real>
int Engine<Real>::end_halo_wait_timing(int token) {
    if (token < 0) return WV_OK;
    WV_HIP(hipEventRecord(halo_events_[token + 1], stream_));
    halo_ev_used_ = token + 2;
    return WV_OK;
}

// -------------------------------------------------------------------------------------------
template <typename Real>
int Engine<Real>::step(int32_t* flag) {
    DeviceGuard guard(device_);
    pre_post_done_ = false;  // (a batch that failed while being enqueued may have left it set)
    batch_flags_reset_ = false;
    int rc = enqueue_step(0, false, 0, false);
    if (rc) return rc;
    WV_HIP(hipMemcpyAsync(flags_host_, flags_, sizeof(int), hipMemcpyDeviceToHost, stream_));
    WV_HIP(hipStreamSynchronize(stream_));
    if ((rc = drain_timing())) return rc;
    if (flag) *flag = flags_host_[0];
    return WV_OK;
}

template <typename Real>
int Engine<Real>::swap() {
    std::swap(cur_, prv_);
    xw_valid_ = false;
    ++steps_done;
    // a step driven from outside (wv_step / wv_swap) records no receiver samples: its row of the log is NaN,
    // so that wv_fetch_receivers keeps addressing rows by step
    if (n_recv_) recv_log_.insert(recv_log_.end(), n_recv_, std::numeric_limits<double>::quiet_NaN());
    return WV_OK;
}

// ---- a batch of steps: plan / enqueue / collect / commit ---------------------------------------
// How many of `remaining` steps the next batch may take (0: the source signal is exhausted, which
// ends the run -- hard_source.h:18-20 returns false).
template <typename Real>
uint64_t Engine<Real>::plan_batch(uint64_t remaining) {
    DeviceGuard guard(device_);
    const uint64_t interval = opt_.flag_interval > 0 ? (uint64_t)opt_.flag_interval : (uint64_t)kRing;
    uint64_t batch = std::min<uint64_t>(std::min<uint64_t>(interval, kRing), remaining);
    if (source_kind_ != WV_SOURCE_NONE) {
        const uint64_t left = signal_len_ - std::min(signal_len_, signal_pos_);
        batch = std::min(batch, left);
    }
    batch_can_fuse_ = !comm_ && io_nodes_plain() && opt_.tuning.fuse_pre_post != 0;
    (void)whole_step_ready();  // (looks at the class map once per source / receiver set: here, not inside a capture)
    batch_source_live_ = source_kind_ != WV_SOURCE_NONE;
    // nothing rides across batches: whatever a batch that failed half-way left behind does not count
    pre_post_done_ = pair_mid_done_ = pair_list_done_ = false;
    // A slab resets the flag words of the whole batch here (waveguide.h:82 does it per step; the source / receiver
    // launch does it elsewhere) -- most slabs of a chain hold neither source nor receivers and then have no such launch.
    batch_flags_reset_ = false;
    if (comm_ && batch) {
        static_assert(sizeof(int) == 4, "flag words are 32 bit");
        if (hipMemsetD32Async((hipDeviceptr_t)flags_, static_flag_, (size_t)batch, stream_) == hipSuccess)
            batch_flags_reset_ = true;
        else
            (void)hipGetLastError();  // (the steps then reset their flag words one by one; nothing sticky is left behind)
    }
    return batch;
}

// `next_kind`: what step i + 1 of the batch starts (its source / receiver work may ride in this step's
// boundary launch)
template <typename Real>
int Engine<Real>::enqueue_batch_step(uint64_t i, uint64_t batch, int next_kind) {
    DeviceGuard guard(device_);
    const int rc = enqueue_step((int)i, true, signal_pos_ + i, batch_source_live_,
                                batch_can_fuse_ && i + 1 < batch ? next_kind : 0);
    if (rc) return rc;
    std::swap(cur_, prv_);
    return WV_OK;
}

// Flag words and receiver rows of the batch to the host.  On an RCCL slab chain the flag words are
// OR-ed over the ranks first, so that every rank sees the same first failing step and none is left
// waiting in a receive (waveguide.h:100-119 stops the one and only device; here all of them stop).
template <typename Real>
int Engine<Real>::collect_batch(uint64_t batch) {
    DeviceGuard guard(device_);
    std::string cerr;
    // a slab's faces may have been stepped on its halo stream, which raises flag bits like any other launch
    if (comm_ && !comm_->join_halo(stream_, &cerr)) return fail(WV_E_COMM, cerr);
    if (comm_ && !comm_->or_flags(stream_, flags_, (int)batch, &cerr)) return fail(WV_E_COMM, cerr);
    WV_HIP(hipMemcpyAsync(flags_host_, flags_, batch * sizeof(int), hipMemcpyDeviceToHost, stream_));
    if (n_recv_) {
        WV_HIP(hipMemcpyAsync(recv_stage_, recv_out_, (size_t)batch * n_recv_ * sizeof(Real), hipMemcpyDeviceToHost, stream_));
    }
    if (comm_ && !comm_->is_local()) {
        // (a peer that died leaves this wait on the device for good: give up after wv_options::comm_timeout_s and say who waited)
        if (!comm_->sync(stream_, "the batch of steps " + std::to_string(steps_done) + " .. " + std::to_string(steps_done + batch - 1), &cerr))
            return fail(WV_E_COMM, cerr);
    } else {
        WV_HIP(hipStreamSynchronize(stream_));
    }
    // in-process chain: the last step's face pushes run on the halo streams; wv_run_group collects every slab, so that on
    // its return no copy into anybody's ghost plane is still in flight (a read-back or a wv_destroy may follow)
    if (comm_ && comm_->is_local()) WV_HIP(hipStreamSynchronize(comm_stream_));
    return drain_timing();
}

// `flags[batch]`: this engine's flag words, or their OR over a group of slabs.
template <typename Real>
int Engine<Real>::commit_batch(uint64_t batch, const int* flags, uint64_t* good_out, int32_t* flag_out) {
    uint64_t good = batch;
    int32_t flag = 0;
    for (uint64_t i = 0; i < batch; ++i) {
        if (flags[i]) {
            good = i;
            flag = flags[i];
            break;
        }
    }
    if (n_recv_)
        for (size_t i = 0; i < (size_t)good * n_recv_; ++i) recv_log_.push_back((double)recv_stage_[i]);
    steps_done += good;
    signal_pos_ += good;
    // fields have advanced past a failing step: like the reference after its throw, the state is
    // no longer meaningful; keep the buffer roles consistent with `good` swaps
    if (flag && good < batch && ((batch - good) & 1)) std::swap(cur_, prv_);
    if (flag) xw_valid_ = false;
    *good_out = good;
    *flag_out = flag;
    return WV_OK;
}

template <typename Real>
int Engine<Real>::run(uint64_t n_steps, uint64_t* done, int32_t* flag_out) {
    DeviceGuard guard(device_);
    if (comm_ && comm_->is_local() && comm_->nranks() > 1)
        return fail(WV_E_STATE, "slabs joined by wv_comm_init_local are stepped together: use wv_run_group");
    uint64_t completed = 0;
    int32_t flag = 0;
    const bool chain = comm_ && !comm_->is_local();  // (a one-rank communicator agrees with itself: the loopback test runs this)
    std::string cerr;
    while (completed < n_steps && flag == 0) {
        uint64_t batch = plan_batch(n_steps - completed);
        int eligible = 0, rc = batch_pair_eligible(&eligible);
        if (rc) return rc;
        if (chain) {
            // Every rank of the chain has to enqueue the same steps in the same form, or its exchanges and the flag
            // all-reduce would not pair up with its neighbours': the batch is the shortest any rank allows (only the
            // ranks that hold the source plane know where the signal ends -- hard_source.h:18-20 ends the run there, for
            // all of them), and two-step passes need every rank's consent.
            uint64_t words[2] = {batch, (uint64_t)eligible};
            if (!comm_->agree_min(stream_, words, 2, &cerr)) return fail(WV_E_COMM, cerr);
            batch = words[0];
            eligible = (int)words[1];
        }
        if (batch == 0) break;
        // Small meshes are bound by launches, not bytes: a full batch of steps is captured once
        // into a hipGraph and replayed (the only thing that differs between batches, the
        // position in the source signal, comes from a device scalar).  Even batch lengths only,
        // so that the two fields are back in their roles after every replay.
        const bool use_graph = opt_.tuning.graph != 0 && !comm_ && !timing && (batch % 2) == 0 && batch >= 16 &&
                               stored_nodes_ <= graph_max_nodes_ && outside_dirty_ == 0;
        if (use_graph) {
            if ((rc = replay_batch(batch, batch_source_live_, batch_can_fuse_))) return rc;
        } else {
            // big meshes: two steps per pass over the fields wherever a batch has two left
            int singles_first = -1;
            if (eligible) {
                int ready = 0, mine = 0;
                // (no memory for four fields / the map / the lists is not an error: *ready = 0 and the chain stays with single steps)
                const int prepared = batch_pair_prepare(&ready, &mine);
                if (prepared && !chain) return prepared;
                if (chain) {
                    // A rank that FAILED here (a device fault, not a lack of memory) must not simply return: the others are about
                    // to enter this all-reduce and would wait there for good.  It goes in with them and says so (third word);
                    // every rank then returns an error from the same point, none of them left inside an exchange.
                    // min of (2 - singles) = the most single sweeps any rank needs first
                    uint64_t words[3] = {(uint64_t)(prepared ? 0 : ready), (uint64_t)(prepared ? 2 : 2 - mine), prepared ? 0ull : 1ull};
                    if (!comm_->agree_min(stream_, words, 3, &cerr)) return fail(WV_E_COMM, cerr);
                    if (prepared) return prepared;  // (wv_last_error still says what happened here)
                    if (words[2] == 0)
                        return fail(WV_E_COMM, "another rank of the chain failed while preparing two-step passes (its wv_last_error says why)");
                    ready = (int)words[0];
                    mine = 2 - (int)words[1];
                }
                if (ready) singles_first = mine;
            }
            if (singles_first < 0 && (rc = batch_pair_vetoed())) return rc;
            const bool pairs = singles_first >= 0;
            // ... and three steps per pass wherever it has three left (one domain, a room that fills its mesh: engine_triple.hip.h).
            // The passes do not reset flag words one by one: the whole batch's are set here.
            bool triples = false;
            if (pairs && batch >= (uint64_t)singles_first + 3) {
                int mine = 0;
                const int prepared = batch_triple_prepare(&mine);
                if (prepared && !chain) return prepared;
                if (chain) {  // (every rank's consent, and a rank that failed goes into the all-reduce with the others: as above)
                    uint64_t words[2] = {(uint64_t)(prepared ? 0 : mine), prepared ? 0ull : 1ull};
                    if (!comm_->agree_min(stream_, words, 2, &cerr)) return fail(WV_E_COMM, cerr);
                    if (prepared) return prepared;
                    if (words[1] == 0)
                        return fail(WV_E_COMM, "another rank of the chain failed while preparing three-step passes (its wv_last_error says why)");
                    mine = (int)words[0];
                }
                triples = mine != 0;
                if (triples && !batch_flags_reset_) {
                    WV_HIP(hipMemsetD32Async((hipDeviceptr_t)flags_, static_flag_, (size_t)batch, stream_));
                    batch_flags_reset_ = true;
                }
            }
            auto triple_at = [&](uint64_t i) { return triples && i >= (uint64_t)singles_first && i + 3 <= batch; };
            auto pair_at = [&](uint64_t i) { return pairs && !triple_at(i) && i >= (uint64_t)singles_first && i + 2 <= batch; };
            // (a three-step pass serves its own first step's source / receiver work unless the launch before it has: as a single step would)
            auto kind_at = [&](uint64_t i) { return i >= batch ? 0 : (pair_at(i) ? 2 : 1); };
            for (uint64_t i = 0; i < batch;) {
                if (triple_at(i)) {
                    if (comm_) {
                        for (int part = 0; part < 3; ++part)
                            if ((rc = enqueue_batch_triple(i, part))) return rc;
                    } else if ((rc = enqueue_triple((int)i, signal_pos_ + i, batch_source_live_, kind_at(i + 3)))) {
                        return rc;
                    }
                    i += 3;
                } else if (pair_at(i)) {
                    if ((rc = enqueue_batch_pair(i, 0, 0))) return rc;
                    if ((rc = enqueue_batch_pair(i, 1, kind_at(i + 2)))) return rc;
                    i += 2;
                } else {
                    if ((rc = enqueue_batch_step(i, batch, kind_at(i + 1)))) return rc;
                    i += 1;
                }
            }
        }
        if ((rc = collect_batch(batch))) return rc;
        uint64_t good = 0;
        if ((rc = commit_batch(batch, flags_host_, &good, &flag))) return rc;
        completed += good;
    }
    if (done) *done = completed;
    if (flag_out) *flag_out = flag;
    return WV_OK;
}

template <typename Real>
int Engine<Real>::kernel_time(double* mean_ms, uint64_t* launches, uint64_t* steps) {
    DeviceGuard guard(device_);
    if (mean_ms) *mean_ms = time_n_ ? time_ms_ / (double)time_n_ : 0.0;
    if (launches) *launches = time_n_;
    if (steps) *steps = timed_steps_;
    time_ms_ = 0;
    time_n_ = 0;
    timed_steps_ = 0;
    timing_launches_ = 0;  // the next launch is timed again
    halo_wait_ms_ = 0;
    halo_wait_n_ = 0;
    halo_timing_calls_ = 0;
    for (int p = 0; p < kParts; ++p) {
        part_ms_[p] = 0;
        part_n_[p] = 0;
    }
    part_timing_calls_ = 0;
    return WV_OK;
}

template <typename Real>
int Engine<Real>::query(int what, uint64_t* value) {
    if (!value) return fail(WV_E_INVALID_ARGUMENT, "null argument");
    switch (what) {
        case WV_QUERY_PASSES: *value = passes_taken_; return WV_OK;
        case WV_QUERY_XWALL_ENTRIES: *value = xw_active_ ? n_xw_ : 0; return WV_OK;
        case WV_QUERY_FIELDS: {
            uint64_t n = 0;
            for (int i = 0; i < 4; ++i) n += field_[i] != nullptr;
            n += field1_ != nullptr;
            *value = n;
            return WV_OK;
        }
        // (the march that runs: a sparse room's three-step passes have a work list of their own, narrower waves and all)
        case WV_QUERY_MARCH_LIVE_PERMILLE:
            *value = (triple_units_ && triple_ready_) ? (uint64_t)(triple_live_frac_ * 1000.0 + 0.5) : pair_units_ ? (uint64_t)(pair_live_frac_ * 1000.0 + 0.5) : 1000;
            return WV_OK;
        case WV_QUERY_MARCH_ROUNDS: {
            if (!pair_map_ || pair_nw_ < 1) {
                *value = 0;
                return WV_OK;
            }
            const uint64_t slots = 256ull * (uint64_t)std::max(1, wv::kPairMaxWaves / pair_nw_);
            const uint64_t wgs = pair_units_ ? 8ull * pair_units_longest_
                                             : 8ull * (uint64_t)((pair_strips_ + 7) / 8) * (uint64_t)pair_chunks_ * (uint64_t)std::max(1, pair_windows_);
            *value = (wgs + slots - 1) / slots;
            return WV_OK;
        }
        case WV_QUERY_SWEEP_LIVE_PERMILLE: *value = tile_list_ ? (uint64_t)(tile_active_frac_ * 1000.0 + 0.5) : 1000; return WV_OK;
        case WV_QUERY_HALO_WAIT_NS: *value = (uint64_t)(halo_wait_ms_ * 1e6 + 0.5); return WV_OK;
        case WV_QUERY_HALO_WAITS: *value = halo_wait_n_; return WV_OK;
        case WV_QUERY_HALO_EXCHANGES: *value = comm_ ? comm_->exchanges() : 0; return WV_OK;
        case WV_QUERY_HALO_BYTES_SENT: *value = comm_ ? comm_->planes_sent() * (uint64_t)comm_->plane_bytes() : 0; return WV_OK;
        case WV_QUERY_EARLY_PASSES: *value = early_passes_; return WV_OK;
        case WV_QUERY_BOUNDARY1_NS: *value = (uint64_t)(part_ms_[0] * 1e6 + 0.5); return WV_OK;
        case WV_QUERY_BOUNDARY2_NS: *value = (uint64_t)(part_ms_[1] * 1e6 + 0.5); return WV_OK;
        case WV_QUERY_BOUNDARY_TIMED: *value = std::min(part_n_[0], part_n_[1]); return WV_OK;
        case WV_QUERY_WHOLE_STEPS: *value = whole_steps_; return WV_OK;
        case WV_QUERY_TRIPLE_PASSES: *value = triples_taken_; return WV_OK;
        case WV_QUERY_BOUNDARY3_NS: *value = (uint64_t)(part_ms_[2] * 1e6 + 0.5); return WV_OK;
        case WV_QUERY_FIXUP3_NS: *value = (uint64_t)(part_ms_[3] * 1e6 + 0.5); return WV_OK;
        case WV_QUERY_TRIPLE_PARTS_TIMED: *value = std::min(part_n_[2], part_n_[3]); return WV_OK;
        case WV_QUERY_TRIPLE_MARCH_NS: *value = (uint64_t)(part_ms_[4] * 1e6 + 0.5); return WV_OK;
        case WV_QUERY_TRIPLE_MARCH_TIMED: *value = part_n_[4]; return WV_OK;
        default: return fail(WV_E_INVALID_ARGUMENT, "unknown query");
    }
}

template <typename Real>
int Engine<Real>::synchronize() {
    DeviceGuard guard(device_);
    WV_HIP(hipStreamSynchronize(stream_));
    WV_HIP(hipStreamSynchronize(comm_stream_));
    return WV_OK;
}

}  // namespace wv
