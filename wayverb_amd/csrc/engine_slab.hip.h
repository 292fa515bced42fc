// engine_slab.hip.h -- z-slab chains: joining a communicator, and a chain inside one process stepped in lockstep.
//
// Part of the engine behind the C ABI of include/wayverb_amd.h (engine.hip is the translation unit; see engine.hip.h for
// the class and the map of which file holds what).
#pragma once
#include "engine.hip.h"

namespace wv {

template <typename Real>
int Engine<Real>::comm_init(const void* id, int rank, int nranks) {
    DeviceGuard guard(device_);
    if (comm_) return fail(WV_E_STATE, "communicator already initialised");
    std::unique_ptr<wv::SlabComm> c(new wv::SlabComm());
    std::string err;
    if (!c->init(id, rank, nranks, device_, comm_stream_, opt_.ghost_lo != 0, opt_.ghost_hi != 0, &err))
        return fail(WV_E_COMM, err);
    c->set_timeout(opt_.comm_timeout_s == 0 ? 180.0 : (double)opt_.comm_timeout_s);
    if (opt_.transport == WV_TRANSPORT_IPC) {
        // the neighbours map this rank's fields once, now: the two fields a two-step pass writes have to exist already
        // (no room for them: this rank says "no" whenever the chain asks about passes, and it stays with single steps)
        if (opt_.tuning.pair != 0 && !pair_failed_)
            for (int i = 0; i < 2; ++i) {
                Real*& f = field_[spare_[i]];
                if (f) continue;
                if (hipMalloc((void**)&f, field_bytes_ + 256) != hipSuccess) {
                    (void)hipGetLastError();
                    f = nullptr;
                    pair_failed_ = true;
                    break;
                }
                WV_HIP(hipMemsetAsync(f, 0, field_bytes_ + 256, stream_));
            }
        if (pair_failed_)
            for (int i = 0; i < 2; ++i) {
                Real*& f = field_[spare_[i]];
                if (f) (void)hipFree(f);
                f = nullptr;
            }
        WV_HIP(hipStreamSynchronize(stream_));
        void* fields[4] = {field_[0], field_[1], field_[2], field_[3]};
        c->set_fields(fields, 4, (size_t)pitch_ * ny_ * sizeof(Real), nz_);
        if (!c->init_ipc(&err)) return fail(WV_E_COMM, err);
        comm_ = std::move(c);
        return WV_OK;
    }
    return adopt_comm(std::move(c));
}

template <typename Real>
int Engine<Real>::comm_init_local(int rank, int nranks) {
    DeviceGuard guard(device_);
    if (comm_) return fail(WV_E_STATE, "communicator already initialised");
    std::unique_ptr<wv::SlabComm> c(new wv::SlabComm());
    std::string err;
    if (!c->init_local(rank, nranks, device_, comm_stream_, opt_.ghost_lo != 0, opt_.ghost_hi != 0, &err))
        return fail(WV_E_COMM, err);
    return adopt_comm(std::move(c));
}

template <typename Real>
int Engine<Real>::adopt_comm(std::unique_ptr<wv::SlabComm> c) {
    void* fields[4] = {field_[0], field_[1], field_[2], field_[3]};
    c->set_fields(fields, 4, (size_t)pitch_ * ny_ * sizeof(Real), nz_);
    comm_ = std::move(c);
    return WV_OK;
}

template <typename Real>
int Engine<Real>::comm_destroy() {
    DeviceGuard guard(device_);
    if (comm_ && comm_->dead()) return fail(WV_E_COMM, "the communicator was aborted after a time-out: the engine is good for wv_destroy only");
    comm_.reset();
    return WV_OK;
}

// ---- a chain inside ONE process (wv_comm_init_local / wv_run_group): the slabs are engines driven by one host thread ----
// Same step code as a rank of the RCCL chain; what the ranks of that chain settle by all-reduce (batch length, stepping
// form, flag words) is settled here by looking at every engine.

// engines[r] is slab r: communicators with the in-process transport, neighbours linked
inline int group_init_local(wv_engine* const* engines, int32_t n) {
    if (!engines || n < 1) return fail(WV_E_INVALID_ARGUMENT, "no engines");
    for (int i = 0; i < n; ++i)
        if (!engines[i]) return fail(WV_E_INVALID_ARGUMENT, "null engine");
    for (int i = 0; i < n; ++i) {
        const int rc = engines[i]->comm_init_local(i, n);
        if (rc != WV_OK) {
            for (int k = 0; k < i; ++k) (void)engines[k]->comm_destroy();
            return rc;
        }
    }
    for (int i = 0; i < n; ++i)
        engines[i]->comm()->link_local(i > 0 ? engines[i - 1]->comm() : nullptr, i + 1 < n ? engines[i + 1]->comm() : nullptr);
    return WV_OK;
}

// wv_run on the whole chain, in lockstep: step i of every slab is enqueued before step i + 1 of any, and the two parts of
// a two-step pass likewise -- that is what makes every event a slab waits for refer to a record already enqueued (comm.h)
inline int group_run(wv_engine* const* engines, int32_t n, uint64_t n_steps, uint64_t* steps_done, int32_t* flag_out) {
    if (!engines || n < 1) return fail(WV_E_INVALID_ARGUMENT, "no engines");
    for (int i = 0; i < n; ++i)
        if (!engines[i]) return fail(WV_E_INVALID_ARGUMENT, "null engine");
    // lockstep needs the slabs in the same state: the same field buffer in the same role after the same number of steps
    // (exchanges address the neighbour's buffer by index)
    for (int k = 1; k < n; ++k)
        if (engines[k]->role_signature() != engines[0]->role_signature())
            return fail(WV_E_STATE, "the slabs of a group must have taken the same steps (wv_step / wv_swap / wv_run on one of them alone?)");
    uint64_t completed = 0;
    int32_t flag = 0;
    std::vector<int> ored;
    while (completed < n_steps && flag == 0) {
        // the shortest batch any slab allows (the slab that holds the source knows when it ends)
        uint64_t batch = n_steps - completed;
        for (int k = 0; k < n; ++k) batch = std::min(batch, engines[k]->plan_batch(n_steps - completed));
        if (batch == 0) break;
        // two-step passes only if every slab can take them (asked before anything is allocated for them), after the
        // single steps any of them needs first
        int singles_first = -1, all_eligible = 1;
        for (int k = 0; k < n && all_eligible; ++k) {
            int mine = 0;
            const int rc = engines[k]->batch_pair_eligible(&mine);
            if (rc) return rc;
            all_eligible = mine;
        }
        if (all_eligible) {
            singles_first = 0;
            for (int k = 0; k < n; ++k) {
                int ready = 0, mine = 0;
                const int rc = engines[k]->batch_pair_prepare(&ready, &mine);
                if (rc) return rc;
                if (!ready) {
                    singles_first = -1;
                    break;
                }
                singles_first = std::max(singles_first, mine);
            }
        }
        if (singles_first < 0)
            for (int k = 0; k < n; ++k) {
                const int rc = engines[k]->batch_pair_vetoed();
                if (rc) return rc;
            }
        const bool pairs = singles_first >= 0;
        // ... and three-step passes wherever the batch has three steps left, if every slab can take those
        bool triples = pairs && batch >= (uint64_t)singles_first + 3;
        for (int k = 0; k < n && triples; ++k) {
            int ready = 0;
            const int rc = engines[k]->batch_triple_prepare(&ready);
            if (rc) return rc;
            triples = ready != 0;
        }
        // lockstep: step i of every slab is enqueued before step i + 1 of any, and the parts of a
        // pass likewise (comm.h, local transport)
        auto triple_at = [&](uint64_t i) { return triples && i >= (uint64_t)singles_first && i + 3 <= batch; };
        auto pair_at = [&](uint64_t i) { return pairs && !triple_at(i) && i >= (uint64_t)singles_first && i + 2 <= batch; };
        for (uint64_t i = 0; i < batch;) {
            if (triple_at(i)) {
                for (int part = 0; part < 3; ++part)
                    for (int k = 0; k < n; ++k) {
                        const int rc = engines[k]->enqueue_batch_triple(i, part);
                        if (rc) return rc;
                    }
                i += 3;
            } else if (pair_at(i)) {
                for (int part = 0; part < 2; ++part)
                    for (int k = 0; k < n; ++k) {
                        const int rc = engines[k]->enqueue_batch_pair(i, part, 0);
                        if (rc) return rc;
                    }
                i += 2;
            } else {
                for (int k = 0; k < n; ++k) {
                    const int rc = engines[k]->enqueue_batch_step(i, batch, 0);
                    if (rc) return rc;
                }
                i += 1;
            }
        }
        ored.assign((size_t)batch, 0);
        for (int k = 0; k < n; ++k) {
            const int rc = engines[k]->collect_batch(batch);
            if (rc) return rc;
            const int* f = engines[k]->batch_flags();
            for (uint64_t i = 0; i < batch; ++i) ored[(size_t)i] |= f[i];
        }
        uint64_t good = 0;
        for (int k = 0; k < n; ++k) {
            const int rc = engines[k]->commit_batch(batch, ored.data(), &good, &flag);
            if (rc) return rc;
        }
        completed += good;
    }
    if (steps_done) *steps_done = completed;
    if (flag_out) *flag_out = flag;
    return WV_OK;
}

}  // namespace wv
