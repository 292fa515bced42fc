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
    comm_.reset();
    return WV_OK;
}

}  // namespace wv
