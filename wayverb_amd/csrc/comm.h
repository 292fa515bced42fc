// comm.h -- z-slab ghost-plane exchange over RCCL (xGMI), one rank per GPU.
//
// New design (the reference is single-device).  Memory order is x-fastest,
// z-slowest (src/waveguide/src/cl/utils.cpp:33-36), so a slab's face plane and a ghost plane are
// each one contiguous nx*ny run.  Per step, after the two face planes of the new field are
// final, each rank sends them to its z-1 / z+1 neighbours' ghost planes in one grouped
// ncclSend/ncclRecv on a dedicated stream while the interior planes are still being updated on
// the compute stream.  Slab chain = nearest neighbour only: at most 2 of a GPU's 7 xGMI links.
//
// RCCL is resolved at run time (dlopen) so that a process that already loaded a librccl
// (e.g. through torch.distributed) shares that copy.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace wv {

class SlabComm {
public:
    SlabComm() = default;
    ~SlabComm();
    SlabComm(const SlabComm&) = delete;
    SlabComm& operator=(const SlabComm&) = delete;

    static bool unique_id(void* bytes128, std::string* err);

    bool init(const void* id_bytes128, int rank, int nranks, int device, hipStream_t comm_stream,
              bool has_lo, bool has_hi, std::string* err);

    // The compute stream must not read ghost planes before the previous exchange has landed.
    bool wait_ghosts(hipStream_t compute, std::string* err);
    // Faces of `field` (planes 1 and nz-2 when the matching ghost exists) are final on `compute`:
    // exchange them into the neighbours' ghost planes (planes nz-1 / 0 over there).
    // `also` (may be null): a second event the exchange has to wait for (boundary-node stream).
    bool exchange_faces(hipStream_t compute, hipEvent_t also, void* field, size_t elem_size, int nx, int ny, int nz,
                        std::string* err);

    int rank() const { return rank_; }
    int nranks() const { return nranks_; }

private:
    void* comm_ = nullptr;
    int rank_ = 0, nranks_ = 1;
    bool has_lo_ = false, has_hi_ = false, loopback_ = false;
    hipStream_t stream_ = nullptr;
    hipEvent_t faces_ready_ = nullptr;
    hipEvent_t ghosts_ready_ = nullptr;
    bool pending_ = false;
};

}  // namespace wv
