// wayverb_amd/cl_mirror.h -- the pressure field as a real cl::Buffer, for callers pinned to OpenCL types.
//
// src/combined type-erases its waveguide pressure callback as
//     std::function<void(cl::CommandQueue& queue, const cl::Buffer& buffer, size_t step, size_t steps)>
// (src/combined/include/combined/waveguide_base.h:47-59) and reads the field with
// core::read_from_buffer<float>(queue, buffer) when a visualiser is attached
// (src/combined/src/engine.cpp:158-169).  Including this header after the OpenCL C++ bindings
// (CL/cl.hpp, as src/core/include/core/cl/include.h does) makes `waveguide::canonical` serve such a
// callback whenever the compute context it is given carries an OpenCL context (the reference's
// core::compute_context: cl::Context context; cl::Device device -- core/cl/common.h:13-22):
// a float mirror of the step's pressure field lives in a cl::Buffer of that context and is refreshed
// (HIP field -> host -> cl::Buffer) before each call.
//
// That refresh moves the whole field every step -- what the reference's own loop avoids only because
// its field already is a cl::Buffer.  `cl_mirror_wanted()` lets the owner of the callback say when a
// reader is actually attached (one line in combined::engine: `cl_mirror_wanted() = [&] { return
// !waveguide_node_pressures_changed_.empty(); }`); without it every step is mirrored, which is always
// correct.
#pragma once

#include <functional>

#include "waveguide.h"

namespace wayverb {
namespace waveguide {

/// Empty (default): mirror every step.  Otherwise: mirror the steps for which it returns true.
inline std::function<bool()>& cl_mirror_wanted() {
    static std::function<bool()> f;
    return f;
}

namespace detail {

class cl_mirror_bridge final {
public:
    template <typename Context>
    cl_mirror_bridge(const Context& cc, wv_engine* e, size_t nodes)
            : engine_{e},
              queue_{cc.context, cc.device},
              buffer_{cc.context, CL_MEM_READ_WRITE, sizeof(cl_float) * nodes},
              staging_(nodes, 0.0f) {
        queue_.enqueueWriteBuffer(buffer_, CL_TRUE, 0, sizeof(cl_float) * staging_.size(), staging_.data());
    }
    bool per_step() const { return true; }
    template <typename Callback>
    void invoke(Callback& callback, size_t step, size_t steps) {
        const auto& wanted = cl_mirror_wanted();
        if (!wanted || wanted()) {
            // the step's pre-update `current` is the PREVIOUS buffer after wv_run's swap (waveguide.h:121-123)
            check(wv_read_field(engine_, WV_BUF_PREVIOUS, staging_.data(), 4));
            queue_.enqueueWriteBuffer(buffer_, CL_TRUE, 0, sizeof(cl_float) * staging_.size(), staging_.data());
        }
        callback(queue_, static_cast<const cl::Buffer&>(buffer_), step, steps);
    }

private:
    wv_engine* engine_;
    cl::CommandQueue queue_;
    cl::Buffer buffer_;
    std::vector<float> staging_;
};

// any context with a `.context` member that a cl::CommandQueue can be built from
template <typename Context>
struct callback_bridge_for<Context, decltype(void(cl::CommandQueue{std::declval<const Context&>().context,
                                                                   std::declval<const Context&>().device}))>
        final {
    using type = cl_mirror_bridge;
    static type* make(const Context& cc, wv_engine* e, size_t nodes) { return new type{cc, e, nodes}; }
};

}  // namespace detail
}  // namespace waveguide
}  // namespace wayverb
