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
// a float mirror of the step's pressure field lives in a cl::Buffer of that context.
//
// WHEN the mirror is refreshed (HIP field -> host -> cl::Buffer: the whole field crosses PCIe twice,
// 2 x 4.3 GB at 1024^3 -- what the reference's own loop avoids only because its field already is a
// cl::Buffer) is SURVEY.md 8(b)'s "single behavioural deviation": ONLY FOR THE STEPS SOMEBODY WANTS IT.
// A read of a real cl::Buffer cannot be seen from here (reads of the engine's own handles can, see
// waveguide.h), so the owner of the listener says so:
//
//     waveguide::cl_mirror_wanted() = [&] { return !waveguide_node_pressures_changed_.empty(); };
//
// -- one line where the listener is connected (the application that calls
// engine::connect_waveguide_node_pressures_changed, src/combined/src/engine.cpp:260-264; src/combined
// itself stays as it is), or `cl_mirror_always()` for "whenever the callback runs", or `cl_mirror_never()`.
// While the predicate returns false the buffer keeps its zeros and `canonical` runs whole batches of passes
// on the device; a step for which it returns true gets exactly its own field (if the run has already
// passed that step: roll back, re-run up to it -- waveguide.h, run_device_observed), and the run goes on
// step by step for as long as it keeps returning true.  `cl_mirror_planes()` narrows the refresh to a
// range of z planes (a visualiser's slice).
//
// NOBODY HAS SAID ANYTHING (the predicate is empty: a build that swapped the headers and added no line): the
// buffer is not refreshed either -- an unchanged engine.cpp hands its callback to every run, listener or not, and
// mirroring 4.3 GB per step for nobody would cost it a factor of 400 -- but it holds NaN, not zeros, and the first
// such run says so on stderr: a visualiser that does read it shows nothing and the log says which line is missing,
// instead of a silent field of zeros.
#pragma once

#include <cstdio>
#include <functional>
#include <limits>

#include "waveguide.h"

namespace wayverb {
namespace waveguide {

/// Returning false: nobody reads the cl::Buffer, it is not refreshed (it holds zeros).
/// Returning true when evaluated for a step: that step's field is in the buffer when the callback runs.
/// Empty (default): not refreshed either, but the buffer holds NaN and the first run warns on stderr (above).
inline std::function<bool()>& cl_mirror_wanted() {
    static std::function<bool()> f;
    return f;
}
/// `cl_mirror_wanted() = cl_mirror_always();` -- every step is mirrored (the behaviour of a plain OpenCL field).
inline std::function<bool()> cl_mirror_always() {
    return [] { return true; };
}
/// `cl_mirror_wanted() = cl_mirror_never();` -- "nobody reads the buffer", said out loud: zeros, whole batches, no warning.
inline std::function<bool()> cl_mirror_never() {
    return [] { return false; };
}
/// Planes [z_begin, z_begin + z_count) are refreshed; z_count < 0 (default): the whole field.
struct mirror_planes final {
    int z_begin = 0;
    int z_count = -1;
};
inline mirror_planes& cl_mirror_planes() {
    static mirror_planes p;
    return p;
}

namespace detail {

class cl_mirror_bridge final {
public:
    template <typename Context>
    cl_mirror_bridge(const Context& cc, wv_engine* e, size_t nodes, size_t plane_nodes, field_guard* guard)
            : engine_{e},
              guard_{guard},
              plane_nodes_{plane_nodes ? plane_nodes : 1},
              queue_{cc.context, cc.device},
              buffer_{cc.context, CL_MEM_READ_WRITE, sizeof(cl_float) * nodes},
              nodes_{nodes} {
        // zeros until somebody wants the field (make_zeroed_buffer is what the reference's own field starts as, waveguide.h:47-56) --
        // NaN when nobody has said whether anybody will (an empty predicate): a reader then sees that it is not looking at a field
        const bool unset = !cl_mirror_wanted();
        const cl_float fill = unset ? std::numeric_limits<cl_float>::quiet_NaN() : cl_float{0};
        if (unset) {
            static bool warned = false;
            if (!warned) {
                warned = true;
                std::fprintf(stderr,
                             "wayverb_amd: a pressure callback that takes a cl::Buffer is attached and waveguide::cl_mirror_wanted() has not "
                             "been set: the buffer holds NaN, not the field.  Set it where the listener is connected -- a predicate that says "
                             "when the field is read, cl_mirror_always() or cl_mirror_never() (wayverb_amd/cl_mirror.h).\n");
            }
        }
#if defined(CL_VERSION_1_2)
        queue_.enqueueFillBuffer(buffer_, fill, 0, sizeof(cl_float) * nodes);
        queue_.finish();
#else
        const std::vector<float> block(std::min<size_t>(nodes, size_t{4} << 20), fill);
        for (size_t at = 0; at < nodes; at += block.size())
            queue_.enqueueWriteBuffer(buffer_, CL_TRUE, sizeof(cl_float) * at, sizeof(cl_float) * std::min(block.size(), nodes - at),
                                      block.data());
#endif
    }
    bool wanted_now() const {
        const auto& wanted = cl_mirror_wanted();
        return wanted && wanted();
    }
    template <typename Callback>
    void invoke(Callback& callback, size_t step, size_t steps) {
        if (wanted_now()) {
            guard_->touch();  // the engine's PREVIOUS buffer now holds this step's pre-update `current` (waveguide.h:121-123)
            if (staging_.size() != nodes_) {  // (host memory for the field only once it is wanted; page-locked: the copy down goes by DMA)
                staging_.assign(nodes_, 0.0f);
                registered_ = wv_host_register(staging_.data(), sizeof(float) * nodes_) == WV_OK;
            }
            const size_t planes_total = nodes_ / plane_nodes_;
            const mirror_planes want = cl_mirror_planes();
            const size_t z0 = want.z_count < 0 ? 0 : std::min((size_t)std::max(want.z_begin, 0), planes_total);
            const size_t zn = want.z_count < 0 ? planes_total : std::min((size_t)want.z_count, planes_total - z0);
            if (zn) {
                float* at = staging_.data() + z0 * plane_nodes_;
                check(wv_read_planes(engine_, WV_BUF_PREVIOUS, (int32_t)z0, (int32_t)zn, at, 4));
                queue_.enqueueWriteBuffer(buffer_, CL_TRUE, sizeof(cl_float) * z0 * plane_nodes_,
                                          sizeof(cl_float) * zn * plane_nodes_, at);
            }
            ++last_run_stats().fields_mirrored;
        }
        callback(queue_, static_cast<const cl::Buffer&>(buffer_), step, steps);
    }

    ~cl_mirror_bridge() {
        if (registered_) (void)wv_host_unregister(staging_.data());
    }
    cl_mirror_bridge(const cl_mirror_bridge&) = delete;
    cl_mirror_bridge& operator=(const cl_mirror_bridge&) = delete;

private:
    wv_engine* engine_;
    field_guard* guard_;
    size_t plane_nodes_;
    cl::CommandQueue queue_;
    cl::Buffer buffer_;
    size_t nodes_;
    std::vector<float> staging_;
    bool registered_ = false;
};

// any context with a `.context` member that a cl::CommandQueue can be built from
template <typename Context>
struct callback_bridge_for<Context, decltype(void(cl::CommandQueue{std::declval<const Context&>().context,
                                                                   std::declval<const Context&>().device}))>
        final {
    using type = cl_mirror_bridge;
    static type* make(const Context& cc, wv_engine* e, size_t nodes, size_t plane_nodes, field_guard* guard) {
        return new type{cc, e, nodes, plane_nodes, guard};
    }
};

}  // namespace detail
}  // namespace waveguide
}  // namespace wayverb
