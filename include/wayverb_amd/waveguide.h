// wayverb_amd/waveguide.h -- C++14 host mirror of wayverb's `waveguide::run` interface over the
// C ABI of wayverb_amd.h.  Header only; link against libwayverb_amd.so.
//
// Same names, argument meaning and error behaviour as the reference (paths relative to the
// reference repository root):
//   waveguide::run<pre, post>                 src/waveguide/include/waveguide/waveguide.h:36-126
//   preprocessor::hard_source / soft_source   include/waveguide/preprocessor/{hard,soft}_source.h
//   preprocessor::gaussian                    src/waveguide/src/preprocessor/gaussian.cpp:12-53
//   postprocessor::node / directional_receiver  src/waveguide/src/postprocessor/*.cpp
//   detail::canonical_impl / canonical        include/waveguide/canonical.h:29-88,100-127
//   mesh / vectors / mesh_descriptor          include/waveguide/{mesh,setup,mesh_descriptor}.h
//   core::read_value / write_value / ...      src/core/include/core/cl/common.h:24-57
//   core::exceptions::value_is_nan / _inf     src/core/include/core/exceptions.h:9-30
//
// What differs, deliberately:
//   - step callbacks receive `waveguide::queue&` / `waveguide::buffer&` handles instead of
//     cl::CommandQueue / cl::Buffer.  Callers written with generic lambdas
//     (`[](auto& queue, const auto& buffer, auto step)`, as every call site in the reference is)
//     compile unchanged; the core:: helper overloads below accept the handles.  A caller that is
//     pinned to the OpenCL types -- src/combined/include/combined/waveguide_base.h:47-59 type-erases
//     its pressure callback as std::function<void(cl::CommandQueue&, const cl::Buffer&, size_t,
//     size_t)> -- includes wayverb_amd/cl_mirror.h as well and gets exactly those (a float mirror of
//     the field in a real cl::Buffer, refreshed before each call).
//   - the context, mesh, position and environment parameters are template parameters: the
//     reference's own core::compute_context (OpenCL context + device, the ray tracer keeps using it),
//     glm::vec3 and core::environment are accepted as they are; compat_core.h provides stand-ins
//     for translation units outside the reference tree, and defines nothing when
//     WAYVERB_AMD_HAVE_REFERENCE_CORE says the real headers are there.
//   - `run_device` / `canonical` keep the source and the receivers on the GPU: no per-step PCIe
//     round trip.  `run` with arbitrary callbacks synchronises every step, like the
//     reference does.  `canonical` runs AHEAD of its pressure callback: steps are taken in batches
//     (two-step passes on the device) and the callbacks of a batch fire afterwards, in order, with
//     the same arguments as in the reference.  A callback that looks at the field -- the reference's
//     4-argument form may, canonical.h:66-69 -- gets exactly its step's field all the same: the
//     engine is rolled back to the checkpoint taken before the batch and re-run up to that step
//     (wv_checkpoint / wv_rollback), and the run then proceeds step by step for as long as somebody
//     keeps looking.  Nobody has to say in advance whether a callback reads (`progress_only` is a
//     promise that it never does, which saves the checkpoints).
#pragma once

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <ctime>
#include <experimental/optional>
#include <functional>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../wayverb_amd.h"
#include "compat_core.h"

namespace wayverb {
namespace waveguide {

// ---- engine plumbing ----------------------------------------------------------------------------
class engine_error : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

namespace detail {
inline void check(int rc) {
    if (rc != WV_OK) throw engine_error(std::string("wayverb_amd: ") + wv_last_error());
}
struct engine_deleter {
    void operator()(wv_engine* e) const { wv_destroy(e); }
};
using engine_ptr = std::unique_ptr<wv_engine, engine_deleter>;

/// waveguide.h:102-118 -- the same exception types and messages
inline void throw_for_flag(int flag) {
    if (flag & WV_FLAG_INF) throw core::exceptions::value_is_inf("Pressure value is inf, check filter coefficients.");
    if (flag & WV_FLAG_NAN) throw core::exceptions::value_is_nan("Pressure value is nan, check filter coefficients.");
    if (flag & WV_FLAG_OUTSIDE_MESH) throw std::runtime_error("Tried to read non-existant node.");
    if (flag & WV_FLAG_SUSPICIOUS_BOUNDARY) throw std::runtime_error("Suspicious boundary read.");
}
}  // namespace detail

namespace detail {
/// What stands between `canonical`'s pressure callback and the field.  The run is ahead of its observers (see the header
/// comment): the field of the LAST step of a batch is on the device, that of an earlier step is brought back the moment an
/// observer looks -- every access through a `buffer` handle, or the cl::Buffer mirror when its owner wants it, calls touch().
struct field_guard final {
    std::function<void()> materialise;  // makes the engine's PREVIOUS buffer hold the field of the step being fired
    bool fresh = true;                  // the device holds the field of the step being fired
    bool looked = false;                // somebody looked during the callback being fired
    void touch() {
        looked = true;
        if (!fresh && materialise) {
            materialise();
            fresh = true;
        }
    }
};
}  // namespace detail

namespace detail {
/// The values a step's `post` reads, learned and then served without a trip to the device (run<pre, post> ahead of its callbacks):
/// every index a callback reads through core::read_value on a `buffer` handle is noted; from the next batch on the engine records
/// that node like a receiver, and the read is answered from the step's row.  A read of an index not (yet) in the plan goes to the
/// engine -- through the field_guard, which brings that step's field back if the run has passed it.
struct read_plan final {
    std::vector<uint64_t> indices;  // in column order
    size_t served_columns = 0;      // how many of them the row in `row` holds
    const double* row = nullptr;    // the step being fired: [served_columns]
    bool grew = false;              // indices were added since the engine was last told
    size_t reads_served = 0, reads_missed = 0;
    bool serve(size_t index, double* v) {
        for (size_t c = 0; c < indices.size(); ++c)  // (a handful of receivers: a linear look-up beats a map)
            if (indices[c] == index) {
                if (row && c < served_columns) {
                    *v = row[c];
                    ++reads_served;
                    return true;
                }
                ++reads_missed;
                return false;
            }
        if (indices.size() < 4096) {  // (a callback that reads whole fields value by value keeps its per-step path)
            indices.push_back(index);
            grew = true;
        }
        ++reads_missed;
        return false;
    }
};
}  // namespace detail

/// What a step callback sees instead of cl::CommandQueue (hard_source.h:17, directional_receiver.h:36).
class queue final {
public:
    explicit queue(wv_engine* e) : e_{e} {}
    wv_engine* engine() const { return e_; }

private:
    wv_engine* e_;
};
/// What a step callback sees instead of cl::Buffer: one of the engine's two pressure fields.
class buffer final {
public:
    buffer(wv_engine* e, int which, size_t items, int precision = WV_PRECISION_F64, detail::field_guard* guard = nullptr,
           detail::read_plan* plan = nullptr)
            : e_{e}, which_{which}, items_{items}, precision_{precision}, guard_{guard}, plan_{plan} {}
    /// The engine whose field this is, for wv_read_planes & co. on the handle.  Asking for it counts as looking at the
    /// field (in `canonical` the step's field is brought back first, see detail::field_guard).
    wv_engine* engine() const {
        if (guard_) guard_->touch();
        return e_;
    }
    int which() const { return which_; }
    size_t items() const { return items_; }
    int precision() const { return precision_; }  // WV_PRECISION_*: how the engine stores pressures
    /// a value the run has on the host already (detail::read_plan); false: ask the engine
    bool served(size_t index, double* v) const { return plan_ && plan_->serve(index, v); }

private:
    wv_engine* e_;
    int which_;
    size_t items_;
    int precision_;
    detail::field_guard* guard_;
    detail::read_plan* plan_;
};

}  // namespace waveguide

namespace core {  // the buffer helpers of src/core/include/core/cl/common.h:29-57, on the handles

template <typename T>
size_t items_in_buffer(const waveguide::buffer& b) {
    return b.items();
}
template <typename T>
T read_value(waveguide::queue&, const waveguide::buffer& b, size_t index) {
    double v = 0;
    if (b.served(index, &v)) return static_cast<T>(v);
    waveguide::detail::check(wv_read_value(b.engine(), b.which(), index, &v));
    return static_cast<T>(v);
}
template <typename T>
void write_value(waveguide::queue&, waveguide::buffer& b, size_t index, T val) {
    waveguide::detail::check(wv_write_value(b.engine(), b.which(), index, static_cast<double>(val)));
}
inline void read_buffer_into(const waveguide::buffer& b, float* dst) {
    waveguide::detail::check(wv_read_field(b.engine(), b.which(), dst, 4));
}
inline void read_buffer_into(const waveguide::buffer& b, double* dst) {
    waveguide::detail::check(wv_read_field(b.engine(), b.which(), dst, 8));
}
template <typename T>
std::vector<T> read_from_buffer(waveguide::queue&, const waveguide::buffer& b) {
    std::vector<T> ret(b.items());
    read_buffer_into(b, ret.data());
    return ret;
}
/// cl::copy(queue, begin, end, buffer) as preprocessor/gaussian.cpp:50 uses it
template <typename It>
void copy(waveguide::queue&, It begin, It end, waveguide::buffer& b) {
    using T = typename std::iterator_traits<It>::value_type;
    std::vector<T> tmp(begin, end);
    if (tmp.size() != b.items()) throw std::runtime_error("copy: size does not match the buffer");
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "float or double field");
    waveguide::detail::check(wv_write_field(b.engine(), b.which(), tmp.data(), (int)sizeof(T)));
}

}  // namespace core

namespace waveguide {

// ---- mesh containers ----------------------------------------------------------------------------
using condensed_node = wv_condensed_node;                  // cl/structs.h:19-22
using coefficients_canonical = wv_coefficients_canonical;  // cl/filter_structs.h:65-66
constexpr uint32_t no_neighbor = ~uint32_t{0};             // cl/utils.h:29

template <size_t N>
struct boundary_index_array final {  // cl/boundary_index_array.h:8-11
    uint32_t array[N];
};
struct boundary_index_data final {  // boundary_coefficient_finder.h:36-40
    std::vector<boundary_index_array<1>> b1;
    std::vector<boundary_index_array<2>> b2;
    std::vector<boundary_index_array<3>> b3;
};

struct vec3 final {
    float x, y, z;
};
struct dvec3 final {
    double x, y, z;
};
struct ivec3 final {
    int x, y, z;
};

namespace detail {
// a position is anything with floating-point members x, y, z: `vec3` above, glm::vec3 in the reference tree
template <typename V>
using enable_if_position = typename std::enable_if<std::is_floating_point<decltype(std::declval<V>().x)>::value, int>::type;
}  // namespace detail

struct mesh_descriptor final {  // mesh_descriptor.h:14-20
    vec3 min_corner;
    ivec3 dimensions;
    float spacing;
};

inline size_t compute_index(const mesh_descriptor& d, const ivec3& pos) {  // mesh_descriptor.cpp:7-10
    return (size_t)pos.x + (size_t)pos.y * d.dimensions.x + (size_t)pos.z * d.dimensions.x * d.dimensions.y;
}
inline ivec3 compute_locator(const mesh_descriptor& d, size_t index) {  // mesh_descriptor.cpp:16-20
    const size_t x = index % d.dimensions.x, q = index / d.dimensions.x;
    return ivec3{(int)x, (int)(q % d.dimensions.y), (int)((q / d.dimensions.y) % d.dimensions.z)};
}
template <typename V, detail::enable_if_position<V> = 0>
ivec3 compute_locator(const mesh_descriptor& d, const V& v) {  // :22-25, glm::round
    return ivec3{(int)std::round(((float)v.x - d.min_corner.x) / d.spacing),
                 (int)std::round(((float)v.y - d.min_corner.y) / d.spacing),
                 (int)std::round(((float)v.z - d.min_corner.z) / d.spacing)};
}
template <typename V, detail::enable_if_position<V> = 0>
size_t compute_index(const mesh_descriptor& d, const V& pos) {
    return compute_index(d, compute_locator(d, pos));
}
inline vec3 compute_position(const mesh_descriptor& d, const ivec3& l) {  // :27-30
    return vec3{d.min_corner.x + l.x * d.spacing, d.min_corner.y + l.y * d.spacing, d.min_corner.z + l.z * d.spacing};
}
inline vec3 compute_position(const mesh_descriptor& d, size_t index) {
    return compute_position(d, compute_locator(d, index));
}
inline std::array<uint32_t, 6> compute_neighbors(const mesh_descriptor& d, size_t index) {  // :36-63
    const ivec3 l = compute_locator(d, index);
    const ivec3 n[6] = {{l.x - 1, l.y, l.z}, {l.x + 1, l.y, l.z}, {l.x, l.y - 1, l.z},
                        {l.x, l.y + 1, l.z}, {l.x, l.y, l.z - 1}, {l.x, l.y, l.z + 1}};
    std::array<uint32_t, 6> ret;
    for (int i = 0; i < 6; ++i) {
        const bool inside = n[i].x >= 0 && n[i].y >= 0 && n[i].z >= 0 && n[i].x < d.dimensions.x &&
                            n[i].y < d.dimensions.y && n[i].z < d.dimensions.z;
        ret[i] = inside ? (uint32_t)compute_index(d, n[i]) : no_neighbor;
    }
    return ret;
}
inline double compute_sample_rate(const mesh_descriptor& d, double speed_of_sound) {  // :72-74, config.cpp:19-21
    return 1.0 / (d.spacing / (speed_of_sound * std::sqrt(3.0)));
}
inline size_t compute_num_nodes(const mesh_descriptor& d) {
    return (size_t)d.dimensions.x * d.dimensions.y * d.dimensions.z;
}

constexpr bool is_inside(const condensed_node& c) { return c.boundary_type & WV_ID_INSIDE; }  // setup.h:21-23

class vectors final {  // setup.h:27-48, setup.cpp:7-50
public:
    vectors(std::vector<condensed_node> nodes, std::vector<coefficients_canonical> coefficients,
            boundary_index_data boundary_index_data)
            : condensed_nodes_(std::move(nodes)),
              coefficients_(std::move(coefficients)),
              boundary_index_data_(std::move(boundary_index_data)) {}

    const std::vector<condensed_node>& get_condensed_nodes() const { return condensed_nodes_; }
    const std::vector<coefficients_canonical>& get_coefficients() const { return coefficients_; }
    const boundary_index_data& get_boundary_index_data() const { return boundary_index_data_; }

    void set_coefficients(coefficients_canonical c) { std::fill(coefficients_.begin(), coefficients_.end(), c); }
    void set_coefficients(std::vector<coefficients_canonical> c) {
        if (c.size() != coefficients_.size())
            throw std::runtime_error(
                    "Size of new coefficients vector must be equal to the existing one in order to maintain object "
                    "invariants.");
        coefficients_ = std::move(c);
    }

private:
    std::vector<condensed_node> condensed_nodes_;
    std::vector<coefficients_canonical> coefficients_;
    boundary_index_data boundary_index_data_;
};

class mesh final {  // mesh.h:12-26
public:
    mesh(mesh_descriptor descriptor, vectors vectors) : descriptor_(descriptor), vectors_(std::move(vectors)) {}
    const mesh_descriptor& get_descriptor() const { return descriptor_; }
    const vectors& get_structure() const { return vectors_; }
    void set_coefficients(coefficients_canonical c) { vectors_.set_coefficients(c); }
    void set_coefficients(std::vector<coefficients_canonical> c) { vectors_.set_coefficients(std::move(c)); }

private:
    mesh_descriptor descriptor_;
    vectors vectors_;
};
inline bool is_inside(const mesh& m, size_t node_index) {
    return is_inside(m.get_structure().get_condensed_nodes()[node_index]);
}

/// fitted_boundary.h:21-48
inline coefficients_canonical to_impedance_coefficients(const coefficients_canonical& c) {
    coefficients_canonical ret{};
    for (int i = 0; i < 7; ++i) {
        ret.b[i] = c.a[i] + c.b[i];
        ret.a[i] = c.a[i] - c.b[i];
    }
    if (ret.a[0] != 0) {
        const double norm = 1.0 / ret.a[0];
        for (int i = 0; i < 7; ++i) {
            ret.b[i] *= norm;
            ret.a[i] *= norm;
        }
    }
    return ret;
}
/// fitted_boundary.h:72-75 (core/surfaces.h:25-33: reflectance = sqrt(1 - absorption))
inline coefficients_canonical to_flat_coefficients(double absorption) {
    coefficients_canonical c{};
    c.b[0] = std::sqrt(1.0 - absorption);
    c.a[0] = 1.0;
    return to_impedance_coefficients(c);
}

/// calibration.h:20-31
inline double rectilinear_calibration_factor(double grid_spacing, double acoustic_impedance) {
    return std::sqrt(acoustic_impedance / (4 * M_PI)) / (0.3405 * grid_spacing);
}

/// Synthetic box mesh: every wall takes coefficient 0.
inline mesh make_box_mesh(int nx, int ny, int nz, float spacing, coefficients_canonical wall) {
    std::vector<condensed_node> nodes((size_t)nx * ny * nz);
    uint64_t counts[3] = {0, 0, 0};
    if (wv_make_box_nodes(nx, ny, nz, 0, nz, 0, nz, nodes.data(), counts) != WV_OK)
        throw std::runtime_error("box needs at least 5 nodes per axis");
    boundary_index_data bid;
    bid.b1.assign(counts[0], boundary_index_array<1>{{0}});
    bid.b2.assign(counts[1], boundary_index_array<2>{{0, 0}});
    bid.b3.assign(counts[2], boundary_index_array<3>{{0, 0, 0}});
    return mesh{mesh_descriptor{vec3{0, 0, 0}, ivec3{nx, ny, nz}, spacing},
                vectors{std::move(nodes), std::vector<coefficients_canonical>{wall}, std::move(bid)}};
}

// ---- engine construction ------------------------------------------------------------------------
namespace detail {
// HIP device of a context: its `.device` when that is an ordinal (compat_core.h), else -1 = the calling
// thread's current device (the reference's context names an OpenCL device, which says nothing here)
template <typename Context>
auto device_of(const Context& cc, int) -> decltype(int{cc.device}) {
    return cc.device;
}
template <typename Context>
int device_of(const Context&, long) {
    return -1;
}
template <typename Context>
int device_of(const Context& cc) {
    return device_of(cc, 0);
}

/// `Mesh` is this header's `mesh` or the reference's own (mesh.h:12-26): anything with its accessors
/// whose element types have the device layouts of cl/structs.h (checked below).
template <typename Context, typename Mesh>
engine_ptr make_engine(const Context& cc, const Mesh& m, int precision) {
    const auto& s = m.get_structure();
    const auto& bid = s.get_boundary_index_data();
    using node_t = typename std::decay<decltype(s.get_condensed_nodes()[0])>::type;
    using coeff_t = typename std::decay<decltype(s.get_coefficients()[0])>::type;
    static_assert(sizeof(node_t) == sizeof(wv_condensed_node), "condensed_node layout (cl/structs.h:19-22)");
    static_assert(sizeof(coeff_t) == sizeof(wv_coefficients_canonical), "coefficients_canonical layout");
    static_assert(sizeof(bid.b1[0]) == 4 && sizeof(bid.b2[0]) == 8 && sizeof(bid.b3[0]) == 12,
                  "boundary_index_array<N> layout (cl/boundary_index_array.h:8-11)");
    wv_mesh wm{};
    wm.nx = m.get_descriptor().dimensions.x;
    wm.ny = m.get_descriptor().dimensions.y;
    wm.nz = m.get_descriptor().dimensions.z;
    wm.nodes = reinterpret_cast<const wv_condensed_node*>(s.get_condensed_nodes().data());
    wm.coefficients = reinterpret_cast<const wv_coefficients_canonical*>(s.get_coefficients().data());
    wm.num_coefficients = (uint32_t)s.get_coefficients().size();
    wm.boundary_indices_1 = bid.b1.empty() ? nullptr : reinterpret_cast<const uint32_t*>(bid.b1.data());
    wm.boundary_indices_2 = bid.b2.empty() ? nullptr : reinterpret_cast<const uint32_t*>(bid.b2.data());
    wm.boundary_indices_3 = bid.b3.empty() ? nullptr : reinterpret_cast<const uint32_t*>(bid.b3.data());
    wm.num_boundary_1 = bid.b1.size();
    wm.num_boundary_2 = bid.b2.size();
    wm.num_boundary_3 = bid.b3.size();
    wv_options opt;
    wv_default_options(&opt);
    opt.precision = precision;
    opt.device = device_of(cc);
    wv_engine* raw = nullptr;
    check(wv_create(&wm, &opt, &raw));
    return engine_ptr(raw);
}
}  // namespace detail

/// Pressure storage, process-wide.  Default WV_PRECISION_F64 -- the double-precision engine of the
/// north star, the default of every layer (wv_default_options, the Python binding).  Callbacks see
/// floats either way (read_value<float>, read_from_buffer<float>).  WV_PRECISION_F32 stores
/// pressures as the reference does (cl_float) and reproduces its fields bit for bit.
inline int& default_precision() {
    static int p = WV_PRECISION_F64;
    return p;
}

// ---- run: arbitrary step callbacks (waveguide.h:36-126) -------------------------------------------
namespace preprocessor {
template <typename It>
class hard_source;
template <typename It>
class soft_source;
}  // namespace preprocessor
namespace detail {
// the step pre-processors the engine can run by itself (wv_set_source): this header's own hard_source / soft_source
template <typename T>
struct device_source : std::false_type {};
template <typename It>
struct device_source<preprocessor::hard_source<It>> : std::true_type {
    static constexpr int kind = WV_SOURCE_HARD;
};
template <typename It>
struct device_source<preprocessor::soft_source<It>> : std::true_type {
    static constexpr int kind = WV_SOURCE_SOFT;
};
template <typename Context, typename Mesh, typename Source, typename step_postprocessor>
size_t run_ahead(const Context& cc, const Mesh& mesh, Source& pre, step_postprocessor& post, const std::atomic_bool& keep_going);

// the loop as the reference writes it: one host round trip per step, `pre` and `post` free to do anything to the field
template <typename Context, typename Mesh, typename step_preprocessor, typename step_postprocessor>
size_t run_step_by_step(const Context& cc, const Mesh& mesh, step_preprocessor& pre, step_postprocessor& post,
                        const std::atomic_bool& keep_going) {
    auto engine = make_engine(cc, mesh, default_precision());
    const size_t num_nodes = mesh.get_structure().get_condensed_nodes().size();
    queue q{engine.get()};
    buffer current{engine.get(), WV_BUF_CURRENT, num_nodes, default_precision()};  // the handle follows the swaps
    size_t step = 0;
    for (; pre(q, current, step) && keep_going; ++step) {
        int32_t flag = 0;
        check(wv_step(engine.get(), &flag));
        throw_for_flag(flag);
        post(q, current, step);
        check(wv_swap(engine.get()));
    }
    return step;
}
template <typename Context, typename Mesh, typename step_preprocessor, typename step_postprocessor>
size_t run_dispatch(const Context& cc, const Mesh& mesh, step_preprocessor& pre, step_postprocessor& post, const std::atomic_bool& keep_going,
                    std::true_type) {
    return run_ahead(cc, mesh, pre, post, keep_going);
}
template <typename Context, typename Mesh, typename step_preprocessor, typename step_postprocessor>
size_t run_dispatch(const Context& cc, const Mesh& mesh, step_preprocessor& pre, step_postprocessor& post, const std::atomic_bool& keep_going,
                    std::false_type) {
    return run_step_by_step(cc, mesh, pre, post, keep_going);
}
}  // namespace detail

/// `pre` any callable: the reference's loop, a host round trip per step.  `pre` one of this header's preprocessor::hard_source /
/// soft_source (what 10 of the reference's 11 call sites pass -- bin/boundary_test/boundary_test.cpp:137-147, src/waveguide/tests/
/// waveguide_tests.cpp:95-107, ...): the source runs on the device and the steps are taken in batches AHEAD of `post`, which still fires
/// once per step, in order, with the step's field behind its handle -- the values it reads through core::read_value are learned in the
/// first steps and recorded on the device from then on (detail::read_plan); a read nobody has made before gets its step's field all
/// the same (rollback, re-run: run_device_observed).  last_run_stats() says what a run did.
template <typename Context, typename Mesh, typename step_preprocessor, typename step_postprocessor>
size_t run(const Context& cc, const Mesh& mesh, step_preprocessor&& pre, step_postprocessor&& post,
           const std::atomic_bool& keep_going) {
    using source_t = typename std::decay<step_preprocessor>::type;
    return detail::run_dispatch(cc, mesh, pre, post, keep_going, std::integral_constant<bool, detail::device_source<source_t>::value>{});
}

// ---- step pre-processors --------------------------------------------------------------------------
namespace preprocessor {

template <typename It>
class hard_source final {  // preprocessor/hard_source.h:9-29
public:
    hard_source(size_t node, It begin, It end) : node_{node}, begin_{begin}, end_{end} {}
    template <typename Q, typename B>
    bool operator()(Q& q, B& b, size_t) {
        if (begin_ == end_) return false;
        core::write_value(q, b, node_, *begin_++);
        return true;
    }
    size_t get_node() const { return node_; }
    It begin() const { return begin_; }
    It end() const { return end_; }
    void advance(size_t samples) { std::advance(begin_, samples); }  // (run: the device took them)

private:
    size_t node_;
    It begin_, end_;
};
template <typename It>
auto make_hard_source(size_t node, It begin, It end) {
    return hard_source<It>{node, begin, end};
}

template <typename It>
class soft_source final {  // preprocessor/soft_source.h:9-31
public:
    soft_source(size_t node, It begin, It end) : node_{node}, begin_{begin}, end_{end} {}
    template <typename Q, typename B>
    bool operator()(Q& q, B& b, size_t) {
        if (begin_ == end_) return false;
        // the sum is taken in the field's own precision: float like the reference (soft_source.h:21-24)
        // on a float field, double on the fp64 engine (where rounding the field to float first would
        // throw away what the engine carries)
        if (b.precision() == WV_PRECISION_F32) {
            const auto current_pressure = core::read_value<float>(q, b, node_);
            core::write_value(q, b, node_, current_pressure + static_cast<float>(*begin_++));
        } else {
            const auto current_pressure = core::read_value<double>(q, b, node_);
            core::write_value(q, b, node_, current_pressure + static_cast<double>(*begin_++));
        }
        return true;
    }
    size_t get_node() const { return node_; }
    It begin() const { return begin_; }
    It end() const { return end_; }
    void advance(size_t samples) { std::advance(begin_, samples); }  // (run: the device took them)

private:
    size_t node_;
    It begin_, end_;
};
template <typename It>
auto make_soft_source(size_t node, It begin, It end) {
    return soft_source<It>{node, begin, end};
}

class gaussian final {  // src/preprocessor/gaussian.cpp:12-53
public:
    static float compute(const vec3& x, float sdev) {
        const double len2 = (double)x.x * x.x + (double)x.y * x.y + (double)x.z * x.z;
        return (float)(std::exp(-len2 / (2 * std::pow(sdev, 2))) / std::pow(sdev * std::sqrt(2 * M_PI), 3));
    }
    gaussian(const mesh_descriptor& descriptor, const vec3& centre_pos, float sdev, size_t steps)
            : descriptor_(descriptor), centre_pos_(centre_pos), sdev_{sdev}, steps_{steps} {}
    template <typename Q, typename B>
    bool operator()(Q& q, B& b, size_t step) const {
        if (step == steps_) return false;
        if (step == 0) {
            const size_t nodes = core::items_in_buffer<float>(b);
            std::vector<float> pressures;
            pressures.reserve(nodes);
            for (size_t i = 0; i != nodes; ++i) {
                const vec3 p = compute_position(descriptor_, i);
                pressures.emplace_back(compute(vec3{p.x - centre_pos_.x, p.y - centre_pos_.y, p.z - centre_pos_.z}, sdev_));
            }
            core::copy(q, pressures.begin(), pressures.end(), b);
        }
        return true;
    }

private:
    mesh_descriptor descriptor_;
    vec3 centre_pos_;
    float sdev_;
    size_t steps_;
};

}  // namespace preprocessor

// ---- step post-processors -------------------------------------------------------------------------
namespace postprocessor {

class node final {  // src/postprocessor/node.cpp:11-20
public:
    explicit node(size_t output_node) : output_node_{output_node} {}
    using return_type = float;
    template <typename Q, typename B>
    return_type operator()(Q& q, const B& b, size_t) const {
        return core::read_value<float>(q, b, output_node_);
    }
    size_t get_output_node() const { return output_node_; }

private:
    size_t output_node_;
};

class directional_receiver final {  // src/postprocessor/directional_receiver.cpp:10-69
public:
    directional_receiver(const mesh_descriptor& md, double sample_rate, double ambient_density, size_t output_node)
            : mesh_spacing_{md.spacing},
              sample_rate_{sample_rate},
              ambient_density_{ambient_density},
              output_node_{output_node},
              surrounding_nodes_(compute_neighbors(md, output_node)) {
        for (const auto& i : surrounding_nodes_)
            if (i == no_neighbor)
                throw std::runtime_error(
                        "Can't place directional_receiver at this node as it is adjacent to a boundary.");
    }
    struct output final {
        vec3 intensity;
        float pressure;
    };
    using return_type = output;

    /// the arithmetic of directional_receiver.cpp:29-67 on 7 already-read float pressures
    return_type accumulate(float pressure, const float* neighbours6) {
        float surrounding[6];
        for (int i = 0; i < 6; ++i) surrounding[i] = (float)((neighbours6[i] - pressure) / mesh_spacing_);
        const dvec3 m{(surrounding[1] - surrounding[0]) * 0.5, (surrounding[3] - surrounding[2]) * 0.5,
                      (surrounding[5] - surrounding[4]) * 0.5};
        const double k = ambient_density_ * sample_rate_;
        velocity_.x -= m.x / k;
        velocity_.y -= m.y / k;
        velocity_.z -= m.z / k;
        const double p = static_cast<double>(pressure);
        return {vec3{(float)(velocity_.x * p), (float)(velocity_.y * p), (float)(velocity_.z * p)}, pressure};
    }
    template <typename Q, typename B>
    return_type operator()(Q& q, const B& b, size_t) {
        const auto pressure = core::read_value<float>(q, b, output_node_);
        float n[6];
        for (int i = 0; i < 6; ++i) n[i] = core::read_value<float>(q, b, surrounding_nodes_[i]);
        return accumulate(pressure, n);
    }
    size_t get_output_node() const { return output_node_; }
    const std::array<uint32_t, 6>& get_surrounding_nodes() const { return surrounding_nodes_; }

private:
    double mesh_spacing_, sample_rate_, ambient_density_;
    size_t output_node_;
    std::array<uint32_t, 6> surrounding_nodes_;
    dvec3 velocity_{0, 0, 0};
};

}  // namespace postprocessor

/// What the calling thread's most recent `run_device` / `canonical` did, for tests, tools and the curious.
struct run_stats final {
    size_t steps = 0;            // loop iterations completed (= callbacks fired)
    size_t batches = 0;          // wv_run calls that advanced the run
    size_t checkpoints = 0;      // batches taken speculatively (wv_checkpoint before them)
    size_t rollbacks = 0;        // times an observer looked at a step the run had already passed
    size_t steps_rerun = 0;      // steps computed a second time after a rollback
    size_t fields_looked_at = 0; // callbacks during which somebody looked at the field
    size_t fields_mirrored = 0;  // cl_mirror.h: whole-field (or plane-range) copies into the cl::Buffer
    size_t reads_served = 0;     // run<pre, post> ahead of its callbacks: core::read_value calls answered from recorded rows ...
    size_t reads_missed = 0;     // ... and those that went to the engine (the first steps, an index nobody had read before)
    uint64_t passes = 0;         // two-step passes the engine took (WV_QUERY_PASSES)
    double seconds = 0;          // wall time of the step loop (engine set-up excluded)
};
inline run_stats& last_run_stats() {
    static thread_local run_stats s;
    return s;
}

/// How long a batch of steps may keep the device to itself (seconds of predicted work, from the rate of the batch before): what bounds
/// the latency of `keep_going` and the cadence of progress callbacks whatever the mesh size -- the reference looks at keep_going before
/// every step (waveguide.h:80) and reports progress after every step (src/combined/src/engine.cpp:171-172); a batch of 256 steps of a
/// 1024^3 mesh would be 0.7 s.  <= 0: batches are bounded by their step count only.
inline double& batch_seconds() {
    static double s = 0.05;
    return s;
}

namespace detail {
inline double seconds_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
}  // namespace detail

namespace detail {
/// steps that fit `batch_seconds()` at `step_seconds` per step (measured on the batch before; 0 = not known yet: `probe` steps)
inline size_t steps_within_budget(double step_seconds, size_t probe) {
    const double budget = batch_seconds();
    if (budget <= 0) return ~size_t{0};
    if (step_seconds <= 0) return probe;
    const double n = budget / step_seconds;
    return n < 1.0 ? size_t{1} : (n > 1e9 ? ~size_t{0} : (size_t)n);
}
}  // namespace detail

// ---- run_device: single-node source + node receivers, device resident -----------------------------
/// Equivalent to `run(cc, mesh, hard/soft_source(node, begin, end), <read `receivers` each step>,
/// keep_going)`, without a host round trip per step.  `on_batch(first_step, n_steps, samples)` is
/// called with samples[n_steps][receivers.size()] (doubles holding the field's own precision)
/// every `batch` steps; keep_going is polled at the same cadence.  Returns completed steps.
enum class source_kind { hard = WV_SOURCE_HARD, soft = WV_SOURCE_SOFT };

/// on_batch may also take a fourth argument, the engine (`wv_engine*`): how `canonical` hands the field
/// to a pressure callback.
namespace detail {
template <typename OnBatch>
auto call_on_batch(OnBatch& f, wv_engine* e, size_t first, size_t n, const std::vector<double>& s, int)
        -> decltype(f(first, n, s, e), void()) {
    f(first, n, s, e);
}
template <typename OnBatch>
void call_on_batch(OnBatch& f, wv_engine*, size_t first, size_t n, const std::vector<double>& s, long) {
    f(first, n, s);
}
}  // namespace detail

template <typename Context, typename Mesh, typename It, typename OnBatch>
size_t run_device(const Context& cc, const Mesh& mesh, source_kind kind, size_t source_node, It begin, It end,
                  const std::vector<uint64_t>& receivers, OnBatch&& on_batch, const std::atomic_bool& keep_going,
                  size_t batch = 256) {
    auto engine = detail::make_engine(cc, mesh, default_precision());
    std::vector<double> signal(begin, end);
    detail::check(wv_set_source(engine.get(), (int)kind, source_node, signal.data(), signal.size()));
    detail::check(wv_set_receivers(engine.get(), receivers.data(), (uint32_t)receivers.size()));
    size_t done_total = 0;
    std::vector<double> samples;
    run_stats& stats = last_run_stats();
    stats = run_stats{};
    const double t0 = detail::seconds_now();
    struct at_exit final {  // (also when a flag or a callback throws)
        run_stats& stats;
        wv_engine* e;
        const size_t& done_total;
        double t0;
        ~at_exit() {
            uint64_t passes = 0;
            if (wv_query(e, WV_QUERY_PASSES, &passes) == WV_OK) stats.passes = passes;
            stats.steps = done_total;
            stats.seconds = detail::seconds_now() - t0;
        }
    } finish{stats, engine.get(), done_total, t0};
    double step_seconds = 0;  // of the batch before (the first batch is a short one that finds out)
    while (done_total < signal.size() && keep_going) {
        const uint64_t want = std::min<uint64_t>(std::min<uint64_t>(batch, detail::steps_within_budget(step_seconds, 8)), signal.size() - done_total);
        uint64_t done = 0;
        int32_t flag = 0;
        const double t_batch = detail::seconds_now();
        detail::check(wv_run(engine.get(), want, &done, &flag));
        if (done) step_seconds = (detail::seconds_now() - t_batch) / (double)done;
        if (done) {
            ++stats.batches;
            samples.resize((size_t)done * receivers.size());
            detail::check(wv_fetch_receivers(engine.get(), done_total, done, samples.data()));
            detail::call_on_batch(on_batch, engine.get(), done_total, (size_t)done, samples, 0);
        }
        done_total += (size_t)done;
        detail::throw_for_flag(flag);
        if (done < want) break;
    }
    return done_total;
}

// ---- canonical (canonical.h:29-88, 100-127) --------------------------------------------------------
struct band final {  // bandpass_band.h:11-15
    std::vector<postprocessor::directional_receiver::output> directional;
    double sample_rate;
};
struct bandpass_band final {  // bandpass_band.h:17-20
    waveguide::band band;
    util::range<double> valid_hz;
};
struct single_band_parameters final {  // simulation_parameters.h:9-16
    double cutoff;
    double usable_portion;
};
constexpr double compute_sampling_frequency(double cutoff, double usable_portion) {  // :65-68
    return cutoff / (0.25 * usable_portion);
}
constexpr double compute_sampling_frequency(const single_band_parameters& p) {  // :70-72
    return compute_sampling_frequency(p.cutoff, p.usable_portion);
}

/// Marks a pressure callback that never looks at the field (progress bars): a promise that saves `canonical` the
/// checkpoints it otherwise takes so that a callback MAY look (see the header comment).
template <typename F>
struct progress_only_t final {
    F f;
};
template <typename F>
progress_only_t<typename std::decay<F>::type> progress_only(F&& f) {
    return {std::forward<F>(f)};
}

namespace detail {

/// How a pressure callback gets to see the field.  This one hands out the engine handles; cl_mirror.h
/// adds the one for contexts that carry an OpenCL context (real cl::CommandQueue / cl::Buffer).
class handle_bridge final {
public:
    // wv_run has swapped the fields by the time the callback fires: the step's pre-update `current`
    // (what waveguide.h:121 hands to `post`) is the engine's PREVIOUS buffer now
    // (the handle carries the ENGINE's pressure type: a callback may branch on buffer::precision(), as soft_source does)
    handle_bridge(wv_engine* e, size_t nodes, size_t /*plane_nodes*/, field_guard* guard)
            : queue_{e}, current_{e, WV_BUF_PREVIOUS, nodes, default_precision(), guard} {}
    bool wanted_now() const { return false; }  // whether a callback reads shows when it does (the handle tells the guard)
    template <typename Callback>
    void invoke(Callback& callback, size_t step, size_t steps) {
        callback(queue_, static_cast<const buffer&>(current_), step, steps);
    }

private:
    queue queue_;
    buffer current_;
};
template <typename Context, typename = void>
struct callback_bridge_for final {  // cl_mirror.h specialises this for contexts with a cl::Context member
    using type = handle_bridge;
    static type* make(const Context&, wv_engine* e, size_t nodes, size_t plane_nodes, field_guard* guard) {
        return new type{e, nodes, plane_nodes, guard};
    }
};

// a callback that cannot see the field: (step, steps) only, or wrapped in progress_only
template <typename Callback, typename = void>
struct takes_field : std::true_type {};
template <typename F>
struct takes_field<progress_only_t<F>, void> : std::false_type {};
template <typename Callback>
struct takes_field<Callback, decltype(std::declval<Callback&>()(size_t{}, size_t{}), void())> : std::false_type {};

template <typename Bridge, typename Callback>
void fire(Bridge& bridge, Callback& callback, size_t step, size_t steps, std::true_type) {
    bridge.invoke(callback, step, steps);
}
template <typename Bridge, typename F>
void fire(Bridge& bridge, progress_only_t<F>& callback, size_t step, size_t steps, std::false_type) {
    bridge.invoke(callback.f, step, steps);  // same arguments, batched: the field is not the step's
}
template <typename Bridge, typename Callback>
void fire(Bridge&, Callback& callback, size_t step, size_t steps, std::false_type) {
    callback(step, steps);
}

/// The device-resident run of `run_device`, AHEAD of per-step observers that may look at the field.
///
/// `observe(step, row)` is called once per completed step, in order, with the step's receiver samples; while it runs,
/// `guard.touch()` (any access through a guarded `buffer` handle, or the bridge) makes the engine's PREVIOUS buffer the
/// step's pre-update `current` -- what waveguide.h:121 hands to `post`.
///
/// Batches grow 1, 2, 4 ... `max_batch` while nobody looks.  Before a batch of more than one step the engine's state is
/// copied aside on the device (wv_checkpoint: 2 fields + filter memories, about the traffic of one step); a look at step i
/// of the batch rolls the engine back and re-runs i + 1 steps (bit-identical: the engine is deterministic), the rest of
/// the batch is abandoned and run again later, and the run proceeds one step at a time until `hold_steps` steps have
/// gone by unobserved.  An observer that looks at regular intervals (a visualiser that takes every k-th step) is met half way: the
/// interval between its last two looks is the guess for the next, and batches are cut so that the step it is expected to look at is
/// the LAST of its batch -- whose field is on the device as it is: no rollback, nothing run twice; a look elsewhere falls back to the above.
/// `wanted_now()` (the bridge knows a reader is attached) keeps it at one step per batch too.
/// Without room for the checkpoint the run simply stays at one step per batch, like the reference's loop.
template <typename Context, typename Mesh, typename It, typename MakeBridge, typename WantedNow, typename Observe>
size_t run_device_observed(const Context& cc, const Mesh& mesh, source_kind kind, size_t source_node, It begin, It end,
                           const std::vector<uint64_t>& receivers, field_guard& guard, MakeBridge&& make_bridge,
                           WantedNow&& wanted_now, Observe&& observe, const std::atomic_bool& keep_going,
                           size_t max_batch = 256, size_t hold_steps = 16, read_plan* plan = nullptr) {
    auto engine = make_engine(cc, mesh, default_precision());
    wv_engine* e = engine.get();
    std::vector<double> signal(begin, end);
    check(wv_set_source(e, (int)kind, source_node, signal.data(), signal.size()));
    check(wv_set_receivers(e, receivers.data(), (uint32_t)receivers.size()));
    make_bridge(e);
    run_stats& stats = last_run_stats();
    stats = run_stats{};
    const double t0 = seconds_now();
    // (`plan`: the receivers are what the callbacks have been seen to read -- `receivers` is the plan's own list and grows)
    size_t n_recv = receivers.size();
    size_t done_total = 0, batch = 1, hold = 0;
    double step_seconds = 0;  // of the batch before: batches are bounded by time as well as by count (batch_seconds())
    bool can_speculate = max_batch > 1, cancelled = false;
    constexpr size_t never = ~size_t{0};
    size_t last_look = never, period = 0;  // the step last looked at; the interval between the last two looks
    bool periodic = false;                 // the next look is expected at last_look + period
    std::vector<double> samples;
    const auto finish = [&] {
        uint64_t passes = 0;
        if (wv_query(e, WV_QUERY_PASSES, &passes) == WV_OK) stats.passes = passes;
        stats.steps = done_total;
        stats.seconds = seconds_now() - t0;
        guard.materialise = nullptr;
        if (plan) {
            plan->row = nullptr;
            stats.reads_served = plan->reads_served;
            stats.reads_missed = plan->reads_missed;
        }
    };
    try {
        while (done_total < signal.size() && keep_going) {
            size_t want = std::min(batch, signal.size() - done_total);
            if (periodic && can_speculate) {
                const size_t target = last_look + period;
                if (target >= done_total) {
                    want = std::min(std::min(max_batch, target + 1 - done_total), signal.size() - done_total);
                } else {
                    periodic = false;  // (the step came and went unobserved)
                }
            }
            if (wanted_now()) want = 1;
            want = std::max<size_t>(1, std::min(want, steps_within_budget(step_seconds, want)));
            if (plan && plan->grew) {  // what the callbacks read is recorded on the device from this batch on
                check(wv_set_receivers(e, receivers.data(), (uint32_t)receivers.size()));
                n_recv = receivers.size();
                plan->grew = false;
            }
            bool have_checkpoint = false;
            if (want > 1) {
                if (wv_checkpoint(e) == WV_OK) {
                    have_checkpoint = true;
                    ++stats.checkpoints;
                } else {  // no room for a copy of the fields: one step at a time from here on
                    can_speculate = false;
                    batch = want = 1;
                }
            }
            uint64_t done = 0;
            int32_t flag = 0;
            const double t_batch = seconds_now();
            check(wv_run(e, want, &done, &flag));
            if (done) {
                step_seconds = (seconds_now() - t_batch) / (double)done;
                ++stats.batches;
                samples.resize((size_t)done * n_recv);
                check(wv_fetch_receivers(e, done_total, done, samples.data()));
            }
            size_t fired = 0;
            bool looked = false, rewound = false;
            for (size_t i = 0; i < (size_t)done && !rewound; ++i) {
                // the fields of a batch that met a flag have advanced past the failing step: none of them is a step's
                guard.fresh = i + 1 == (size_t)done && flag == 0;
                guard.looked = false;
                guard.materialise = [&, i] {
                    check(wv_rollback(e));
                    uint64_t again = 0;
                    int32_t flag_again = 0;
                    check(wv_run(e, i + 1, &again, &flag_again));
                    if (again != i + 1 || flag_again)
                        throw engine_error("wayverb_amd: the re-run after a rollback did not reproduce the batch");
                    ++stats.rollbacks;
                    stats.steps_rerun += i + 1;
                    rewound = true;
                };
                if (plan) {
                    plan->row = samples.data() + i * n_recv;
                    plan->served_columns = n_recv;
                }
                observe(done_total + i, samples.data() + i * n_recv);
                ++fired;
                if (guard.looked) {
                    looked = true;
                    ++stats.fields_looked_at;
                    const size_t at = done_total + i;
                    if (last_look != never) {  // (the interval between the last two looks is the guess for the next one)
                        period = at - last_look;
                        periodic = true;
                    }
                    last_look = at;
                }
                // keep_going turned off by this step's observer: the reference tests it before every iteration (waveguide.h:80), so no later
                // step's callback may fire -- and the engine goes back to where this step left it, if the batch has taken it further
                if (!keep_going && i + 1 < (size_t)done) {
                    if (!rewound && have_checkpoint) {
                        check(wv_rollback(e));
                        uint64_t again = 0;
                        int32_t flag_again = 0;
                        check(wv_run(e, i + 1, &again, &flag_again));
                        if (again != i + 1 || flag_again)
                            throw engine_error("wayverb_amd: the re-run after a rollback did not reproduce the batch");
                        ++stats.rollbacks;
                        stats.steps_rerun += i + 1;
                    }
                    cancelled = true;
                    break;
                }
            }
            done_total += fired;
            if (cancelled) break;
            if (!rewound) {
                throw_for_flag(flag);
                if (done < want) break;
            }
            // pace: one step at a time while somebody is looking, doubling batches once nobody has for a while
            if (looked && periodic && period > 1) {
                batch = std::min(max_batch, period);  // (whatever the prediction leaves over is planned from the observer's own interval)
                hold = 0;
            } else if (looked) {
                batch = 1;
                hold = hold_steps;
            } else if (hold > fired) {
                hold -= fired;
            } else {
                hold = 0;
                if (can_speculate) batch = std::min(max_batch, batch * 2);
            }
        }
    } catch (...) {
        finish();
        throw;
    }
    finish();
    return done_total;
}

/// run<pre, post> with a source the device runs by itself (see `run`).
template <typename Context, typename Mesh, typename Source, typename step_postprocessor>
size_t run_ahead(const Context& cc, const Mesh& mesh, Source& pre, step_postprocessor& post, const std::atomic_bool& keep_going) {
    const size_t num_nodes = mesh.get_structure().get_condensed_nodes().size();
    field_guard guard;
    read_plan plan;
    std::unique_ptr<queue> q;
    std::unique_ptr<buffer> current;  // (wv_run has swapped the fields when `post` fires: the step's pre-update `current` is PREVIOUS now)
    const auto first = pre.begin(), last = pre.end();
    const size_t steps = run_device_observed(
            cc, mesh, static_cast<source_kind>(device_source<Source>::kind), pre.get_node(), first, last, plan.indices, guard,
            [&](wv_engine* e) {
                q.reset(new queue{e});
                current.reset(new buffer{e, WV_BUF_PREVIOUS, num_nodes, default_precision(), &guard, &plan});
            },
            [] { return false; },
            [&](size_t step, const double*) { post(*q, static_cast<const buffer&>(*current), step); }, keep_going, 256, 4, &plan);
    // the source object is left as the reference's loop leaves it: one sample taken per completed step, and one more by the call
    // that found keep_going off (waveguide.h:80: `pre` runs before keep_going is looked at)
    const size_t n = (size_t)std::distance(first, last);
    pre.advance(std::min(n, steps + (steps < n ? 1 : 0)));
    return steps;
}

/// canonical.h:29-88.  `callback(queue, buffer, step, ideal_steps)` fires once per completed step, in
/// order, after the directional receiver has taken its sample -- exactly the reference's lambda at
/// :66-69 -- with `buffer` the step's (pre-update) pressure field.
template <typename Context, typename Mesh, typename Vec3, typename Environment, typename Callback>
std::experimental::optional<band> canonical_impl(const Context& cc, const Mesh& mesh, double simulation_time,
                                                 const Vec3& source, const Vec3& receiver,
                                                 const Environment& environment, const std::atomic_bool& keep_going,
                                                 Callback&& callback) {
    const auto sample_rate = compute_sample_rate(mesh.get_descriptor(), environment.speed_of_sound);
    const size_t num_nodes = mesh.get_structure().get_condensed_nodes().size();
    const auto compute_mesh_index = [&](const Vec3& pt) {
        const auto ret = compute_index(mesh.get_descriptor(), pt);
        if (ret >= num_nodes || !(mesh.get_structure().get_condensed_nodes()[ret].boundary_type & WV_ID_INSIDE))
            throw std::runtime_error{"Source/receiver node position appears to be outside mesh."};
        return ret;
    };
    const size_t ideal_steps = (size_t)std::ceil(sample_rate * simulation_time);
    std::vector<float> input(ideal_steps, 0.0f);
    if (!input.empty())
        input.front() = (float)rectilinear_calibration_factor(mesh.get_descriptor().spacing,
                                                              environment.acoustic_impedance);
    const size_t receiver_index = compute_mesh_index(receiver);
    postprocessor::directional_receiver dr{mesh.get_descriptor(), sample_rate,
                                           environment.acoustic_impedance / environment.speed_of_sound, receiver_index};
    std::vector<uint64_t> nodes{receiver_index};
    for (auto n : dr.get_surrounding_nodes()) nodes.push_back(n);

    using callback_t = typename std::decay<Callback>::type;
    constexpr bool sees_field = takes_field<callback_t>::value;
    band ret{{}, sample_rate};
    ret.directional.reserve(ideal_steps);
    using bridge_t = typename callback_bridge_for<Context>::type;
    std::unique_ptr<bridge_t> bridge;
    field_guard guard;  // (declared before the bridge's handles use it; fresh and without a way back unless the run below says otherwise)
    const size_t plane_nodes = (size_t)mesh.get_descriptor().dimensions.x * (size_t)mesh.get_descriptor().dimensions.y;
    const auto record = [&](const double* row) {
        float nb[6];
        for (int k = 0; k < 6; ++k) nb[k] = (float)row[1 + k];
        ret.directional.emplace_back(dr.accumulate((float)row[0], nb));
    };
    size_t steps = 0;
    if (sees_field) {
        steps = run_device_observed(
                cc, mesh, source_kind::hard, compute_mesh_index(source), input.begin(), input.end(), nodes, guard,
                [&](wv_engine* e) { bridge.reset(callback_bridge_for<Context>::make(cc, e, num_nodes, plane_nodes, &guard)); },
                [&] { return bridge->wanted_now(); },
                [&](size_t step, const double* row) {
                    record(row);
                    fire(*bridge, callback, step, ideal_steps, std::integral_constant<bool, sees_field>{});
                },
                keep_going);
    } else {
        // the callback cannot see the field (or has promised not to look): whole batches, no checkpoints
        steps = run_device(
                cc, mesh, source_kind::hard, compute_mesh_index(source), input.begin(), input.end(), nodes,
                [&](size_t first, size_t n, const std::vector<double>& s, wv_engine* e) {
                    if (!bridge) bridge.reset(callback_bridge_for<Context>::make(cc, e, num_nodes, plane_nodes, &guard));
                    for (size_t i = 0; i < n; ++i) {
                        record(s.data() + i * 7);
                        fire(*bridge, callback, first + i, ideal_steps, std::integral_constant<bool, sees_field>{});
                    }
                },
                keep_going, 256);
    }
    if (steps != ideal_steps) return std::experimental::nullopt;
    return ret;
}
}  // namespace detail

/// canonical.h:100-127 for an already-built mesh.
template <typename Context, typename Vec3, typename Environment, typename PressureCallback>
std::experimental::optional<std::vector<bandpass_band>> canonical(
        const Context& cc, const mesh& mesh, const Vec3& source, const Vec3& receiver, const Environment& environment,
        const single_band_parameters& sim_params, double simulation_time, const std::atomic_bool& keep_going,
        PressureCallback&& pressure_callback) {
    if (auto ret = detail::canonical_impl(cc, mesh, simulation_time, source, receiver, environment, keep_going,
                                          pressure_callback)) {
        return std::vector<bandpass_band>{bandpass_band{std::move(*ret), util::make_range(0.0, sim_params.cutoff)}};
    }
    return std::experimental::nullopt;
}

/// canonical.h:100-127 as the reference declares it: `voxelised` is a voxels_and_mesh -- setup.h's, the
/// reference's own (mesh.h:52-58), anything with a `.mesh` member.
template <typename Context, typename VoxelsAndMesh, typename Vec3, typename Environment, typename PressureCallback>
auto canonical(const Context& cc, VoxelsAndMesh voxelised, const Vec3& source, const Vec3& receiver,
               const Environment& environment, const single_band_parameters& sim_params, double simulation_time,
               const std::atomic_bool& keep_going, PressureCallback&& pressure_callback)
        -> decltype((void)voxelised.mesh, std::experimental::optional<std::vector<bandpass_band>>{}) {
    if (auto ret = detail::canonical_impl(cc, voxelised.mesh, simulation_time, source, receiver, environment, keep_going,
                                          pressure_callback)) {
        return std::vector<bandpass_band>{bandpass_band{std::move(*ret), util::make_range(0.0, sim_params.cutoff)}};
    }
    return std::experimental::nullopt;
}

}  // namespace waveguide
}  // namespace wayverb
