// tests/cpp/combined_shape_test.cpp -- does the mirror serve src/combined AS IT IS WRITTEN?
//
// This translation unit is laid out like one of the reference's own: the OpenCL C++ bindings and the
// `wayverb::core` names come first (here: a minimal restatement of their SHAPES -- the reference's
// headers are not available to this build -- each with the file:line it stands for), then
// WAYVERB_AMD_HAVE_REFERENCE_CORE tells wayverb_amd's headers to define none of them, then follow
//   - `combined::waveguide_base` / `concrete_waveguide<T>::run`, call shape of
//         src/combined/include/combined/waveguide_base.h:33-60, src/combined/src/waveguide_base.cpp:8-47
//     (pressure callback type-erased on cl::CommandQueue& / const cl::Buffer&, voxels_and_mesh passed on
//     by value, glm::vec3 positions),
//   - the waveguide leg of `combined::engine::impl::run`, call shape of src/combined/src/engine.cpp:150-173
//     (generic lambda; core::read_from_buffer<float>(queue, buffer) when a listener is attached).
// Exit code 0 = all assertions held; 3 = no OpenCL GPU device; 2 = exception (e.g. no HIP device).
#define __CL_ENABLE_EXCEPTIONS
#include <CL/cl.hpp>  // src/core/include/core/cl/include.h:3-4

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <vector>

// ---- shapes of the reference's own headers --------------------------------------------------------
namespace glm {
struct vec3 {  // glm/vec3.hpp: what src/combined passes as positions
    float x, y, z;
};
}  // namespace glm

namespace util {
template <typename T>
class range final {  // utilities/range.h:11-48
public:
    constexpr range() : min_{0}, max_{0} {}
    constexpr range(T a, T b) : min_{a < b ? a : b}, max_{a < b ? b : a} {}
    constexpr T get_min() const { return min_; }
    constexpr T get_max() const { return max_; }

private:
    T min_, max_;
};
template <typename T>
constexpr range<T> make_range(T a, T b) {
    return range<T>{a, b};
}
}  // namespace util

namespace wayverb {
namespace core {
namespace exceptions {  // core/exceptions.h:9-30
struct exception : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct suspicious_value : exception {
    using exception::exception;
};
struct value_is_nan final : suspicious_value {
    using suspicious_value::suspicious_value;
};
struct value_is_inf final : suspicious_value {
    using suspicious_value::suspicious_value;
};
}  // namespace exceptions

struct environment final {  // core/environment.h:6-9
    double speed_of_sound{340.0};
    double acoustic_impedance{400.0};
};

class compute_context final {  // core/cl/common.h:13-22: an OpenCL context and one of its devices
public:
    compute_context() {
        std::vector<cl::Platform> platforms;
        cl::Platform::get(&platforms);
        for (auto& p : platforms) {
            std::vector<cl::Device> devices;
            try {
                p.getDevices(CL_DEVICE_TYPE_GPU, &devices);
            } catch (const cl::Error&) {
                continue;
            }
            if (!devices.empty()) {
                device = devices.front();
                context = cl::Context{device};
                return;
            }
        }
        throw std::runtime_error{"no OpenCL GPU device"};
    }
    cl::Context context;
    cl::Device device;
};

template <typename T>
size_t items_in_buffer(const cl::Buffer& buffer) {  // core/cl/common.h:29-32
    return buffer.getInfo<CL_MEM_SIZE>() / sizeof(T);
}
template <typename T>
std::vector<T> read_from_buffer(cl::CommandQueue& queue, const cl::Buffer& buffer) {  // core/cl/common.h:34-40
    std::vector<T> ret(items_in_buffer<T>(buffer));
    cl::copy(queue, buffer, ret.begin(), ret.end());
    return ret;
}

template <typename T, typename Ret = typename T::return_type>
class callback_accumulator final {  // core/callback_accumulator.h:8-28
public:
    template <typename... Ts>
    callback_accumulator(Ts&&... ts) : postprocessor_{std::forward<Ts>(ts)...} {}
    template <typename... Ts>
    void operator()(Ts&&... ts) {
        output_.emplace_back(postprocessor_(std::forward<Ts>(ts)...));
    }
    const std::vector<Ret>& get_output() const { return output_; }

private:
    std::vector<Ret> output_;
    T postprocessor_;
};
}  // namespace core
}  // namespace wayverb

// ---- the engine's headers, told that `wayverb::core` is already there ------------------------------
#define WAYVERB_AMD_HAVE_REFERENCE_CORE
#include "wayverb_amd/cl_mirror.h"
#include "wayverb_amd/setup.h"

#define REQUIRE(cond)                                                        \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("REQUIRE failed: %s (line %d)\n", #cond, __LINE__);  \
            std::exit(1);                                                    \
        }                                                                    \
    } while (0)

namespace wayverb {
namespace combined {

class waveguide_base {  // waveguide_base.h:33-60
public:
    virtual ~waveguide_base() noexcept = default;
    virtual std::unique_ptr<waveguide_base> clone() const = 0;
    virtual double compute_sampling_frequency() const = 0;
    virtual std::experimental::optional<std::vector<waveguide::bandpass_band>> run(
            const core::compute_context& cc, const waveguide::voxels_and_mesh& voxelised, const glm::vec3& source,
            const glm::vec3& receiver, const core::environment& environment, double simulation_time,
            const std::atomic_bool& keep_going,
            std::function<void(cl::CommandQueue& queue, const cl::Buffer& buffer, size_t step, size_t steps)>
                    pressure_callback) = 0;
};

template <typename T>
class concrete_waveguide final : public waveguide_base {  // waveguide_base.cpp:8-47
public:
    concrete_waveguide(const T& t) : sim_params_{t} {}
    std::unique_ptr<waveguide_base> clone() const override { return std::make_unique<concrete_waveguide>(*this); }
    double compute_sampling_frequency() const override { return waveguide::compute_sampling_frequency(sim_params_); }
    std::experimental::optional<std::vector<waveguide::bandpass_band>> run(
            const core::compute_context& cc, const waveguide::voxels_and_mesh& voxelised, const glm::vec3& source,
            const glm::vec3& receiver, const core::environment& environment, double simulation_time,
            const std::atomic_bool& keep_going,
            std::function<void(cl::CommandQueue& queue, const cl::Buffer& buffer, size_t step, size_t steps)>
                    pressure_callback) override {
        return waveguide::canonical(cc, std::move(voxelised), source, receiver, environment, sim_params_, simulation_time,
                                    keep_going, std::move(pressure_callback));
    }

private:
    T sim_params_;
};

std::unique_ptr<waveguide_base> make_waveguide_ptr(const waveguide::single_band_parameters& t) {
    return std::make_unique<concrete_waveguide<waveguide::single_band_parameters>>(std::move(t));
}

}  // namespace combined
}  // namespace wayverb

using namespace wayverb;

int main() {
    try {
        const core::compute_context compute_context_{};  // the ray tracer's context: OpenCL
        // a 6 x 5 x 4 m box room, two materials -- what engine.cpp:98-103 builds from the scene
        core::scene_data scene;
        for (int i = 0; i < 8; ++i)
            scene.vertices.push_back(core::scene_vertex{(i & 1) ? 6.0f : 0.0f, (i & 2) ? 5.0f : 0.0f, (i & 4) ? 4.0f : 0.0f, 0.0f});
        const uint32_t quads[6][4] = {{0, 1, 3, 2}, {4, 6, 7, 5}, {0, 4, 5, 1}, {2, 3, 7, 6}, {0, 2, 6, 4}, {1, 5, 7, 3}};
        for (uint32_t q = 0; q < 6; ++q) {
            const uint32_t s = q == 0 ? 1u : 0u;
            scene.triangles.push_back(core::triangle{s, quads[q][0], quads[q][1], quads[q][2]});
            scene.triangles.push_back(core::triangle{s, quads[q][0], quads[q][2], quads[q][3]});
        }
        scene.surfaces.push_back(core::surface_absorption{{0.05, 0.05, 0.05, 0.05, 0.05, 0.05, 0.05, 0.05}});
        scene.surfaces.push_back(core::surface_absorption{{0.30, 0.30, 0.45, 0.65, 0.56, 0.59, 0.71, 0.71}});
        const core::environment environment_{};
        const glm::vec3 source_{1.5f, 1.5f, 1.5f}, receiver_{4.0f, 3.0f, 2.0f};
        const auto waveguide_ = combined::make_waveguide_ptr(waveguide::single_band_parameters{150.0, 0.6});
        const auto voxels_and_mesh_ = waveguide::compute_voxels_and_mesh(
                compute_context_, scene, receiver_, waveguide_->compute_sampling_frequency(), environment_.speed_of_sound);
        const std::atomic_bool keep_going{true};

        // ---- engine.cpp:150-173 ---------------------------------------------------------------------
        struct {
            std::vector<std::pair<std::vector<float>, double>> got;
            bool attached = true;
            bool empty() const { return !attached; }
            void operator()(std::vector<float> p, double distance) { got.emplace_back(std::move(p), distance); }
        } waveguide_node_pressures_changed_;
        std::vector<double> progress;
        auto engine_state_changed_ = [&](int, double p) { progress.push_back(p); };
        const double max_stochastic_time = 0.12;

        auto waveguide_output = waveguide_->run(
                compute_context_, voxels_and_mesh_, source_, receiver_, environment_, max_stochastic_time, keep_going,
                [&](auto& queue, const auto& buffer, auto step, auto steps) {
                    //  If there are node pressure listeners.
                    if (!waveguide_node_pressures_changed_.empty()) {
                        auto pressures = core::read_from_buffer<float>(queue, buffer);
                        const auto time = step / waveguide_->compute_sampling_frequency();
                        const auto distance = time * environment_.speed_of_sound;
                        waveguide_node_pressures_changed_(std::move(pressures), distance);
                    }
                    engine_state_changed_(0, step / (steps - 1.0));
                });
        REQUIRE(keep_going && waveguide_output);

        // ---- what must hold -------------------------------------------------------------------------
        const auto& band = waveguide_output->front().band;
        const size_t steps = band.directional.size();
        const auto& d = voxels_and_mesh_.mesh.get_descriptor();
        REQUIRE(steps == (size_t)std::ceil(band.sample_rate * max_stochastic_time) && steps > 40);
        REQUIRE(progress.size() == steps && progress.front() == 0.0 && progress.back() == 1.0);
        REQUIRE(waveguide_node_pressures_changed_.got.size() == steps);
        const size_t receiver_index = waveguide::compute_index(d, receiver_);
        const size_t source_index = waveguide::compute_index(d, source_);
        for (size_t i = 0; i < steps; ++i) {
            const auto& p = waveguide_node_pressures_changed_.got[i].first;
            REQUIRE(p.size() == waveguide::compute_num_nodes(d));
            // the cl::Buffer held step i's field: the receiver's sample of that step is in it ...
            REQUIRE(p[receiver_index] == band.directional[i].pressure);
            // ... and the hard source's sample of that step (calibrated impulse, then zeros: canonical.h:55-63)
            if (i == 0) REQUIRE(p[source_index] == (float)waveguide::rectilinear_calibration_factor(d.spacing, 400.0));
            if (i > 0) REQUIRE(p[source_index] == 0.0f);
        }
        // the same run with the engine's own handles and the same callback text gives the same fields
        std::vector<std::vector<float>> fields;
        // (a context with OpenCL members always takes the mirror; the handle form is reached through
        // a context without them)
        struct hip_context {
        } hip_cc;
        const auto handles = waveguide::canonical(hip_cc, voxels_and_mesh_, source_, receiver_, environment_,
                                                  waveguide::single_band_parameters{150.0, 0.6}, max_stochastic_time, keep_going,
                                                  [&](auto& queue, const auto& buffer, auto, auto) {
                                                      fields.push_back(core::read_from_buffer<float>(queue, buffer));
                                                  });
        REQUIRE(bool(handles) && fields.size() == steps);
        for (size_t i = 0; i < steps; ++i)
            REQUIRE(std::memcmp(fields[i].data(), waveguide_node_pressures_changed_.got[i].first.data(),
                                fields[i].size() * sizeof(float)) == 0);
        // no listener attached + the predicate set: nothing is mirrored, the run is the same
        waveguide_node_pressures_changed_.attached = false;
        waveguide::cl_mirror_wanted() = [&] { return !waveguide_node_pressures_changed_.empty(); };
        const auto quiet = waveguide_->run(compute_context_, voxels_and_mesh_, source_, receiver_, environment_,
                                           max_stochastic_time, keep_going,
                                           [&](auto&, const auto&, auto, auto) {});
        waveguide::cl_mirror_wanted() = nullptr;
        REQUIRE(bool(quiet) && std::memcmp(quiet->front().band.directional.data(), band.directional.data(),
                                           steps * sizeof(band.directional[0])) == 0);
    } catch (const cl::Error& e) {
        std::printf("OpenCL error: %s (%d)\n", e.what(), e.err());
        return 3;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return std::strstr(e.what(), "no OpenCL GPU device") ? 3 : 2;
    }
    std::puts("COMBINED SHAPE OK");
    return 0;
}
