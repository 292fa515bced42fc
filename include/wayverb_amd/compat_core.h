// wayverb_amd/compat_core.h -- the handful of `wayverb::core` / `util` names the waveguide mirror uses,
// for translation units that do NOT have the reference's own headers.
//
// Inside the reference tree (src/combined, bin/...), the real headers define these names and must
// win: define WAYVERB_AMD_HAVE_REFERENCE_CORE before including any wayverb_amd header, after having
// included
//     core/exceptions.h          core::exceptions::value_is_nan / value_is_inf      (:9-30)
//     core/environment.h         core::environment, get_ambient_density             (:6-17)
//     core/callback_accumulator.h  core::callback_accumulator                       (:8-36)
//     core/cl/common.h           core::compute_context (cl::Context + cl::Device)   (:13-22)
//     utilities/range.h          util::range                                        (:11-48)
// and nothing in this file is declared (tests/cpp/combined_shape_test.cpp compiles that way).
#pragma once

#ifndef WAYVERB_AMD_HAVE_REFERENCE_CORE

#include <stdexcept>
#include <utility>
#include <vector>

namespace util {
template <typename T>
class range final {  // utilities/range.h:11-48 (the members the waveguide path touches)
public:
    using value_type = T;
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

namespace exceptions {  // src/core/include/core/exceptions.h:9-30
class exception : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};
class suspicious_value : public exception {
public:
    using exception::exception;
};
class value_is_nan final : public suspicious_value {
public:
    using suspicious_value::suspicious_value;
};
class value_is_inf final : public suspicious_value {
public:
    using suspicious_value::suspicious_value;
};
}  // namespace exceptions

struct environment final {  // src/core/include/core/environment.h:6-13
    double speed_of_sound{340.0};
    double acoustic_impedance{400.0};
};
constexpr double get_ambient_density(const environment& s) { return s.acoustic_impedance / s.speed_of_sound; }

/// Stand-in for core::compute_context (src/core/include/core/cl/common.h:13-22): which HIP device.
/// (`run` / `canonical` are templates on the context type: the reference's OpenCL context is accepted
/// as well and selects the calling thread's current HIP device.)
struct compute_context final {
    int device{-1};
};

template <typename T, typename Ret = typename T::return_type>
class callback_accumulator final {  // src/core/include/core/callback_accumulator.h:8-28
public:
    callback_accumulator(T t) : postprocessor_{std::move(t)} {}
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
template <typename T>
auto make_callback_accumulator(T t) {
    return callback_accumulator<T>{std::move(t)};
}

}  // namespace core
}  // namespace wayverb

#endif  // WAYVERB_AMD_HAVE_REFERENCE_CORE
