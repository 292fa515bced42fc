// postprocess.cpp -- receiver traces -> audio: SURVEY.md 8(f) rank 3, host side.
//
// Replaces, for the waveguide's output:
//   attenuate / make_attenuate_mapper        src/waveguide/include/waveguide/attenuator.h:13-49
//   attenuation(microphone, incident)        src/core/src/attenuator/microphone.cpp:18-25
//   postprocess(band, method, Z, out_sr)     src/waveguide/include/waveguide/postprocess.h:57-72
//   postprocess(bandpass_bands, ...)         postprocess.h:74-126
//   adjust_sampling_rate                     src/waveguide/src/config.cpp:29-56
//   frequency_domain::filter::run            src/frequency_domain/src/filter.cpp:22-47
//   compute_{lopass,hipass,bandpass}_magnitude, band edges
//                                            src/frequency_domain/src/envelope.cpp:20-115
//   best_fft_length                          src/frequency_domain/include/frequency_domain/multiband_filter.h:35-43
//
// Two third-party pieces sit under the reference here and neither is in its tree:
//  - fftw3f (single-precision r2c/c2r): any correct transform gives the same result to rounding;
//    the transforms below run in double and the result is rounded to float once.
//  - libsamplerate, src_simple(SRC_SINC_BEST_QUALITY): a band-limited windowed-sinc interpolator
//    with a fixed coefficient table.  Its table is not reproducible without the library, so this
//    file designs its own Kaiser-windowed sinc to the converter's published figures (pass band
//    96 % of the narrower Nyquist band, stop band from 100 %, >= 140 dB rejection): same
//    length, gain and band limits, different coefficients.  Parity of this stage is therefore
//    "unpinned"; it is tested against the analytic behaviour of an ideal band-limited resampler.
// What IS pinned: the chain as a whole with microphone capsules, at resampling ratio 1 -- the band energies the reference's own
// bin/mic_test printed (bin/mic_test/output/*/waveguide.txt) are reproduced to 6e-4 (tests/test_mic_test_reference.py).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <thread>
#include <limits>
#include <vector>

#include "../../include/wayverb_amd.h"

namespace wv {
int fail_with(int code, const std::string& msg);  // engine.hip
}

namespace {

using cd = std::complex<double>;
constexpr double kPi = 3.14159265358979323846;

void fft_pow2(std::vector<cd>& x, int sign) {
    const size_t n = x.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(x[i], x[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = sign * 2.0 * kPi / (double)len;
        const cd wl(std::cos(ang), std::sin(ang));
        for (size_t i = 0; i < n; i += len) {
            cd w(1, 0);
            for (size_t k = 0; k < len / 2; ++k) {
                if ((k & 63) == 0) w = cd(std::cos(ang * (double)k), std::sin(ang * (double)k));
                const cd u = x[i + k], v = x[i + k + len / 2] * w;
                x[i + k] = u + v;
                x[i + k + len / 2] = u - v;
                w *= wl;
            }
        }
    }
}

// envelope.cpp:20-46 with l levels of steepening
double band_edge_impl(double p, double P, unsigned l) {
    return l != 0 ? std::sin(kPi * band_edge_impl(p, P, l - 1) / 2) : (((p / P) + 1) / 2);
}
double lower_band_edge(double p, double P, unsigned l) {
    if (P == 0) return 0 <= p ? 1.0 : 0.0;
    return std::pow(std::sin(kPi * band_edge_impl(p, P, l) / 2), 2.0);
}
double upper_band_edge(double p, double P, unsigned l) {
    if (P == 0) return p < 0 ? 1.0 : 0.0;
    return std::pow(std::cos(kPi * band_edge_impl(p, P, l) / 2), 2.0);
}
// envelope.cpp:73-112
double lopass_magnitude(double frequency, double edge, double width_factor, unsigned l) {
    const double w = edge * width_factor;
    if (frequency < edge - w) return 1;
    if (frequency < edge + w) return upper_band_edge(frequency - edge, w, l);
    return 0;
}
double hipass_magnitude(double frequency, double edge, double width_factor, unsigned l) {
    const double w = edge * width_factor;
    if (frequency < edge - w) return 0;
    if (frequency < edge + w) return lower_band_edge(frequency - edge, w, l);
    return 1;
}

// multiband_filter.h:35-43
size_t best_fft_length(size_t n) { return (size_t)std::pow(2.0, std::ceil(std::log2((double)n))); }

// frequency_domain::filter{best_fft_length(n) << 2}.run(sig, sig+n, sig, bin *= magnitude(freq)):
// zero-padded real transform, per-bin real gain evaluated at float(i) / N, inverse, first n kept.
template <typename Gain>
void frequency_domain_filter(float* sig, size_t n, Gain gain) {
    if (n == 0) return;
    const size_t N = best_fft_length(n) << 2;
    std::vector<cd> x(N, cd(0, 0));
    for (size_t i = 0; i < n; ++i) x[i] = (double)sig[i];
    fft_pow2(x, -1);
    for (size_t i = 0; i <= N / 2; ++i) {
        const float freq = (float)i / (float)N;  // filter.cpp:31
        const double g = (double)(float)gain((double)freq);  // the callbacks cast to float
        x[i] *= g;
        if (i != 0 && i != N / 2) x[N - i] = std::conj(x[i]);
    }
    fft_pow2(x, +1);
    for (size_t i = 0; i < n; ++i) sig[i] = (float)(x[i].real() / (double)N);
}

// modified Bessel function of the first kind, order 0
double bessel_i0(double x) {
    double sum = 1, term = 1;
    const double q = x * x / 4;
    for (int k = 1; k < 500; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}

// config.cpp:29-56.  ratio = out_sr / in_sr; output length (size_t)(ratio * n); unit-gain
// band-limited interpolation, then the reference's 1 / ratio level correction.
std::vector<float> adjust_sampling_rate(const float* data, size_t n, double in_sr, double out_sr) {
    if (!(in_sr && out_sr)) throw std::runtime_error("Sample rate of 0 gives few hints about how to proceed.");
    const double ratio = out_sr / in_sr;
    std::vector<float> out((size_t)(ratio * (double)n), 0.0f);
    if (n == 0) return out;
    const double band = std::min(1.0, ratio);  // narrower Nyquist band, relative to the input's
    const double fc = 0.98 * band;             // -6 dB point: pass to 0.96, stop from 1.00
    const double beta = 0.1102 * (140.0 - 8.7);
    const double half = std::ceil((140.0 - 8.0) / (2.285 * 2 * kPi * 0.02 * band) / 2);  // taps / 2, input samples
    // Kaiser window on a fine table (linear interpolation error ~1e-9, below float rounding)
    constexpr int kTable = 1 << 16;
    std::vector<double> window(kTable + 2);
    const double i0b = bessel_i0(beta);
    for (int i = 0; i <= kTable; ++i) {
        const double u = (double)i / kTable;
        window[i] = bessel_i0(beta * std::sqrt(std::max(0.0, 1 - u * u))) / i0b;
    }
    window[kTable + 1] = window[kTable];
    const double volume_scale = 1 / ratio;
    auto work = [&](size_t first, size_t last) {
        for (size_t m = first; m < last; ++m) {
            const double t = (double)m / ratio;  // position on the input grid
            const long lo = std::max(0L, (long)std::ceil(t - half));
            const long hi = std::min((long)n - 1, (long)std::floor(t + half));
            double acc = 0;
            for (long k = lo; k <= hi; ++k) {
                const double x = t - (double)k;
                const double pos = std::min(1.0, std::abs(x) / half) * kTable;
                const int cell = (int)pos;
                const double win = window[cell] + (pos - cell) * (window[cell + 1] - window[cell]);
                const double arg = kPi * fc * x;
                const double sinc = arg == 0 ? 1.0 : std::sin(arg) / arg;
                acc += (double)data[k] * fc * sinc * win;
            }
            out[m] = (float)(acc * volume_scale);
        }
    };
    const size_t n_threads = std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), out.size() / 1024 + 1);
    std::vector<std::thread> pool;
    const size_t chunk = (out.size() + n_threads - 1) / n_threads;
    for (size_t t = 0; t < n_threads; ++t) {
        const size_t first = t * chunk, last = std::min(out.size(), first + chunk);
        if (first < last) pool.emplace_back(work, first, last);
    }
    for (auto& th : pool) th.join();
    return out;
}

// attenuator.h:13-30 in the reference's float arithmetic
float attenuate(int method, const float pointing[3], float shape, float Z, const wv_directional_output& s) {
    if (method == WV_ATTENUATOR_NULL) return s.pressure;
    const float ix = -s.intensity[0], iy = -s.intensity[1], iz = -s.intensity[2];
    const float l = std::sqrt(ix * ix + iy * iy + iz * iz);
    float att = 0;
    if (l) att = (1 - shape) + shape * (pointing[0] * (ix / l) + pointing[1] * (iy / l) + pointing[2] * (iz / l));
    const float intensity = l * (att * att);  // |-I| == |I|
    return std::copysign(std::sqrt(intensity * Z), s.pressure);
}

// ---- HRTF attenuator (src/core/src/attenuator/hrtf.cpp, vector_look_up_table.h, az_el.cpp, orientation.cpp)
struct Orientation {  // orientation.cpp:10-33: right-handed, +y up, -z forwards
    float m[3][3];    // columns x_axis, y_axis, z_axis of get_matrix()
};

void normalize3(float v[3]) {
    const float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int k = 0; k < 3; ++k) v[k] /= l;
}
void cross3(const float a[3], const float b[3], float out[3]) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

Orientation make_orientation(const float pointing_in[3], const float up_in[3]) {
    float pointing[3] = {pointing_in[0], pointing_in[1], pointing_in[2]}, up[3] = {up_in[0], up_in[1], up_in[2]};
    normalize3(pointing);
    normalize3(up);
    Orientation o;
    float z_axis[3] = {-pointing[0], -pointing[1], -pointing[2]}, x_axis[3], y_axis[3];
    cross3(up, z_axis, x_axis);
    normalize3(x_axis);
    cross3(z_axis, x_axis, y_axis);
    normalize3(y_axis);
    for (int k = 0; k < 3; ++k) {
        o.m[k][0] = x_axis[k];
        o.m[k][1] = y_axis[k];
        o.m[k][2] = z_axis[k];
    }
    return o;
}

// transform (orientation.cpp:41-43): inverse(get_matrix()) * vec, general 3x3 inverse by cofactors
void to_object_space(const Orientation& o, const float v[3], float out[3]) {
    const float(*a)[3] = o.m;
    const float c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2],
                c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    const float det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
    const float inv[3][3] = {
            {c00 / det, (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det, (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det},
            {c01 / det, (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det, (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det},
            {c02 / det, (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det, (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det}};
    for (int r = 0; r < 3; ++r) out[r] = inv[r][0] * v[0] + inv[r][1] * v[1] + inv[r][2] * v[2];
}

// almost_equal(a, b, ulps) of src/core/include/core/almost_equal.h, for the poles of az_el.cpp:64-71
bool almost_equal_ulps(float x, float y, int ulp) {
    return std::abs(x - y) < std::numeric_limits<float>::epsilon() * std::abs(x + y) * (float)ulp ||
           std::abs(x - y) < std::numeric_limits<float>::min();
}

struct TableIndex {
    size_t azimuth, elevation;
};

// vector_look_up_table<T, az_num, el_num>::index (vector_look_up_table.h:50-110)
TableIndex table_index(const float dir[3], uint32_t az_num, uint32_t el_num) {
    float azimuth = std::atan2(dir[0], -dir[2]);  // az_el.cpp:56-62
    const float elevation = std::asin(dir[1]);
    if (almost_equal_ulps(elevation, (float)(-kPi / 2), 10) || almost_equal_ulps(elevation, (float)(kPi / 2), 10)) azimuth = 0;
    const auto degrees = [](float radians) { return (float)(radians * 180 / kPi); };
    const double az_angle = 360.0 / az_num, el_angle = 180.0 / (el_num + 1);
    double az = (double)degrees(-azimuth) + az_angle / 2;
    while (az < 0) az += 360;
    double el = (double)degrees(elevation) + 90 + el_angle / 2;
    while (el < 0) el += 360;
    const size_t adjusted = (size_t)(el / el_angle) % (2 * ((size_t)el_num + 1));
    if ((size_t)el_num + 1 < adjusted) throw std::runtime_error("Elevation out of range.");
    return TableIndex{(size_t)(az / az_angle) % az_num, std::max<size_t>(1, std::min<size_t>(el_num, adjusted)) - 1};
}

// attenuation(hrtf, incident) (hrtf.cpp:121-133): the table's 8 band energies for the direction `incident`
// comes from, seen from the oriented head; zeros for a zero vector
void hrtf_attenuation(const wv_hrtf_table& t, const Orientation& o, int channel, const float incident[3], float bands[8]) {
    const float l = std::sqrt(incident[0] * incident[0] + incident[1] * incident[1] + incident[2] * incident[2]);
    for (int b = 0; b < 8; ++b) bands[b] = 0;
    if (!l) return;
    const float unit[3] = {incident[0] / l, incident[1] / l, incident[2] / l};
    float local[3];
    to_object_space(o, unit, local);
    const TableIndex i = table_index(local, t.az_num, t.el_num);
    const double* e = t.energy + (((size_t)i.azimuth * t.el_num + i.elevation) * 2 + (channel ? 1 : 0)) * 8;
    for (int b = 0; b < 8; ++b) bands[b] = (float)e[b];
}

// attenuate (attenuator.h:13-23) with a bands_type attenuation: 8 values per sample
void attenuate_hrtf(const wv_hrtf_table& t, const Orientation& o, int channel, float Z, const wv_directional_output& s,
                    float out[8]) {
    const float minus[3] = {-s.intensity[0], -s.intensity[1], -s.intensity[2]};
    float att[8];
    hrtf_attenuation(t, o, channel, minus, att);
    const float l = std::sqrt(s.intensity[0] * s.intensity[0] + s.intensity[1] * s.intensity[1] + s.intensity[2] * s.intensity[2]);
    for (int b = 0; b < 8; ++b) {
        const float intensity = l * std::pow(att[b], 2.0f);
        out[b] = std::copysign(std::sqrt(intensity * Z), s.pressure);
    }
}

// hrtf_data::hrtf_band_params (multiband.h:26-36): 9 edges of the 8 bands over 20 Hz .. 20 kHz, relative to the
// sample rate, and the crossover width factor for overlap 1 (envelope.cpp:5-16)
void hrtf_band_params(double sample_rate, double edges[9], double* width_factor) {
    for (int i = 0; i < 9; ++i) edges[i] = 20.0 * std::pow(20000.0 / 20.0, i / 8.0) / sample_rate;
    const double base = std::pow(20000.0 / 20.0, 1.0 / 8);
    *width_factor = (base - 1) / (base + 1);
}

// multiband_filter_and_mixdown (mixdown.h:17-26, multiband_filter.h:47-90): band i of every sample is band-passed
// to band i's range, then the 8 bands of a sample are summed
std::vector<float> multiband_filter_and_mixdown(std::vector<float>& bands, size_t n, double sample_rate) {
    double edges[9], wf;
    hrtf_band_params(sample_rate, edges, &wf);
    std::vector<float> column(n);
    for (int b = 0; b < 8; ++b) {
        for (size_t i = 0; i < n; ++i) column[i] = bands[i * 8 + b];
        frequency_domain_filter(column.data(), n, [&](double f) {
            return lopass_magnitude(f, edges[b + 1], wf, 0) * hipass_magnitude(f, edges[b], wf, 0);
        });
        for (size_t i = 0; i < n; ++i) bands[i * 8 + b] = column[i];
    }
    std::vector<float> out(n);
    for (size_t i = 0; i < n; ++i) {
        float s = 0;
        for (int b = 0; b < 8; ++b) s += bands[i * 8 + b];
        out[i] = s;
    }
    return out;
}

bool table_ok(const wv_hrtf_table* t) { return t && t->energy && t->az_num >= 1 && t->el_num >= 1 && (t->el_num % 2) == 1; }

}  // namespace

extern "C" int wv_attenuate(int32_t method, const float pointing[3], float shape, float acoustic_impedance,
                            const wv_directional_output* in, uint64_t n, float* out) {
    if ((n && (!in || !out)) || (method != WV_ATTENUATOR_NULL && method != WV_ATTENUATOR_MICROPHONE) ||
        (method == WV_ATTENUATOR_MICROPHONE && !pointing))
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument");
    if (acoustic_impedance < 300 || 500 <= acoustic_impedance)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "Acoustic impedance outside expected range.");
    const float clamped = std::min(1.0f, std::max(0.0f, shape));  // microphone.cpp:8-10
    for (uint64_t i = 0; i < n; ++i) out[i] = attenuate(method, pointing, clamped, acoustic_impedance, in[i]);
    return WV_OK;
}

extern "C" int wv_adjust_sampling_rate(const float* in, uint64_t n, double in_sample_rate, double out_sample_rate,
                                       float* out, uint64_t capacity, uint64_t* n_out) {
    if ((n && !in) || !n_out) return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument");
    try {
        if (!(in_sample_rate && out_sample_rate))
            throw std::runtime_error("Sample rate of 0 gives few hints about how to proceed.");
        *n_out = (uint64_t)((out_sample_rate / in_sample_rate) * (double)n);
        if (!out || capacity < *n_out) return WV_OK;  // size query
        const std::vector<float> r = adjust_sampling_rate(in, n, in_sample_rate, out_sample_rate);
        std::copy(r.begin(), r.end(), out);
    } catch (const std::exception& e) {
        return wv::fail_with(WV_E_INVALID_ARGUMENT, e.what());
    }
    return WV_OK;
}

extern "C" int wv_frequency_domain_filter(float* signal, uint64_t n, int32_t kind, double edge_lo, double edge_hi,
                                          double width_factor, uint32_t steepness) {
    if ((n && !signal) || width_factor < 0 || 1 < width_factor)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "Width_factor must be between 0 and 1.");
    switch (kind) {
        case WV_FILTER_LOPASS:
            frequency_domain_filter(signal, n, [&](double f) { return lopass_magnitude(f, edge_hi, width_factor, steepness); });
            break;
        case WV_FILTER_HIPASS:
            frequency_domain_filter(signal, n, [&](double f) { return hipass_magnitude(f, edge_lo, width_factor, steepness); });
            break;
        case WV_FILTER_BANDPASS:
            frequency_domain_filter(signal, n, [&](double f) {
                return lopass_magnitude(f, edge_hi, width_factor, steepness) *
                       hipass_magnitude(f, edge_lo, width_factor, steepness);
            });
            break;
        default: return wv::fail_with(WV_E_INVALID_ARGUMENT, "unknown filter kind");
    }
    return WV_OK;
}

extern "C" int wv_postprocess_waveguide(const wv_waveguide_band* bands, uint32_t n_bands, int32_t method,
                                        const float pointing[3], float shape, float acoustic_impedance,
                                        double output_sample_rate, float* out, uint64_t capacity, uint64_t* n_out) {
    if ((n_bands && !bands) || !n_out) return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument");
    try {
        std::vector<float> ret;
        for (uint32_t bi = 0; bi < n_bands; ++bi) {
            const wv_waveguide_band& band = bands[bi];
            std::vector<float> attenuated(band.n);
            const int rc = wv_attenuate(method, pointing, shape, acoustic_impedance, band.directional, band.n,
                                        attenuated.data());
            if (rc != WV_OK) return rc;
            std::vector<float> processed =
                    adjust_sampling_rate(attenuated.data(), attenuated.size(), band.sample_rate, output_sample_rate);
            // band-pass at this band's valid range (postprocess.h:86-101)
            const double lo = band.valid_hz_min / output_sample_rate, hi = band.valid_hz_max / output_sample_rate;
            frequency_domain_filter(processed.data(), processed.size(), [&](double f) {
                return lopass_magnitude(f, hi, 0.1, 0) * hipass_magnitude(f, lo, 0.1, 0);
            });
            ret.resize(std::max(ret.size(), processed.size()), 0.0f);
            for (size_t i = 0; i < processed.size(); ++i) ret[i] += processed[i];
        }
        // DC block at 10 Hz (postprocess.h:108-122)
        const double dc_block = 10.0 / output_sample_rate;
        frequency_domain_filter(ret.data(), ret.size(), [&](double f) { return hipass_magnitude(f, dc_block, 0.9, 0); });
        *n_out = ret.size();
        if (out && capacity >= ret.size()) std::copy(ret.begin(), ret.end(), out);
    } catch (const std::exception& e) {
        return wv::fail_with(WV_E_INVALID_ARGUMENT, e.what());
    }
    return WV_OK;
}

// ---- HRTF receiver (core::attenuator::hrtf) -----------------------------------------------------------
extern "C" int wv_hrtf_attenuation(const wv_hrtf_table* table, const float pointing[3], const float up[3], int32_t channel,
                                   const float incident[3], float bands[8]) {
    if (!table_ok(table) || !pointing || !up || !incident || !bands)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument (elevation divisions must be odd)");
    try {
        hrtf_attenuation(*table, make_orientation(pointing, up), channel, incident, bands);
    } catch (const std::exception& e) {
        return wv::fail_with(WV_E_INVALID_ARGUMENT, e.what());
    }
    return WV_OK;
}

extern "C" int wv_hrtf_ear_position(const float pointing[3], const float up[3], int32_t channel, float radius,
                                    const float base_position[3], float ear[3]) {
    if (!pointing || !up || !base_position || !ear) return wv::fail_with(WV_E_INVALID_ARGUMENT, "null argument");
    if (radius < 0 || 1 < radius) return wv::fail_with(WV_E_INVALID_ARGUMENT, "Hrtf radius outside reasonable range.");
    const Orientation o = make_orientation(pointing, up);
    const float x = channel ? radius : -radius;  // hrtf.cpp:135-141: left ear at -radius along the head's x axis
    for (int k = 0; k < 3; ++k) ear[k] = base_position[k] + o.m[k][0] * x;
    return WV_OK;
}

extern "C" int wv_attenuate_hrtf(const wv_hrtf_table* table, const float pointing[3], const float up[3], int32_t channel,
                                 float acoustic_impedance, const wv_directional_output* in, uint64_t n, float* out) {
    if (!table_ok(table) || !pointing || !up || (n && (!in || !out)))
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument (elevation divisions must be odd)");
    if (acoustic_impedance < 300 || 500 <= acoustic_impedance)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "Acoustic impedance outside expected range.");
    try {
        const Orientation o = make_orientation(pointing, up);
        for (uint64_t i = 0; i < n; ++i) attenuate_hrtf(*table, o, channel, acoustic_impedance, in[i], out + i * 8);
    } catch (const std::exception& e) {
        return wv::fail_with(WV_E_INVALID_ARGUMENT, e.what());
    }
    return WV_OK;
}

extern "C" int wv_multiband_filter_and_mixdown(float* bands, uint64_t n, double sample_rate, float* out) {
    if ((n && (!bands || !out)) || !(sample_rate > 0)) return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument");
    std::vector<float> work(bands, bands + n * 8);
    const std::vector<float> mixed = multiband_filter_and_mixdown(work, n, sample_rate);
    std::copy(work.begin(), work.end(), bands);
    std::copy(mixed.begin(), mixed.end(), out);
    return WV_OK;
}

extern "C" int wv_postprocess_waveguide_hrtf(const wv_waveguide_band* bands, uint32_t n_bands, const wv_hrtf_table* table,
                                             const float pointing[3], const float up[3], int32_t channel,
                                             float acoustic_impedance, double output_sample_rate, float* out,
                                             uint64_t capacity, uint64_t* n_out) {
    if ((n_bands && !bands) || !n_out || !table_ok(table) || !pointing || !up)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument (elevation divisions must be odd)");
    if (acoustic_impedance < 300 || 500 <= acoustic_impedance)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "Acoustic impedance outside expected range.");
    try {
        const Orientation o = make_orientation(pointing, up);
        std::vector<float> ret;
        for (uint32_t bi = 0; bi < n_bands; ++bi) {
            const wv_waveguide_band& band = bands[bi];
            // postprocess(band, method, ...) (postprocess.h:57-72) with the bands_type overload (:37-46)
            std::vector<float> per_band((size_t)band.n * 8);
            for (uint64_t i = 0; i < band.n; ++i)
                attenuate_hrtf(*table, o, channel, acoustic_impedance, band.directional[i], per_band.data() + i * 8);
            const std::vector<float> mixed = multiband_filter_and_mixdown(per_band, band.n, band.sample_rate);
            std::vector<float> processed = adjust_sampling_rate(mixed.data(), mixed.size(), band.sample_rate, output_sample_rate);
            const double lo = band.valid_hz_min / output_sample_rate, hi = band.valid_hz_max / output_sample_rate;
            frequency_domain_filter(processed.data(), processed.size(), [&](double f) {
                return lopass_magnitude(f, hi, 0.1, 0) * hipass_magnitude(f, lo, 0.1, 0);
            });
            ret.resize(std::max(ret.size(), processed.size()), 0.0f);
            for (size_t i = 0; i < processed.size(); ++i) ret[i] += processed[i];
        }
        const double dc_block = 10.0 / output_sample_rate;
        frequency_domain_filter(ret.data(), ret.size(), [&](double f) { return hipass_magnitude(f, dc_block, 0.9, 0); });
        *n_out = ret.size();
        if (out && capacity >= ret.size()) std::copy(ret.begin(), ret.end(), out);
    } catch (const std::exception& e) {
        return wv::fail_with(WV_E_INVALID_ARGUMENT, e.what());
    }
    return WV_OK;
}
