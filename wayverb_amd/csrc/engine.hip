// engine.hip -- the C ABI of include/wayverb_amd.h over `wv::Engine<Real>` (engine.hip.h and the engine_*.hip.h files it
// names), plus the two unit kernels that ship with it (device triad, IIR unit test).
//
// Replaces the body of `waveguide::run` (src/waveguide/include/waveguide/waveguide.h:36-126): device buffers, the
// step loop, the error-flag protocol, and (device-resident) the single-node source / node-gather receivers every
// caller of `run` uses.
//
// There is no CPU path in this library: without a HIP device every entry point fails.
#include "engine.hip.h"
#include "engine_setup.hip.h"
#include "engine_single.hip.h"
#include "engine_pair.hip.h"
#include "engine_triple.hip.h"
#include "engine_batch.hip.h"
#include "engine_io.hip.h"
#include "engine_slab.hip.h"

namespace {
thread_local std::string g_last_error;
thread_local hipError_t g_last_hip_error = hipSuccess;
}  // namespace

namespace wv {
int fail_with(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
void note_hip_error(hipError_t err) { g_last_hip_error = err; }
hipError_t last_hip_error() { return g_last_hip_error; }
}  // namespace wv

using wv::DeviceGuard;
using wv::Engine;
using wv::fail;
using wv::ScopedDevice;

// =============================================================================================
extern "C" {

const char* wv_last_error(void) { return g_last_error.c_str(); }

void wv_default_options(wv_options* o) {
    std::memset(o, 0, sizeof(*o));
    o->struct_size = (int32_t)sizeof(wv_options);
    o->precision = WV_PRECISION_F64;
    o->device = -1;
    o->flag_interval = 0;
    o->stream_variant = 2;
    wv_tuning& t = o->tuning;
    t.pair = -1;
    t.pair_chunks = 0;
    t.pair_inner_fix = 1;
    t.pair_wide = 1;
    t.pair_unit_waves = 1;
    t.pair_unit_planes = 32;
    t.pair_units_by_chunk = 1;
    t.tile_lists = 1;
    t.fuse_pre_post = 1;
    t.graph = 0;
    t.boundary_lds = 1;
    t.boundary_order = 1;
    t.boundary_xwall = 1;
    t.slab_early = -1;
    t.pair_split_rows = 0;
    t.fuse_planes = 1;
    t.whole_step = -1;
    t.triple = -1;
    t.triple_chunks = 0;
    t.triple_lanes = 0;
}

#ifdef WV_DEBUG_ENV
// Measurement builds only (tools/): WV_<FIELD> in the environment overrides a tuning field when an engine is
// created.  The product library is built without this and reads no environment variables.
static void tuning_from_environment(wv_options* o) {
    struct Knob {
        const char* name;
        int32_t* field;
    };
    wv_tuning& t = o->tuning;
    const Knob knobs[] = {{"WV_PAIR", &t.pair}, {"WV_PAIR_CHUNKS", &t.pair_chunks}, {"WV_PAIR_INNER_FIX", &t.pair_inner_fix},
                          {"WV_PAIR_WIDE", &t.pair_wide}, {"WV_PAIR_UNIT_WAVES", &t.pair_unit_waves},
                          {"WV_PAIR_UNIT_PLANES", &t.pair_unit_planes}, {"WV_PAIR_UNITS_BY_CHUNK", &t.pair_units_by_chunk}, {"WV_TILE_LISTS", &t.tile_lists},
                          {"WV_FUSE_PRE_POST", &t.fuse_pre_post}, {"WV_GRAPH", &t.graph}, {"WV_BOUNDARY_LDS", &t.boundary_lds},
                          {"WV_BOUNDARY_ORDER", &t.boundary_order}, {"WV_BOUNDARY_XWALL", &t.boundary_xwall},
                          {"WV_SLAB_EARLY", &t.slab_early}, {"WV_PAIR_SPLIT_ROWS", &t.pair_split_rows}, {"WV_FUSE_PLANES", &t.fuse_planes}, {"WV_WHOLE_STEP", &t.whole_step}, {"WV_TRIPLE", &t.triple}, {"WV_TRIPLE_CHUNKS", &t.triple_chunks}, {"WV_TRIPLE_LANES", &t.triple_lanes},
                          {"WV_STREAM_VARIANT", &o->stream_variant},
                          {"WV_STREAM_RY", &t.stream_ry}, {"WV_STREAM_NWX", &t.stream_nwx}, {"WV_STREAM_NWY", &t.stream_nwy},
                          {"WV_STREAM_ZCHUNKS", &t.stream_zchunks}};
    for (const Knob& k : knobs)
        if (const char* v = std::getenv(k.name)) *k.field = std::atoi(v);
}
#endif

int wv_create(const wv_mesh* mesh, const wv_options* options, wv_engine** out) {
    if (!mesh || !out) return fail(WV_E_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    wv_options opt;
    wv_default_options(&opt);
    if (options) {
        const size_t n = std::min<size_t>(sizeof(opt), options->struct_size > 0 ? (size_t)options->struct_size : sizeof(opt));
        std::memcpy(&opt, options, n);
        opt.struct_size = (int32_t)sizeof(opt);
    }
#ifdef WV_DEBUG_ENV
    tuning_from_environment(&opt);
#endif
    std::unique_ptr<wv_engine> e;
    if (opt.precision == WV_PRECISION_F32) {
        e.reset(new Engine<float>());
    } else if (opt.precision == WV_PRECISION_F64) {
        e.reset(new Engine<double>());
    } else {
        return fail(WV_E_INVALID_ARGUMENT, "unknown precision");
    }
    const int rc = e->init(*mesh, opt);
    if (rc != WV_OK) return rc;
    *out = e.release();
    return WV_OK;
}

void wv_destroy(wv_engine* e) { delete e; }

#define WV_NEED(e) \
    if (!(e)) return fail(WV_E_INVALID_ARGUMENT, "null engine")

int wv_read_value(wv_engine* e, int buffer, uint64_t index, double* value) {
    WV_NEED(e);
    return e->read_value(buffer, index, value);
}
int wv_write_value(wv_engine* e, int buffer, uint64_t index, double value) {
    WV_NEED(e);
    return e->write_value(buffer, index, value);
}
int wv_read_field(wv_engine* e, int buffer, void* dst, int elem_size) {
    WV_NEED(e);
    return e->read_field(buffer, dst, elem_size);
}
int wv_write_field(wv_engine* e, int buffer, const void* src, int elem_size) {
    WV_NEED(e);
    return e->write_field(buffer, src, elem_size);
}
int wv_read_planes(wv_engine* e, int buffer, int32_t z_begin, int32_t z_count, void* dst, int elem_size) {
    WV_NEED(e);
    return e->read_planes(buffer, z_begin, z_count, dst, elem_size);
}
int wv_write_planes(wv_engine* e, int buffer, int32_t z_begin, int32_t z_count, const void* src, int elem_size) {
    WV_NEED(e);
    return e->write_planes(buffer, z_begin, z_count, src, elem_size);
}
int wv_read_boundary_data(wv_engine* e, int dim, wv_boundary_data* dst) {
    WV_NEED(e);
    return e->boundary_data(dim, dst, false);
}
int wv_write_boundary_data(wv_engine* e, int dim, const wv_boundary_data* src) {
    WV_NEED(e);
    return e->boundary_data(dim, const_cast<wv_boundary_data*>(src), true);
}
int wv_set_coefficients(wv_engine* e, const wv_coefficients_canonical* c, uint32_t n) {
    WV_NEED(e);
    return e->set_coefficients(c, n);
}
int wv_device_buffer(wv_engine* e, int buffer, void** p) {
    WV_NEED(e);
    return e->device_buffer(buffer, p);
}
int wv_host_register(void* p, uint64_t bytes) {
    if (!p || !bytes) return fail(WV_E_INVALID_ARGUMENT, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    WV_HIP(hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault));
    return WV_OK;
}
int wv_host_unregister(void* p) {
    if (!p) return fail(WV_E_INVALID_ARGUMENT, "null argument");
    WV_HIP(hipHostUnregister(p));
    return WV_OK;
}
int wv_checkpoint(wv_engine* e) {
    WV_NEED(e);
    return e->checkpoint(0);
}
int wv_rollback(wv_engine* e) {
    WV_NEED(e);
    return e->checkpoint(1);
}
int wv_drop_checkpoint(wv_engine* e) {
    WV_NEED(e);
    return e->checkpoint(2);
}
int wv_step(wv_engine* e, int32_t* flag) {
    WV_NEED(e);
    return e->step(flag);
}
int wv_swap(wv_engine* e) {
    WV_NEED(e);
    return e->swap();
}
int wv_set_source(wv_engine* e, int kind, uint64_t node, const double* signal, uint64_t n) {
    WV_NEED(e);
    return e->set_source(kind, node, signal, n);
}
int wv_set_receivers(wv_engine* e, const uint64_t* nodes, uint32_t n) {
    WV_NEED(e);
    return e->set_receivers(nodes, n);
}
int wv_run(wv_engine* e, uint64_t n_steps, uint64_t* done, int32_t* flag) {
    WV_NEED(e);
    return e->run(n_steps, done, flag);
}
int wv_fetch_receivers(wv_engine* e, uint64_t first, uint64_t n, double* dst) {
    WV_NEED(e);
    return e->fetch_receivers(first, n, dst);
}
int wv_step_count(wv_engine* e, uint64_t* steps) {
    WV_NEED(e);
    *steps = e->steps_done;
    return WV_OK;
}
int wv_kernel_time_ms(wv_engine* e, double* mean_ms, uint64_t* launches) {
    WV_NEED(e);
    return e->kernel_time(mean_ms, launches, nullptr);
}
int wv_kernel_time_detail(wv_engine* e, double* mean_ms, uint64_t* launches, uint64_t* steps) {
    WV_NEED(e);
    return e->kernel_time(mean_ms, launches, steps);
}
int wv_enable_kernel_timing(wv_engine* e, int enable) {
    WV_NEED(e);
    e->timing = enable != 0;
    return WV_OK;
}
int wv_query(wv_engine* e, int what, uint64_t* value) {
    WV_NEED(e);
    return e->query(what, value);
}
int wv_synchronize(wv_engine* e) {
    WV_NEED(e);
    return e->synchronize();
}
int wv_set_stream_tuning(wv_engine* e, int variant, int ry, int nwx, int nwy, int zchunks) {
    WV_NEED(e);
    return e->set_tuning(variant, ry, nwx, nwy, zchunks);
}
int wv_comm_use_library(const char* path) {
    std::string err;
    if (!wv::SlabComm::use_library(path, &err)) return fail(WV_E_STATE, err);
    return WV_OK;
}
int wv_comm_unique_id(void* id_bytes) {
    std::string err;
    if (!wv::SlabComm::unique_id(id_bytes, &err)) return fail(WV_E_COMM, err);
    return WV_OK;
}
int wv_comm_init(wv_engine* e, const void* id_bytes, int rank, int nranks) {
    WV_NEED(e);
    return e->comm_init(id_bytes, rank, nranks);
}
int wv_comm_destroy(wv_engine* e) {
    WV_NEED(e);
    return e->comm_destroy();
}

int wv_comm_init_local(wv_engine* const* engines, int32_t n) { return wv::group_init_local(engines, n); }

int wv_run_group(wv_engine* const* engines, int32_t n, uint64_t n_steps, uint64_t* steps_done, int32_t* flag_out) {
    return wv::group_run(engines, n, n_steps, steps_done, flag_out);
}

int wv_field_pitch(wv_engine* e, uint64_t* pitch_elements) {
    WV_NEED(e);
    *pitch_elements = e->field_pitch();
    return WV_OK;
}

// a[i] = b[i] + s * c[i] over `n` doubles, `iters` timed launches after one warm-up: the classic device
// triad, as the yardstick bench.py prints next to the stencil's own bandwidth.  Written
// the way this chip streams best: short-lived workgroups in address order, 16 B per
// lane, 16 KiB per workgroup and array.
__global__ void __launch_bounds__(256) triad_kernel(double* a, const double* b, const double* c, double s, int64_t n2) {
    typedef double V2 __attribute__((ext_vector_type(2)));
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    V2 x[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + j * 256;
        x[j] = i < n2 ? __builtin_nontemporal_load(reinterpret_cast<const V2*>(b) + i) : (V2)(0.0);
        y[j] = i < n2 ? __builtin_nontemporal_load(reinterpret_cast<const V2*>(c) + i) : (V2)(0.0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + j * 256;
        if (i < n2) __builtin_nontemporal_store(x[j] + s * y[j], reinterpret_cast<V2*>(a) + i);
    }
}

int wv_measure_triad(int32_t device, uint64_t n_doubles, int32_t iters, double* gb_per_s) {
    if (!gb_per_s || n_doubles < 2 || iters < 1) return fail(WV_E_INVALID_ARGUMENT, "bad triad arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    DeviceGuard guard(device);
    ScopedDevice a, b, c;
    const size_t bytes = (size_t)n_doubles * sizeof(double);
    WV_HIP(hipMalloc(&a.p, bytes));
    WV_HIP(hipMalloc(&b.p, bytes));
    WV_HIP(hipMalloc(&c.p, bytes));
    WV_HIP(hipMemset(b.p, 0, bytes));
    WV_HIP(hipMemset(c.p, 0, bytes));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    WV_HIP(hipEventCreate(&e0));
    WV_HIP(hipEventCreate(&e1));
    const int64_t n2 = (int64_t)(n_doubles / 2);
    const unsigned grid = (unsigned)((n2 + 1023) / 1024);
    for (int it = 0; it < iters + 1; ++it) {
        if (it == 1) WV_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(triad_kernel, dim3(grid), dim3(256), 0, 0, static_cast<double*>(a.p), static_cast<const double*>(b.p),
                           static_cast<const double*>(c.p), 0.5, n2);
    }
    WV_HIP(hipEventRecord(e1, 0));
    WV_HIP(hipEventSynchronize(e1));
    float ms = 0;
    WV_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    WV_HIP(hipGetLastError());
    *gb_per_s = 3.0 * (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
    return WV_OK;
}

int wv_filter_test_2(const float* input, float* output, double* memory, const wv_coefficients_canonical* coeffs,
                     uint32_t n_filters, uint32_t n_samples) {
    if (!input || !output || !memory || !coeffs) return fail(WV_E_INVALID_ARGUMENT, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    const size_t n = n_filters, total = (size_t)n_filters * n_samples;
    ScopedDevice m_in, m_out, m_mem, m_c;
    WV_HIP(hipMalloc(&m_in.p, std::max<size_t>(total, 1) * sizeof(float)));
    WV_HIP(hipMalloc(&m_out.p, std::max<size_t>(total, 1) * sizeof(float)));
    WV_HIP(hipMalloc(&m_mem.p, std::max<size_t>(n, 1) * 6 * sizeof(double)));
    WV_HIP(hipMalloc(&m_c.p, std::max<size_t>(n, 1) * 14 * sizeof(double)));
    float *d_in = static_cast<float*>(m_in.p), *d_out = static_cast<float*>(m_out.p);
    double *d_mem = static_cast<double*>(m_mem.p), *d_c = static_cast<double*>(m_c.p);
    if (total == 0) return WV_OK;
    WV_HIP(hipMemcpy(d_in, input, total * sizeof(float), hipMemcpyHostToDevice));
    WV_HIP(hipMemcpy(d_mem, memory, n * 6 * sizeof(double), hipMemcpyHostToDevice));
    WV_HIP(hipMemcpy(d_c, coeffs, n * 14 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(wv::filter_test_2_kernel, dim3((n_filters + 63) / 64), dim3(64), 0, 0, d_in, d_out, d_mem, d_c,
                       n_filters, n_samples);
    WV_HIP(hipGetLastError());
    WV_HIP(hipMemcpy(output, d_out, total * sizeof(float), hipMemcpyDeviceToHost));
    WV_HIP(hipMemcpy(memory, d_mem, n * 6 * sizeof(double), hipMemcpyDeviceToHost));
    return WV_OK;
}

}  // extern "C"

