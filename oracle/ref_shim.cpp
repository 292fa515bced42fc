// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Host-side harness for oracle/_ref: the reference's OpenCL C waveguide program
// (src/waveguide/src/program.cpp:11-532 and the helper sources it concatenates) compiled
// for x86-64 by oracle/build_ref.py.  This file contains NO reference code.  It provides
//   (1) the OpenCL 1.2 *language builtins* the kernel text calls, by their Itanium-mangled
//       OpenCL names, with the semantics the OpenCL 1.2 specification gives them;
//   (2) serial / z-chunk-threaded driver loops that play the role of
//       clEnqueueNDRangeKernel(NDRange(num_nodes)) (src/waveguide/include/waveguide/waveguide.h:85-97).
//
// Built twice: -DWVREF_REAL=float -DWVREF_TAG=f32 and -DWVREF_REAL=double -DWVREF_TAG=f64.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <thread>
#include <vector>

typedef int int3 __attribute__((ext_vector_type(3)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef WVREF_REAL real;

#define WV_CAT2(a, b) a##b
#define WV_CAT(a, b) WV_CAT2(a, b)
#define WV_NAME(x) WV_CAT(WV_CAT(x, _), WVREF_TAG)

static thread_local size_t g_global_id = 0;

// ---- OpenCL builtins (mangled as clang -x cl emits the calls) --------------------------
size_t ocl_get_global_id(unsigned) asm("_Z13get_global_idj");
size_t ocl_get_global_id(unsigned) { return g_global_id; }

float3 ocl_convert_float3(int3) asm("_Z14convert_float3Dv3_i");
float3 ocl_convert_float3(int3 v) {
    float3 r;
    r.x = (float)v.x;
    r.y = (float)v.y;
    r.z = (float)v.z;
    return r;
}

// any(): true if the most significant bit of any component is set (OpenCL 1.2, 6.12.6)
int ocl_any(int3) asm("_Z3anyDv3_i");
int ocl_any(int3 v) { return (v.x < 0) || (v.y < 0) || (v.z < 0); }

float ocl_sqrtf(float) asm("_Z4sqrtf");
float ocl_sqrtf(float x) { return std::sqrt(x); }
int ocl_isinff(float) asm("_Z5isinff");
int ocl_isinff(float x) { return std::isinf(x) ? 1 : 0; }
int ocl_isnanf(float) asm("_Z5isnanf");
int ocl_isnanf(float x) { return std::isnan(x) ? 1 : 0; }

double ocl_sqrtd(double) asm("_Z4sqrtd");
double ocl_sqrtd(double x) { return std::sqrt(x); }
int ocl_isinfd(double) asm("_Z5isinfd");
int ocl_isinfd(double x) { return std::isinf(x) ? 1 : 0; }
int ocl_isnand(double) asm("_Z5isnand");
int ocl_isnand(double x) { return std::isnan(x) ? 1 : 0; }

int ocl_popcount(int) asm("_Z8popcounti");
int ocl_popcount(int x) { return __builtin_popcount((unsigned)x); }

int ocl_atomic_or(volatile int*, int) asm("_Z9atomic_orPU8CLglobalVii");
int ocl_atomic_or(volatile int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

// ---- kernel entry points exported by the compiled program object ------------------------
extern "C" void condensed_waveguide(real* previous, const real* current, const void* nodes,
                                    int3 dimensions, void* b1, void* b2, void* b3,
                                    const void* coefficients, volatile int* error_flag);
extern "C" void filter_test(const float* input, float* output, void* biquad_memory,
                            const void* biquad_coefficients);
extern "C" void filter_test_2(const float* input, float* output, void* canonical_memory,
                              const void* canonical_coefficients);

// ---- drivers ------------------------------------------------------------------------------
extern "C" {

// One kernel launch over all nodes: previous <- next (in place), as waveguide.h:85-97.
void WV_NAME(wvref_step)(real* previous, const real* current, const void* nodes, int nx, int ny,
                         int nz, void* b1, void* b2, void* b3, const void* coefficients,
                         int* error_flag, int threads) {
    int3 dims;
    dims.x = nx;
    dims.y = ny;
    dims.z = nz;
    const size_t n = (size_t)nx * ny * nz;
    auto body = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            g_global_id = i;
            condensed_waveguide(previous, current, nodes, dims, b1, b2, b3, coefficients,
                                error_flag);
        }
    };
    if (threads <= 1) {
        body(0, n);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        size_t lo = n * t / threads, hi = n * (t + 1) / threads;
        pool.emplace_back(body, lo, hi);
    }
    for (auto& t : pool) t.join();
}

// tests/rectangular_kernel.cpp:180 launches these over NDRange(256): one work-item per filter.
void WV_NAME(wvref_filter_test_2)(const float* input, float* output, void* memory,
                                  const void* coefficients, int n) {
    for (int i = 0; i < n; ++i) {
        g_global_id = (size_t)i;
        filter_test_2(input, output, memory, coefficients);
    }
}

void WV_NAME(wvref_filter_test)(const float* input, float* output, void* memory,
                                const void* coefficients, int n) {
    for (int i = 0; i < n; ++i) {
        g_global_id = (size_t)i;
        filter_test(input, output, memory, coefficients);
    }
}

int WV_NAME(wvref_sizeof_real)(void) { return (int)sizeof(real); }
}
