// oracle/ref_shim_builtins.h -- TEST INFRASTRUCTURE ONLY.  The OpenCL 1.2 language builtins (by mangled
// name) that the host-compiled reference set-up programs call; shared by ref_shim_setup.cpp and
// ref_shim_bcf.cpp.  No reference code.  dot / cross / length are the plain left-to-right float
// expressions (see the precision note in ref_shim_setup.cpp).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

typedef int int3 __attribute__((ext_vector_type(3)));
typedef float float3 __attribute__((ext_vector_type(3)));

static thread_local size_t g_global_id = 0;
static thread_local size_t g_global_size = 0;

size_t ocl_get_global_id(unsigned) asm("_Z13get_global_idj");
size_t ocl_get_global_id(unsigned) { return g_global_id; }
size_t ocl_get_global_size(unsigned) asm("_Z15get_global_sizej");
size_t ocl_get_global_size(unsigned) { return g_global_size; }

int ocl_atomic_xchg(volatile int*, int) asm("_Z11atomic_xchgPU8CLglobalVii");
int ocl_atomic_xchg(volatile int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }

int3 ocl_convert_int3(float3) asm("_Z12convert_int3Dv3_f");
int3 ocl_convert_int3(float3 v) {  // default conversion: round toward zero
    int3 r;
    r.x = (int)v.x;
    r.y = (int)v.y;
    r.z = (int)v.z;
    return r;
}
float3 ocl_convert_float3(int3) asm("_Z14convert_float3Dv3_i");
float3 ocl_convert_float3(int3 v) {
    float3 r;
    r.x = (float)v.x;
    r.y = (float)v.y;
    r.z = (float)v.z;
    return r;
}
// any / all: most significant bit of any / every component (OpenCL 1.2, 6.12.6)
int ocl_all(int3) asm("_Z3allDv3_i");
int ocl_all(int3 v) { return (v.x < 0) && (v.y < 0) && (v.z < 0); }
int ocl_any(int3) asm("_Z3anyDv3_i");
int ocl_any(int3 v) { return (v.x < 0) || (v.y < 0) || (v.z < 0); }

float ocl_dot(float3, float3) asm("_Z3dotDv3_fS_");
float ocl_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float3 ocl_cross(float3, float3) asm("_Z5crossDv3_fS_");
float3 ocl_cross(float3 a, float3 b) {
    float3 r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}
float ocl_length(float3) asm("_Z6lengthDv3_f");
float ocl_length(float3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
float3 ocl_normalize(float3) asm("_Z9normalizeDv3_f");
float3 ocl_normalize(float3 a) {
    const float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    float3 r;
    r.x = a.x / l;
    r.y = a.y / l;
    r.z = a.z / l;
    return r;
}
float3 ocl_fabs3(float3) asm("_Z4fabsDv3_f");
float3 ocl_fabs3(float3 a) {
    float3 r;
    r.x = std::fabs(a.x);
    r.y = std::fabs(a.y);
    r.z = std::fabs(a.z);
    return r;
}
float ocl_fabsf(float) asm("_Z4fabsf");
float ocl_fabsf(float a) { return std::fabs(a); }
float3 ocl_floor3(float3) asm("_Z5floorDv3_f");
float3 ocl_floor3(float3 a) {
    float3 r;
    r.x = std::floor(a.x);
    r.y = std::floor(a.y);
    r.z = std::floor(a.z);
    return r;
}
// vector relational builtins return -1 (all bits set) per true component
int3 ocl_isnan3(float3) asm("_Z5isnanDv3_f");
int3 ocl_isnan3(float3 a) {
    int3 r;
    r.x = std::isnan(a.x) ? -1 : 0;
    r.y = std::isnan(a.y) ? -1 : 0;
    r.z = std::isnan(a.z) ? -1 : 0;
    return r;
}
int3 ocl_signbit3(float3) asm("_Z7signbitDv3_f");
int3 ocl_signbit3(float3 a) {
    int3 r;
    r.x = std::signbit(a.x) ? -1 : 0;
    r.y = std::signbit(a.y) ? -1 : 0;
    r.z = std::signbit(a.z) ? -1 : 0;
    return r;
}
// select(a, b, c): per component, b if the MSB of c is set, else a
float3 ocl_select_f(float3, float3, int3) asm("_Z6selectDv3_fS_Dv3_i");
float3 ocl_select_f(float3 a, float3 b, int3 c) {
    float3 r;
    r.x = c.x < 0 ? b.x : a.x;
    r.y = c.y < 0 ? b.y : a.y;
    r.z = c.z < 0 ? b.z : a.z;
    return r;
}
int3 ocl_select_i(int3, int3, int3) asm("_Z6selectDv3_iS_S_");
int3 ocl_select_i(int3 a, int3 b, int3 c) {
    int3 r;
    r.x = c.x < 0 ? b.x : a.x;
    r.y = c.y < 0 ? b.y : a.y;
    r.z = c.z < 0 ? b.z : a.z;
    return r;
}

