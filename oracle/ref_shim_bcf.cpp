// oracle/ref_shim_bcf.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Host-side harness for oracle/_ref/libwvref_bcf.so: the reference's OpenCL C *boundary
// coefficient* program (src/waveguide/src/boundary_coefficient_program.cpp:12-484 plus the sources
// it concatenates, :486-504) compiled for x86-64 by oracle/build_ref.py.  No reference code here:
// OpenCL builtins (ref_shim_builtins.h plus the few below) and a driver that enqueues the three
// kernels the way compute_boundary_index_data does
// (src/waveguide/src/boundary_coefficient_finder.cpp:73-124).
//
// Work items run one after the other in global-id order.  That matters for ONE output: the 1-D
// kernel's `popcount(boundary_type) == 1` test also admits id_inside (= 1) nodes, whose
// boundary_index is 0, so on a real device every inside node races to write entry 0 of the 1-D
// array.  Here the last writer in index order wins; see oracle/boundary_surfaces_oracle.c.
#include "ref_shim_builtins.h"

#include <cstdlib>
#include <cstring>

int ocl_popcount(int) asm("_Z8popcounti");
int ocl_popcount(int v) { return __builtin_popcount((unsigned)v); }

float3 ocl_max3(float3, float3) asm("_Z3maxDv3_fS_");
float3 ocl_max3(float3 a, float3 b) {  // OpenCL fmax-like max: y if x < y else x
    float3 r;
    r.x = a.x < b.x ? b.x : a.x;
    r.y = a.y < b.y ? b.y : a.y;
    r.z = a.z < b.z ? b.z : a.z;
    return r;
}
int3 ocl_maxi3(int3, int3) asm("_Z3maxDv3_iS_");
int3 ocl_maxi3(int3 a, int3 b) {
    int3 r;
    r.x = a.x < b.x ? b.x : a.x;
    r.y = a.y < b.y ? b.y : a.y;
    r.z = a.z < b.z ? b.z : a.z;
    return r;
}
int3 ocl_mini3(int3, int3) asm("_Z3minDv3_iS_");
int3 ocl_mini3(int3 a, int3 b) {
    int3 r;
    r.x = b.x < a.x ? b.x : a.x;
    r.y = b.y < a.y ? b.y : a.y;
    r.z = b.z < a.z ? b.z : a.z;
    return r;
}
float3 ocl_ceil3(float3) asm("_Z4ceilDv3_f");
float3 ocl_ceil3(float3 a) {
    float3 r;
    r.x = std::ceil(a.x);
    r.y = std::ceil(a.y);
    r.z = std::ceil(a.z);
    return r;
}
float ocl_distance(float3, float3) asm("_Z8distanceDv3_fS_");
float ocl_distance(float3 a, float3 b) {
    const float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
    return std::sqrt(x * x + y * y + z * z);
}

struct mesh_descriptor_cl {  // src/waveguide/include/waveguide/mesh_descriptor.h:58-66
    float3 min_corner;
    int3 dimensions;
    float spacing;
};
struct aabb_cl {  // src/core/include/core/cl/voxel_structs.h:13-21
    float3 c0;
    float3 c1;
};

extern "C" void boundary_coefficient_finder_1d(const void* nodes, mesh_descriptor_cl descriptor, void* boundary,
                                               const unsigned* voxel_index, aabb_cl global_aabb, unsigned side,
                                               const void* triangles, unsigned num_triangles,
                                               const float3* vertices);
extern "C" void boundary_coefficient_finder_2d(const void* nodes, mesh_descriptor_cl descriptor, void* boundary_2d,
                                               const void* boundary_1d);
extern "C" void boundary_coefficient_finder_3d(const void* nodes, mesh_descriptor_cl descriptor, void* boundary_3d,
                                               const void* boundary_1d);

extern "C" {

// The three kernels over all nodes.  `nodes` carry the first numbering of
// compute_boundary_index_data (1-D index counts re-entrant nodes too).  out1 [n1], out2 [n2][2],
// out3 [n3][3] are zero-filled first (the reference leaves slots it never writes uninitialised).
void wvref_boundary_coefficient_finder(const void* nodes, int nx, int ny, int nz, float spacing,
                                       const float* min_corner, const void* triangles, unsigned num_triangles,
                                       const void* vertices, unsigned* out1, size_t n1, unsigned* out2, size_t n2,
                                       unsigned* out3, size_t n3) {
    mesh_descriptor_cl d;
    d.min_corner.x = min_corner[0];
    d.min_corner.y = min_corner[1];
    d.min_corner.z = min_corner[2];
    d.dimensions.x = nx;
    d.dimensions.y = ny;
    d.dimensions.z = nz;
    d.spacing = spacing;
    aabb_cl box;  // the 1-D kernel takes the voxel arguments but only calls slow_closest_triangle
    box.c0 = (float3)(0.0f);
    box.c1 = (float3)(1.0f);
    std::memset(out1, 0, n1 * sizeof(unsigned));
    std::memset(out2, 0, n2 * 2 * sizeof(unsigned));
    std::memset(out3, 0, n3 * 3 * sizeof(unsigned));
    const size_t n = (size_t)nx * ny * nz;
    g_global_size = n;
    for (size_t i = 0; i < n; ++i) {
        g_global_id = i;
        boundary_coefficient_finder_1d(nodes, d, out1, nullptr, box, 1, triangles, num_triangles,
                                       (const float3*)vertices);
    }
    for (size_t i = 0; i < n; ++i) {
        g_global_id = i;
        boundary_coefficient_finder_2d(nodes, d, out2, out1);
    }
    for (size_t i = 0; i < n; ++i) {
        g_global_id = i;
        boundary_coefficient_finder_3d(nodes, d, out3, out1);
    }
}
}
