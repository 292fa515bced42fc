// oracle/ref_shim_setup.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Host-side harness for oracle/_ref/libwvref_setup.so: the reference's OpenCL C *mesh set-up*
// program (src/waveguide/src/mesh_setup_program.cpp:13-172 plus the core geometry / voxel sources
// it concatenates, :175-193) compiled for x86-64 by oracle/build_ref.py.  Like ref_shim.cpp this
// file contains NO reference code: only the OpenCL 1.2 language builtins the kernel text calls
// and driver loops standing in for clEnqueueNDRangeKernel (src/waveguide/src/mesh.cpp:75-111).
//
// Precision note: OpenCL leaves the evaluation order of dot / cross / length / normalize to the
// implementation; the definitions below are the plain left-to-right float expressions, compiled
// without contraction.  Parity of the inside test is therefore "against this evaluation order".
#include "ref_shim_builtins.h"

// ---- the program's device structs as the kernel entry points receive them -----------------------
struct mesh_descriptor_cl {  // src/waveguide/include/waveguide/mesh_descriptor.h:58-66
    float3 min_corner;
    int3 dimensions;
    float spacing;
};
struct aabb_cl {  // src/core/include/core/cl/voxel_structs.h:13-21
    float3 c0;
    float3 c1;
};

extern "C" void set_node_boundary_type(void* nodes, mesh_descriptor_cl descriptor);
extern "C" void set_node_inside(void* nodes, mesh_descriptor_cl descriptor, const unsigned* voxel_index,
                                aabb_cl global_aabb, unsigned side, const void* triangles, const float3* vertices);

extern "C" {

// set_node_boundary_type over all nodes (src/waveguide/src/mesh.cpp:105-108).  Serial on purpose:
// the kernel reads neighbours' types while writing its own (Q5 in SURVEY.md App. D) -- only
// `== id_inside` is ever compared, and inside nodes never change, so any order gives one answer.
void wvref_set_node_boundary_type(void* nodes, int nx, int ny, int nz, float spacing) {
    mesh_descriptor_cl d;
    d.min_corner = (float3)(0.0f);
    d.dimensions.x = nx;
    d.dimensions.y = ny;
    d.dimensions.z = nz;
    d.spacing = spacing;
    const size_t n = (size_t)nx * ny * nz;
    g_global_size = n;
    for (size_t i = 0; i < n; ++i) {
        g_global_id = i;
        set_node_boundary_type(nodes, d);
    }
}

// set_node_inside over all nodes (mesh.cpp:78-90).  vertices: float4-strided (cl_float3).
void wvref_set_node_inside(void* nodes, int nx, int ny, int nz, float spacing, const float* min_corner,
                           const unsigned* voxel_index, const float* aabb_c0, const float* aabb_c1, unsigned side,
                           const void* triangles, const void* vertices) {
    mesh_descriptor_cl d;
    d.min_corner.x = min_corner[0];
    d.min_corner.y = min_corner[1];
    d.min_corner.z = min_corner[2];
    d.dimensions.x = nx;
    d.dimensions.y = ny;
    d.dimensions.z = nz;
    d.spacing = spacing;
    aabb_cl box;
    box.c0.x = aabb_c0[0];
    box.c0.y = aabb_c0[1];
    box.c0.z = aabb_c0[2];
    box.c1.x = aabb_c1[0];
    box.c1.y = aabb_c1[1];
    box.c1.z = aabb_c1[2];
    const size_t n = (size_t)nx * ny * nz;
    g_global_size = n;
    for (size_t i = 0; i < n; ++i) {
        g_global_id = i;
        set_node_inside(nodes, d, voxel_index, box, side, triangles, (const float3*)vertices);
    }
}
}
