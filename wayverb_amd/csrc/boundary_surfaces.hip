// boundary_surfaces.hip -- which scene surface each boundary filter of the mesh takes:
// SURVEY.md 8(f) rank 1, third slice.
//
// Replaces compute_boundary_index_data (src/waveguide/src/boundary_coefficient_finder.cpp:38-131)
// and its three kernels (src/waveguide/src/boundary_coefficient_program.cpp):
//   boundary_coefficient_finder_1d :310-338  nearest triangle (brute force over the whole list,
//                                            slow_closest_triangle :218-235, exact point-triangle
//                                            distance :16-143) -> that triangle's surface
//   boundary_coefficient_finder_2d :356-413  per port, the first "1-D" node among the 6 face
//                                            neighbours donates its surface
//   boundary_coefficient_finder_3d :429-484  same over the 12 edge neighbours
// followed by the host pass that drops re-entrant slots from the 1-D array and renumbers the true
// 1-D nodes (boundary_coefficient_finder.cpp:91-103,128).
//
// Layout for the device: the host compacts the nodes that need a nearest-triangle search into a
// list (they are a surface's worth of the volume, so a thread-per-node launch would leave most
// lanes idle), and packs the triangles as 9 floats + surface so a workgroup can stage them through
// LDS and every lane reads the same triangle at the same time (LDS broadcast, no bank conflicts).
// Distances are single precision in the reference's expression order, no contraction, so ties
// between triangles (strict <, lowest index wins) fall the same way as in the reference.
//
// One deliberate difference: the reference's 1-D kernel tests
// popcount(boundary_type) == 1, which id_inside nodes (type 1, index 0) also pass, so on a real
// device every inside node races with the rightful owner for entry 0 of the 1-D array.  Here only
// the owner writes it.  Everything downstream of that test is kept as written -- including that
// the 2-D / 3-D kernels accept inside and re-entrant neighbours as donors.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wayverb_amd.h"

namespace wv {
int fail_with(int code, const std::string& msg);  // engine.hip
}

namespace {

struct f3 {
    float x, y, z;
};
__device__ inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// parameter along one triangle edge: 0 at the near vertex, 1 at the far one
__device__ inline float along_edge(float b, float a) {
    if (0 <= b) return 0.0f;
    if (a <= -b) return 1.0f;
    return -b / a;
}

// boundary_coefficient_program.cpp:16-143: squared distance from p to the triangle, by the region
// of the (t0, t1) parameter plane the unconstrained minimum falls in.
__device__ inline float point_triangle_dist2(f3 v0, f3 v1, f3 v2, f3 p) {
    const f3 diff = p - v0, e0 = v1 - v0, e1 = v2 - v0;
    const float a00 = dot3(e0, e0), a01 = dot3(e0, e1), a11 = dot3(e1, e1);
    const float b0 = -dot3(diff, e0), b1 = -dot3(diff, e1);
    const float det = a00 * a11 - a01 * a01;
    float t0 = a01 * b1 - a11 * b0;
    float t1 = a01 * b0 - a00 * b1;
    if (t0 + t1 <= det) {
        if (t0 < 0) {
            if (t1 < 0 && b0 < 0) {
                t1 = 0;
                t0 = a00 <= -b0 ? 1.0f : -b0 / a00;
            } else {
                t0 = 0;
                t1 = along_edge(b1, a11);
            }
        } else if (t1 < 0) {
            t1 = 0;
            t0 = along_edge(b0, a00);
        } else {
            const float inv = 1 / det;
            t0 *= inv;
            t1 *= inv;
        }
    } else {
        const float denom = a00 - 2 * a01 + a11;  // the hypotenuse's squared length
        if (t0 < 0) {
            const float m0 = a01 + b0, m1 = a11 + b1;
            if (m0 < m1) {
                const float numer = m1 - m0;
                t0 = denom <= numer ? 1.0f : numer / denom;
                t1 = denom <= numer ? 0.0f : 1 - t0;
            } else {
                t0 = 0;
                t1 = m1 <= 0 ? 1.0f : along_edge(b1, a11);
            }
        } else if (t1 < 0) {
            const float m0 = a01 + b1, m1 = a00 + b0;
            if (m0 < m1) {
                const float numer = m1 - m0;
                t1 = denom <= numer ? 1.0f : numer / denom;
                t0 = denom <= numer ? 0.0f : 1 - t1;
            } else {
                t1 = 0;
                t0 = m1 <= 0 ? 1.0f : along_edge(b0, a00);
            }
        } else {
            const float numer = a11 + b1 - a01 - b0;
            if (numer <= 0) {
                t0 = 0;
                t1 = 1;
            } else {
                t0 = denom <= numer ? 1.0f : numer / denom;
                t1 = denom <= numer ? 0.0f : 1 - t0;
            }
        }
    }
    const f3 closest = {v0.x + e0.x * t0 + e1.x * t1, v0.y + e0.y * t0 + e1.y * t1, v0.z + e0.z * t0 + e1.z * t1};
    const f3 d = p - closest;
    return dot3(d, d);
}

constexpr int kBlock = 256;
constexpr int kChunk = 512;  // triangles per LDS stage: 512 * 9 floats = 18 KiB

struct NearestArgs {
    const uint64_t* node_of_entry;  // [n_entries] linear node index
    const uint32_t* slot_of_entry;  // [n_entries] row of out1 to write
    const float* corners;           // [n_triangles][9]
    const uint32_t* surface;        // [n_triangles]
    uint32_t* out1;
    uint64_t n_entries;
    uint32_t n_triangles;
    int nx, ny;
    f3 min_corner;
    float spacing;
};

__global__ __launch_bounds__(kBlock) void nearest_surface_kernel(NearestArgs a) {
    __shared__ float tri[kChunk * 9];
    const uint64_t entry = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = entry < a.n_entries;
    f3 p = {0, 0, 0};
    if (live) {
        const uint64_t i = a.node_of_entry[entry];
        const int x = (int)(i % (uint64_t)a.nx);
        const uint64_t q = i / (uint64_t)a.nx;
        const int y = (int)(q % (uint64_t)a.ny), z = (int)(q / (uint64_t)a.ny);
        // compute_node_position (src/waveguide/src/cl/utils.cpp): min_corner + locator * spacing
        p = {a.min_corner.x + (float)x * a.spacing, a.min_corner.y + (float)y * a.spacing,
             a.min_corner.z + (float)z * a.spacing};
    }
    uint32_t best = 0;
    float best_d = INFINITY;
    for (uint32_t base = 0; base < a.n_triangles; base += kChunk) {
        const uint32_t count = min((uint32_t)kChunk, a.n_triangles - base);
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < count * 9; w += kBlock) tri[w] = a.corners[(size_t)base * 9 + w];
        __syncthreads();
        if (live) {
            for (uint32_t k = 0; k < count; ++k) {
                const float* c = tri + 9 * k;
                const float d = point_triangle_dist2({c[0], c[1], c[2]}, {c[3], c[4], c[5]}, {c[6], c[7], c[8]}, p);
                if (d < best_d) {
                    best = base + k;
                    best_d = d;
                }
            }
        }
    }
    if (live) a.out1[a.slot_of_entry[entry]] = a.surface[best];
}

struct GatherArgs {
    const wv_condensed_node* nodes;
    const uint64_t* node_of_entry;  // the D-dimensional boundary nodes
    const uint32_t* out1;
    uint32_t* out;  // [n_D][D]
    uint64_t n_entries;
    int nx, ny, nz;
};

__constant__ int8_t k_face_offsets[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
__constant__ int8_t k_edge_offsets[12][3] = {{-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 0, -1}, {-1, 0, 1},
                                             {1, 0, -1},  {1, 0, 1},  {0, -1, -1}, {0, -1, 1}, {0, 1, -1}, {0, 1, 1}};

template <int D>
__global__ __launch_bounds__(kBlock) void gather_surfaces_kernel(GatherArgs a) {
    const uint64_t entry = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (entry >= a.n_entries) return;
    const uint64_t i = a.node_of_entry[entry];
    const wv_condensed_node me = a.nodes[i];
    const int x = (int)(i % (uint64_t)a.nx);
    const uint64_t q = i / (uint64_t)a.nx;
    const int y = (int)(q % (uint64_t)a.ny), z = (int)(q / (uint64_t)a.ny);
    uint32_t* row = a.out + (size_t)me.boundary_index * D;
    constexpr int n_offsets = D == 2 ? 6 : 12;
    // The donor search does not depend on the port, so find it once and hand it to every port.
    bool found = false;
    uint32_t surface = 0;
    for (int j = 0; j < n_offsets && !found; ++j) {
        const int8_t* o = D == 2 ? k_face_offsets[j] : k_edge_offsets[j];
        const int ax = x + o[0], ay = y + o[1], az = z + o[2];
        if (ax < 0 || ay < 0 || az < 0 || a.nx <= ax || a.ny <= ay || a.nz <= az) continue;
        const wv_condensed_node other = a.nodes[((size_t)az * a.ny + ay) * a.nx + ax];
        if (__popc((uint32_t)other.boundary_type) != 1) continue;
        surface = a.out1[other.boundary_index];
        found = true;
    }
    if (found)
        for (int k = 0; k < D; ++k) row[k] = surface;  // popcount(type) == D ports, in port order
}

template <typename T>
struct DeviceArray {
    T* p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
    hipError_t upload(const T* src, size_t n) {
        hipError_t rc = alloc(n);
        if (rc == hipSuccess && n) rc = hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice);
        return rc;
    }
    ~DeviceArray() {
        if (p) (void)hipFree(p);
    }
};

}  // namespace

namespace wv {

// The three finder kernels on arrays that already live on the device (stream-ordered).  lists[d] /
// lens[d]: node indices of the 1-D-or-re-entrant, 2-D and 3-D nodes in node order; slot_of_entry:
// row of out1 each entry of lists[0] writes; out1/out2/out3 zero-filled by the caller.
// Shared with scene_mesh.hip.
hipError_t boundary_surfaces_on_device(const wv_condensed_node* d_nodes, int nx, int ny, int nz,
                                       const float min_corner[3], float spacing, const uint64_t* const lists[3],
                                       const uint64_t lens[3], const uint32_t* d_slot_of_entry, const float* d_corners,
                                       const uint32_t* d_surface, uint32_t n_triangles, uint32_t* d_out1,
                                       uint32_t* d_out2, uint32_t* d_out3, hipStream_t stream) {
    NearestArgs na{};
    na.node_of_entry = lists[0];
    na.slot_of_entry = d_slot_of_entry;
    na.corners = d_corners;
    na.surface = d_surface;
    na.out1 = d_out1;
    na.n_entries = lens[0];
    na.n_triangles = n_triangles;
    na.nx = nx;
    na.ny = ny;
    na.min_corner = {min_corner[0], min_corner[1], min_corner[2]};
    na.spacing = spacing;
    if (na.n_entries)
        hipLaunchKernelGGL(nearest_surface_kernel, dim3((unsigned)((na.n_entries + kBlock - 1) / kBlock)), dim3(kBlock),
                           0, stream, na);
    GatherArgs ga{};
    ga.nodes = d_nodes;
    ga.out1 = d_out1;
    ga.nx = nx;
    ga.ny = ny;
    ga.nz = nz;
    ga.node_of_entry = lists[1];
    ga.out = d_out2;
    ga.n_entries = lens[1];
    if (ga.n_entries)
        hipLaunchKernelGGL(gather_surfaces_kernel<2>, dim3((unsigned)((ga.n_entries + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, stream, ga);
    ga.node_of_entry = lists[2];
    ga.out = d_out3;
    ga.n_entries = lens[2];
    if (ga.n_entries)
        hipLaunchKernelGGL(gather_surfaces_kernel<3>, dim3((unsigned)((ga.n_entries + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, stream, ga);
    return hipGetLastError();
}

// triangles {surface, v0, v1, v2} + cl_float3 vertices -> 9 corner floats + surface per triangle
void pack_triangles(const uint32_t* triangles, uint32_t n_triangles, const float* vertices, std::vector<float>& corners,
                    std::vector<uint32_t>& surface) {
    corners.resize((size_t)n_triangles * 9);
    surface.resize(n_triangles);
    for (uint32_t t = 0; t < n_triangles; ++t) {
        surface[t] = triangles[4 * (size_t)t];
        for (int k = 0; k < 3; ++k)
            for (int e = 0; e < 3; ++e)
                corners[(size_t)t * 9 + k * 3 + e] = vertices[4 * (size_t)triangles[4 * (size_t)t + 1 + k] + e];
    }
}

}  // namespace wv

extern "C" int wv_boundary_index_data(int32_t nx, int32_t ny, int32_t nz, const float min_corner[3], float spacing,
                                      wv_condensed_node* nodes, const uint32_t* triangles, uint32_t n_triangles,
                                      const float* vertices, uint32_t n_vertices, uint32_t* b1, uint64_t capacity_1,
                                      uint32_t* b2, uint64_t capacity_2, uint32_t* b3, uint64_t capacity_3,
                                      uint64_t counts[3]) {
    if (nx < 1 || ny < 1 || nz < 1 || !min_corner || !nodes || !triangles || !vertices || !counts || n_triangles == 0)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument");
    for (uint32_t t = 0; t < n_triangles; ++t)
        for (int k = 1; k < 4; ++k)
            if (triangles[4 * (size_t)t + k] >= n_vertices)
                return wv::fail_with(WV_E_INVALID_ARGUMENT, "triangle refers to a missing vertex");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return wv::fail_with(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");

    // first numbering (boundary_coefficient_finder.cpp:44-54): the 1-D array also has a slot for
    // every re-entrant node; lists of the nodes each kernel has work for
    const size_t n = (size_t)nx * ny * nz;
    // (two passes over the planes on all host cores: count per plane, prefix, fill)
    auto dim_of = [](int32_t bt) -> int {
        if (bt == WV_ID_REENTRANT) return 0;
        if (bt == WV_ID_NONE || (bt & (WV_ID_INSIDE | WV_ID_REENTRANT))) return -1;
        const int bits = __builtin_popcount((uint32_t)bt);
        return bits >= 1 && bits <= 3 ? bits - 1 : -1;
    };
    const size_t plane = (size_t)nx * ny;
    const int n_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), (unsigned)nz);
    std::vector<uint32_t> start((size_t)(nz + 1) * 4, 0);  // per plane: 1-D+re-entrant, 2-D, 3-D, true 1-D
    auto for_planes = [&](auto&& body) {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t)
            pool.emplace_back([&, t] {
                for (int z = t; z < nz; z += n_threads) body(z);
            });
        for (auto& th : pool) th.join();
    };
    for_planes([&](int z) {
        uint32_t k[4] = {0, 0, 0, 0};
        for (size_t i = (size_t)z * plane; i < (size_t)(z + 1) * plane; ++i) {
            const int d = dim_of(nodes[i].boundary_type);
            if (d >= 0) ++k[d];
            if (d == 0 && nodes[i].boundary_type != WV_ID_REENTRANT) ++k[3];
        }
        for (int d = 0; d < 4; ++d) start[(size_t)(z + 1) * 4 + d] = k[d];
    });
    for (int z = 0; z < nz; ++z)
        for (int d = 0; d < 4; ++d) start[(size_t)(z + 1) * 4 + d] += start[(size_t)z * 4 + d];
    const uint32_t* total = &start[(size_t)nz * 4];
    const uint32_t c[3] = {total[0], total[1], total[2]};
    const uint64_t true_1d = total[3];
    std::vector<uint64_t> list[3];
    for (int d = 0; d < 3; ++d) list[d].resize(c[d]);
    std::vector<uint32_t> slot1(c[0]);
    for_planes([&](int z) {
        uint32_t k[3] = {start[(size_t)z * 4], start[(size_t)z * 4 + 1], start[(size_t)z * 4 + 2]};
        for (size_t i = (size_t)z * plane; i < (size_t)(z + 1) * plane; ++i) {
            const int d = dim_of(nodes[i].boundary_type);
            if (d < 0) {
                nodes[i].boundary_index = 0;
                continue;
            }
            nodes[i].boundary_index = k[d];
            list[d][k[d]] = i;
            if (d == 0) slot1[k[d]] = k[d];
            ++k[d];
        }
    });
    counts[0] = true_1d;
    counts[1] = c[1];
    counts[2] = c[2];
    if (!c[0] || !c[1] || !c[2])  // init_buffer, boundary_coefficient_finder.cpp:30-33
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "No boundaries.");
    if (!b1 && !b2 && !b3) return WV_OK;  // size query: counts only (the nodes' indices hold the first numbering)
    if (capacity_1 < true_1d || capacity_2 < c[1] || capacity_3 < c[2] || !b1 || !b2 || !b3)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "boundary index arrays too small (see counts)");

    std::vector<float> corners;
    std::vector<uint32_t> surface;
    wv::pack_triangles(triangles, n_triangles, vertices, corners, surface);

    DeviceArray<wv_condensed_node> d_nodes;
    DeviceArray<uint64_t> d_list[3];
    DeviceArray<uint32_t> d_slot1, d_surface, d_out1, d_out2, d_out3;
    DeviceArray<float> d_corners;
    hipError_t rc = d_nodes.upload(nodes, n);
    for (int d = 0; d < 3 && rc == hipSuccess; ++d) rc = d_list[d].upload(list[d].data(), list[d].size());
    if (rc == hipSuccess) rc = d_slot1.upload(slot1.data(), slot1.size());
    if (rc == hipSuccess) rc = d_surface.upload(surface.data(), surface.size());
    if (rc == hipSuccess) rc = d_corners.upload(corners.data(), corners.size());
    if (rc == hipSuccess) rc = d_out1.alloc(c[0]);
    if (rc == hipSuccess) rc = d_out2.alloc((size_t)c[1] * 2);
    if (rc == hipSuccess) rc = d_out3.alloc((size_t)c[2] * 3);
    // slots no node writes read as 0 (the reference leaves them uninitialised)
    if (rc == hipSuccess) rc = hipMemset(d_out1.p, 0, (size_t)c[0] * sizeof(uint32_t));
    if (rc == hipSuccess) rc = hipMemset(d_out2.p, 0, (size_t)c[1] * 2 * sizeof(uint32_t));
    if (rc == hipSuccess) rc = hipMemset(d_out3.p, 0, (size_t)c[2] * 3 * sizeof(uint32_t));
    if (rc != hipSuccess) return wv::fail_with(WV_E_HIP, hipGetErrorString(rc));

    const uint64_t* lists[3] = {d_list[0].p, d_list[1].p, d_list[2].p};
    const uint64_t lens[3] = {list[0].size(), list[1].size(), list[2].size()};
    rc = wv::boundary_surfaces_on_device(d_nodes.p, nx, ny, nz, min_corner, spacing, lists, lens, d_slot1.p, d_corners.p,
                                         d_surface.p, n_triangles, d_out1.p, d_out2.p, d_out3.p, 0);
    std::vector<uint32_t> first(c[0]);
    if (rc == hipSuccess) rc = hipMemcpy(first.data(), d_out1.p, (size_t)c[0] * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (rc == hipSuccess) rc = hipMemcpy(b2, d_out2.p, (size_t)c[1] * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (rc == hipSuccess) rc = hipMemcpy(b3, d_out3.p, (size_t)c[2] * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (rc != hipSuccess) return wv::fail_with(WV_E_HIP, hipGetErrorString(rc));

    // boundary_coefficient_finder.cpp:91-103,128: keep the true 1-D rows, renumber their nodes;
    // re-entrant nodes keep the index of the first numbering (nothing reads it)
    uint32_t kept = 0;
    for (uint64_t i : list[0]) {
        if (nodes[i].boundary_type == WV_ID_REENTRANT) continue;
        b1[kept] = first[nodes[i].boundary_index];
        nodes[i].boundary_index = kept++;
    }
    return WV_OK;
}
