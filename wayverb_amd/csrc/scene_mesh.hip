// scene_mesh.hip -- triangle scene -> mesh without leaving the GPU: the four set-up stages of
// SURVEY.md 8(f) rank 1 chained on device-resident arrays, and an engine built straight from them.
//
// Replaces compute_mesh (src/waveguide/src/mesh.cpp:54-141) as one unit.  The reference reads the
// node array back after the two set-up kernels, numbers it on the host, uploads it again for the
// coefficient finder, reads the results back, and `run` uploads the nodes a third time
// (mesh.cpp:75-118, boundary_coefficient_finder.cpp:44-131, waveguide.h:52-58).  With 288 GB of HBM
// none of that has to move: inside flags, node types, the two boundary numberings, the per-class
// node lists and the surface arrays are produced and consumed in place, and `wv_create` takes the
// node array from device memory.  What crosses PCIe is the scene (KBs), per-block counters for the
// prefix sums (16 B per 1024 nodes) and, only if asked for, the result.
//
// The stage kernels are the ones behind wv_nodes_inside / wv_classify_nodes /
// wv_boundary_index_data (node_inside.hip, mesh_setup.hip, boundary_surfaces.hip); this file adds
// the numbering in between: boundary_index = running count per class in node order
// (set_boundary_index, boundary_coefficient_finder.cpp:11-19) as a two-level exclusive scan.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/wayverb_amd.h"

namespace wv {
int fail_with(int code, const std::string& msg);  // engine.hip
int validate_scene(const uint32_t* voxel_index, uint64_t n_voxel_words, uint32_t side, const uint32_t* triangles,
                   uint32_t n_triangles, uint32_t n_vertices);  // node_inside.hip
hipError_t nodes_inside_on_device(int nx, int ny, int nz, const float min_corner[3], float spacing,
                                  const uint32_t* d_voxel_index, const float aabb_min[3], const float aabb_max[3],
                                  uint32_t side, const uint32_t* d_triangles, const float* d_vertices, uint8_t* d_inside,
                                  hipStream_t stream);  // node_inside.hip
hipError_t node_types_on_device(const uint8_t* d_inside, int32_t* d_type, int nx, int ny, int nz,
                                hipStream_t stream);  // mesh_setup.hip
hipError_t boundary_surfaces_on_device(const wv_condensed_node* d_nodes, int nx, int ny, int nz,
                                       const float min_corner[3], float spacing, const uint64_t* const lists[3],
                                       const uint64_t lens[3], const uint32_t* d_slot_of_entry, const float* d_corners,
                                       const uint32_t* d_surface, uint32_t n_triangles, uint32_t* d_out1,
                                       uint32_t* d_out2, uint32_t* d_out3, hipStream_t stream);  // boundary_surfaces.hip
void pack_triangles(const uint32_t* triangles, uint32_t n_triangles, const float* vertices, std::vector<float>& corners,
                    std::vector<uint32_t>& surface);  // boundary_surfaces.hip
}  // namespace wv

namespace {

constexpr int kThreads = 256;
constexpr int kPerThread = 4;
constexpr int kChunk = kThreads * kPerThread;  // nodes per workgroup

// class of a node for the numberings: 0 = 1-D boundary or re-entrant (first numbering of the 1-D
// array, boundary_coefficient_finder.h:16-27), 1 = 2-D, 2 = 3-D, -1 = not numbered
__device__ __forceinline__ int numbering_class(int32_t bt) {
    if (bt == WV_ID_REENTRANT) return 0;
    if (bt == WV_ID_NONE || (bt & (WV_ID_INSIDE | WV_ID_REENTRANT))) return -1;
    const int bits = __popc((uint32_t)bt);
    return bits >= 1 && bits <= 3 ? bits - 1 : -1;
}

// per workgroup: how many nodes of each class (column 3: true 1-D nodes, the final numbering)
__global__ void __launch_bounds__(kThreads) count_kernel(const int32_t* type, uint64_t n, uint32_t* block_counts) {
    __shared__ uint32_t total[4];
    if (threadIdx.x < 4) total[threadIdx.x] = 0;
    __syncthreads();
    uint32_t c[4] = {0, 0, 0, 0};
    const uint64_t first = (uint64_t)blockIdx.x * kChunk + (uint64_t)threadIdx.x * kPerThread;
    for (int k = 0; k < kPerThread; ++k) {
        const uint64_t i = first + k;
        if (i >= n) break;
        const int32_t bt = type[i];
        const int d = numbering_class(bt);
        if (d >= 0) ++c[d];
        if (d == 0 && bt != WV_ID_REENTRANT) ++c[3];
    }
    for (int d = 0; d < 4; ++d)
        if (c[d]) atomicAdd(&total[d], c[d]);
    __syncthreads();
    if (threadIdx.x < 4) block_counts[(size_t)blockIdx.x * 4 + threadIdx.x] = total[threadIdx.x];
}

struct AssignArgs {
    const int32_t* type;
    uint64_t n;
    const uint32_t* block_start;  // [blocks][4] exclusive prefix of block_counts
    wv_condensed_node* nodes;     // out: type + first numbering
    uint64_t* list[3];            // out: node index per numbered node, per class
    uint32_t* final_rank;         // out [n class 0]: rank among the true 1-D nodes, ~0 for re-entrant
};

// in-order numbering inside a workgroup: exclusive scan of the per-thread counts through LDS
__global__ void __launch_bounds__(kThreads) assign_kernel(const AssignArgs a) {
    __shared__ uint32_t scan[4][kThreads];
    uint32_t c[4] = {0, 0, 0, 0};
    int32_t bt[kPerThread];
    const uint64_t first = (uint64_t)blockIdx.x * kChunk + (uint64_t)threadIdx.x * kPerThread;
    for (int k = 0; k < kPerThread; ++k) {
        const uint64_t i = first + k;
        bt[k] = i < a.n ? a.type[i] : WV_ID_NONE;
        const int d = numbering_class(bt[k]);
        if (d >= 0) ++c[d];
        if (d == 0 && bt[k] != WV_ID_REENTRANT) ++c[3];
    }
    for (int d = 0; d < 4; ++d) scan[d][threadIdx.x] = c[d];
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        uint32_t add[4];
        for (int d = 0; d < 4; ++d) add[d] = threadIdx.x >= (unsigned)off ? scan[d][threadIdx.x - off] : 0u;
        __syncthreads();
        for (int d = 0; d < 4; ++d) scan[d][threadIdx.x] += add[d];
        __syncthreads();
    }
    uint32_t next[4];
    for (int d = 0; d < 4; ++d) next[d] = a.block_start[(size_t)blockIdx.x * 4 + d] + scan[d][threadIdx.x] - c[d];
    for (int k = 0; k < kPerThread; ++k) {
        const uint64_t i = first + k;
        if (i >= a.n) break;
        const int d = numbering_class(bt[k]);
        wv_condensed_node rec;
        rec.boundary_type = bt[k];
        rec.boundary_index = 0;
        if (d >= 0) {
            rec.boundary_index = next[d];
            a.list[d][next[d]] = i;
            if (d == 0) a.final_rank[next[0]] = bt[k] != WV_ID_REENTRANT ? next[3]++ : ~0u;
            ++next[d];
        }
        a.nodes[i] = rec;
    }
}

// boundary_coefficient_finder.cpp:91-103,128: the true 1-D rows, and their nodes' final index
__global__ void __launch_bounds__(kThreads) finalize_kernel(const uint64_t* list0, const uint32_t* final_rank,
                                                            const uint32_t* out1_first, uint64_t n_entries,
                                                            wv_condensed_node* nodes, uint32_t* b1) {
    const uint64_t e = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= n_entries) return;
    const uint32_t r = final_rank[e];
    if (r == ~0u) return;  // re-entrant: keeps its first-numbering index, nothing reads it
    b1[r] = out1_first[e];
    nodes[list0[e]].boundary_index = r;
}

__global__ void __launch_bounds__(kThreads) iota_kernel(uint32_t* p, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

struct Dev {
    void* p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p);
    }
    void reset() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    ~Dev() { reset(); }
};

}  // namespace

struct wv_scene_mesh {
    int device = 0;
    int nx = 0, ny = 0, nz = 0;
    uint64_t counts[3] = {0, 0, 0};
    Dev nodes, b1, b2, b3;  // what outlives wv_scene_mesh_create
};

#define SM_HIP(expr)                                                                                     \
    do {                                                                                                   \
        hipError_t err__ = (expr);                                                                         \
        if (err__ != hipSuccess) return wv::fail_with(WV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(err__)); \
    } while (0)

extern "C" int wv_scene_mesh_create(int32_t nx, int32_t ny, int32_t nz, const float min_corner[3], float spacing,
                                    const uint32_t* voxel_index, uint64_t n_voxel_words, const float aabb_min[3],
                                    const float aabb_max[3], uint32_t side, const uint32_t* triangles,
                                    uint32_t n_triangles, const float* vertices, uint32_t n_vertices, int32_t device,
                                    wv_scene_mesh** out, uint64_t counts[3]) {
    if (nx < 1 || ny < 1 || nz < 1 || !min_corner || !voxel_index || !aabb_min || !aabb_max || !triangles || !vertices ||
        !out || side < 1 || n_triangles == 0 || n_voxel_words < (uint64_t)side * side * side)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument");
    if (int rc = wv::validate_scene(voxel_index, n_voxel_words, side, triangles, n_triangles, n_vertices)) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return wv::fail_with(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    if (device >= 0) SM_HIP(hipSetDevice(device));
    SM_HIP(hipGetDevice(&device));
    const uint64_t n = (uint64_t)nx * ny * nz;
    if (n >= 0xFFFFFFFEull) return wv::fail_with(WV_E_INVALID_ARGUMENT, "more than 2^32-2 nodes: decompose into z-slabs");

    std::unique_ptr<wv_scene_mesh> sm(new wv_scene_mesh);
    sm->device = device;
    sm->nx = nx;
    sm->ny = ny;
    sm->nz = nz;

    // scene -> device
    Dev d_vox, d_tri, d_vert, d_corners, d_surface;
    std::vector<float> corners;
    std::vector<uint32_t> surface;
    wv::pack_triangles(triangles, n_triangles, vertices, corners, surface);
    SM_HIP(d_vox.alloc(n_voxel_words * 4));
    SM_HIP(d_tri.alloc((size_t)n_triangles * 16));
    SM_HIP(d_vert.alloc((size_t)n_vertices * 16));
    SM_HIP(d_corners.alloc(corners.size() * 4));
    SM_HIP(d_surface.alloc(surface.size() * 4));
    SM_HIP(hipMemcpy(d_vox.p, voxel_index, n_voxel_words * 4, hipMemcpyHostToDevice));
    SM_HIP(hipMemcpy(d_tri.p, triangles, (size_t)n_triangles * 16, hipMemcpyHostToDevice));
    SM_HIP(hipMemcpy(d_vert.p, vertices, (size_t)n_vertices * 16, hipMemcpyHostToDevice));
    SM_HIP(hipMemcpy(d_corners.p, corners.data(), corners.size() * 4, hipMemcpyHostToDevice));
    SM_HIP(hipMemcpy(d_surface.p, surface.data(), surface.size() * 4, hipMemcpyHostToDevice));

    // stage 1 + 2: inside flags, node types
    Dev d_inside, d_type;
    SM_HIP(d_inside.alloc(n));
    SM_HIP(d_type.alloc(n * 4));
    SM_HIP(wv::nodes_inside_on_device(nx, ny, nz, min_corner, spacing, d_vox.as<uint32_t>(), aabb_min, aabb_max, side,
                                      d_tri.as<uint32_t>(), d_vert.as<float>(), d_inside.as<uint8_t>(), 0));
    SM_HIP(wv::node_types_on_device(d_inside.as<uint8_t>(), d_type.as<int32_t>(), nx, ny, nz, 0));
    SM_HIP(hipDeviceSynchronize());
    d_inside.reset();

    // numbering: per-workgroup counts -> host prefix (16 B per 1024 nodes) -> in-order assignment
    const uint64_t blocks = (n + kChunk - 1) / kChunk;
    Dev d_block;
    SM_HIP(d_block.alloc(blocks * 16));
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, 0, d_type.as<int32_t>(), n,
                       d_block.as<uint32_t>());
    SM_HIP(hipGetLastError());
    std::vector<uint32_t> block((size_t)blocks * 4);
    SM_HIP(hipMemcpy(block.data(), d_block.p, blocks * 16, hipMemcpyDeviceToHost));
    uint32_t run[4] = {0, 0, 0, 0};
    for (uint64_t b = 0; b < blocks; ++b)
        for (int d = 0; d < 4; ++d) {
            const uint32_t c = block[(size_t)b * 4 + d];
            block[(size_t)b * 4 + d] = run[d];
            run[d] += c;
        }
    const uint32_t c0 = run[0], c1 = run[1], c2 = run[2], true_1d = run[3];
    sm->counts[0] = true_1d;
    sm->counts[1] = c1;
    sm->counts[2] = c2;
    if (counts)
        for (int d = 0; d < 3; ++d) counts[d] = sm->counts[d];
    if (!c0 || !c1 || !c2)  // init_buffer, boundary_coefficient_finder.cpp:30-33
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "No boundaries.");
    SM_HIP(hipMemcpy(d_block.p, block.data(), blocks * 16, hipMemcpyHostToDevice));

    Dev d_list[3], d_rank, d_slot, d_first;
    SM_HIP(sm->nodes.alloc(n * sizeof(wv_condensed_node)));
    SM_HIP(d_list[0].alloc((size_t)c0 * 8));
    SM_HIP(d_list[1].alloc((size_t)c1 * 8));
    SM_HIP(d_list[2].alloc((size_t)c2 * 8));
    SM_HIP(d_rank.alloc((size_t)c0 * 4));
    AssignArgs aa{};
    aa.type = d_type.as<int32_t>();
    aa.n = n;
    aa.block_start = d_block.as<uint32_t>();
    aa.nodes = sm->nodes.as<wv_condensed_node>();
    for (int d = 0; d < 3; ++d) aa.list[d] = d_list[d].as<uint64_t>();
    aa.final_rank = d_rank.as<uint32_t>();
    hipLaunchKernelGGL(assign_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, 0, aa);
    SM_HIP(hipGetLastError());
    SM_HIP(hipDeviceSynchronize());
    d_type.reset();

    // stage 3: surfaces per filter on the first numbering, then the final 1-D rows
    SM_HIP(d_slot.alloc((size_t)c0 * 4));
    SM_HIP(d_first.alloc((size_t)c0 * 4));
    SM_HIP(sm->b1.alloc((size_t)true_1d * 4));
    SM_HIP(sm->b2.alloc((size_t)c1 * 8));
    SM_HIP(sm->b3.alloc((size_t)c2 * 12));
    SM_HIP(hipMemset(d_first.p, 0, (size_t)c0 * 4));
    SM_HIP(hipMemset(sm->b1.p, 0, (size_t)true_1d * 4));
    SM_HIP(hipMemset(sm->b2.p, 0, (size_t)c1 * 8));
    SM_HIP(hipMemset(sm->b3.p, 0, (size_t)c2 * 12));
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((c0 + kThreads - 1) / kThreads)), dim3(kThreads), 0, 0,
                       d_slot.as<uint32_t>(), (uint64_t)c0);
    const uint64_t* lists[3] = {d_list[0].as<uint64_t>(), d_list[1].as<uint64_t>(), d_list[2].as<uint64_t>()};
    const uint64_t lens[3] = {c0, c1, c2};
    SM_HIP(wv::boundary_surfaces_on_device(sm->nodes.as<wv_condensed_node>(), nx, ny, nz, min_corner, spacing, lists,
                                           lens, d_slot.as<uint32_t>(), d_corners.as<float>(), d_surface.as<uint32_t>(),
                                           n_triangles, d_first.as<uint32_t>(), sm->b2.as<uint32_t>(),
                                           sm->b3.as<uint32_t>(), 0));
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((c0 + kThreads - 1) / kThreads)), dim3(kThreads), 0, 0,
                       d_list[0].as<uint64_t>(), d_rank.as<uint32_t>(), d_first.as<uint32_t>(), (uint64_t)c0,
                       sm->nodes.as<wv_condensed_node>(), sm->b1.as<uint32_t>());
    SM_HIP(hipGetLastError());
    SM_HIP(hipDeviceSynchronize());
    *out = sm.release();
    return WV_OK;
}

extern "C" int wv_scene_mesh_fetch(const wv_scene_mesh* sm, wv_condensed_node* nodes, uint32_t* b1, uint32_t* b2,
                                   uint32_t* b3) {
    if (!sm) return wv::fail_with(WV_E_INVALID_ARGUMENT, "null scene mesh");
    SM_HIP(hipSetDevice(sm->device));
    const uint64_t n = (uint64_t)sm->nx * sm->ny * sm->nz;
    if (nodes) SM_HIP(hipMemcpy(nodes, sm->nodes.p, n * sizeof(wv_condensed_node), hipMemcpyDeviceToHost));
    if (b1) SM_HIP(hipMemcpy(b1, sm->b1.p, sm->counts[0] * 4, hipMemcpyDeviceToHost));
    if (b2) SM_HIP(hipMemcpy(b2, sm->b2.p, sm->counts[1] * 8, hipMemcpyDeviceToHost));
    if (b3) SM_HIP(hipMemcpy(b3, sm->b3.p, sm->counts[2] * 12, hipMemcpyDeviceToHost));
    return WV_OK;
}

extern "C" int wv_scene_mesh_create_engine(const wv_scene_mesh* sm, const wv_coefficients_canonical* coefficients,
                                           uint32_t num_coefficients, const wv_options* options, wv_engine** out) {
    if (!sm || !out) return wv::fail_with(WV_E_INVALID_ARGUMENT, "null argument");
    std::vector<uint32_t> b1(sm->counts[0]), b2(sm->counts[1] * 2), b3(sm->counts[2] * 3);
    if (int rc = wv_scene_mesh_fetch(sm, nullptr, b1.data(), b2.data(), b3.data())) return rc;
    wv_options opt;
    wv_default_options(&opt);
    if (options) {
        // as wv_create: a caller built against a shorter wv_options gives its prefix, the rest keeps the defaults
        const size_t n = std::min<size_t>(sizeof(opt), options->struct_size > 0 ? (size_t)options->struct_size : sizeof(opt));
        std::memcpy(&opt, options, n);
        opt.struct_size = (int32_t)sizeof(opt);
    }
    opt.device = sm->device;
    opt.nodes_on_device = 1;
    wv_mesh m{};
    m.nx = sm->nx;
    m.ny = sm->ny;
    m.nz = sm->nz;
    m.nodes = sm->nodes.as<wv_condensed_node>();
    m.coefficients = coefficients;
    m.num_coefficients = num_coefficients;
    m.boundary_indices_1 = b1.data();
    m.boundary_indices_2 = b2.data();
    m.boundary_indices_3 = b3.data();
    m.num_boundary_1 = sm->counts[0];
    m.num_boundary_2 = sm->counts[1];
    m.num_boundary_3 = sm->counts[2];
    return wv_create(&m, &opt, out);
}

extern "C" void wv_scene_mesh_destroy(wv_scene_mesh* sm) {
    if (!sm) return;
    (void)hipSetDevice(sm->device);
    delete sm;
}
