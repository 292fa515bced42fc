// node_inside.hip -- "is this mesh node inside the room?" for triangle-soup scenes:
// SURVEY.md 8(f) rank 1, second slice.
//
// Replaces the reference's `set_node_inside` kernel (src/waveguide/src/mesh_setup_program.cpp:110-140)
// and what it calls: `voxel_inside` / `single_ray_inside` / `count_intersections` and the 3-D DDA
// of VOXEL_TRAVERSAL_ALGORITHM (src/core/src/cl/voxel.cpp:16-66,98-225), Moeller-Trumbore
// `triangle_vert_intersection`, `is_degenerate`, `almost_equal` (src/core/src/cl/geometry.cpp:7-64);
// plus a host voxeliser producing the flattened voxel -> triangle-list array the kernel walks
// (format of src/core/src/spatial_division/voxel_collection.cpp:9-37).
//
// A node is inside when a ray from it crosses the surface an odd number of times; a crossing
// within 10 ulp of a triangle edge or vertex makes that ray "unsure" and the next of 32 fixed
// directions is tried; all unsure -> outside.  All float arithmetic is single precision in the
// reference's expression order (dot = (ax*bx + ay*by) + az*bz, no contraction), so the flags are
// reproducible bit for bit against the reference kernel compiled for the host.
//
// One node per lane; the work is a handful of voxel steps and triangle tests per node and runs once
// per scene, so this is not a roofline kernel -- it only has to be exact and not silly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
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
__host__ __device__ inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline f3 cross3(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// geometry.cpp:7-11
__device__ inline bool almost_equal(float x, float y, float ulp) {
    const float abs_diff = fabsf(x - y);
    return abs_diff < FLT_EPSILON * fabsf(x + y) * ulp || abs_diff < FLT_MIN;
}

struct Inter {
    float t, u, v;
};

// geometry.cpp:20-56 (Moeller-Trumbore; t == 0 encodes "no hit")
__device__ inline Inter triangle_hit(f3 v0, f3 v1, f3 v2, f3 pos, f3 dir) {
    const Inter none = {0.0f, 0.0f, 0.0f};
    const f3 e0 = v1 - v0;
    const f3 e1 = v2 - v0;
    const f3 pvec = cross3(dir, e1);
    const float det = dot3(e0, pvec);
    if (almost_equal(det, 0.0f, 10.0f)) return none;
    const float invdet = 1.0f / det;
    const f3 tvec = pos - v0;
    const float u = invdet * dot3(tvec, pvec);
    if (u < 0.0f || 1.0f < u) return none;
    const f3 qvec = cross3(tvec, e0);
    const float v = invdet * dot3(dir, qvec);
    if (v < 0.0f || 1.0f < v + u) return none;
    const float t = invdet * dot3(e1, qvec);
    if (t < 0 || almost_equal(t, 0.0f, 10.0f)) return none;
    return {t, u, v};
}

struct InsideArgs {
    uint8_t* inside;
    int nx, ny, nz;
    f3 min_corner;
    float spacing;
    const uint32_t* voxel_index;
    f3 c0, c1;  // voxelised bounding box
    uint32_t side;
    const uint32_t* triangles;  // {surface, v0, v1, v2}
    const float* vertices;      // 4 floats per vertex (cl_float3)
};

// the 32 fixed ray directions of src/core/src/cl/voxel.cpp:156-189 (data, needed verbatim for
// identical results)
__constant__ float kDirections[32][3] = {
        {-0.427602, 0.791267, -0.437096},  {-0.832527, -0.545442, 0.0969113}, {0.633363, 0.413131, 0.65435},
        {0.985873, 0.140209, 0.0916325},   {0.384519, 0.0309011, -0.9226},    {-0.532584, -0.0244727, 0.846023},
        {0.844848, 0.230031, -0.483029},   {-0.186143, -0.291698, -0.938223}, {-0.108511, -0.861706, 0.495669},
        {0.0951741, 0.959367, -0.265625},  {0.407194, 0.907127, -0.106369},   {0.521731, -0.00522727, -0.853094},
        {0.369627, 0.218276, 0.903179},    {-0.518837, 0.815586, -0.25618},   {-0.954901, 0.105507, 0.277548},
        {0.63419, 0.768703, 0.0830607},    {-0.0258027, 0.998294, 0.052379},  {-0.868361, 0.473347, 0.147958},
        {0.346294, -0.131168, 0.928911},   {-0.635896, 0.649019, 0.417624},   {0.293121, 0.235495, -0.926619},
        {-0.55088, -0.0237137, -0.834247}, {-0.661022, -0.653122, -0.369434}, {0.224176, -0.351092, 0.909109},
        {0.456587, 0.736627, -0.498907},   {0.965231, 0.154753, 0.210667},    {0.626034, -0.245898, 0.740011},
        {0.435825, 0.794758, -0.422393},   {0.662049, 0.713267, 0.23009},     {0.261843, -0.620862, 0.738897},
        {0.23673, 0.714889, 0.657946},     {-0.404007, 0.699316, 0.589691},
};

// count_intersections (voxel.cpp:98-125) with the traversal macro (:22-66) inlined.
// Returns the crossing count, or ~0u when a crossing is degenerate.
__device__ uint32_t count_crossings(const InsideArgs& a, f3 pos, f3 dir) {
    const float side_f = (float)a.side;
    const float vd[3] = {(a.c1.x - a.c0.x) / side_f, (a.c1.y - a.c0.y) / side_f, (a.c1.z - a.c0.z) / side_f};
    const float p[3] = {pos.x, pos.y, pos.z}, d[3] = {dir.x, dir.y, dir.z};
    const float c0[3] = {a.c0.x, a.c0.y, a.c0.z};
    int ind[3];
    for (int i = 0; i < 3; ++i) ind[i] = (int)floorf((p[i] - c0[i]) / vd[i]);
    uint32_t count = 0;
    const int side = (int)a.side;
    if (ind[0] < 0 || ind[1] < 0 || ind[2] < 0 || ind[0] >= side || ind[1] >= side || ind[2] >= side) return 0;

    int step[3], just_out[3];
    float t_max[3], t_delta[3];
    for (int i = 0; i < 3; ++i) {
        const float lo = c0[i] + (float)(ind[i] + 0) * vd[i];
        const float hi = c0[i] + (float)(ind[i] + 1) * vd[i];
        const bool neg = signbit(d[i]);
        step[i] = neg ? -1 : 1;
        just_out[i] = neg ? -1 : side;
        const float boundary = neg ? lo : hi;
        const float tm = fabsf((boundary - p[i]) / d[i]);
        t_max[i] = isnan(tm) ? INFINITY : tm;
        t_delta[i] = fabsf(vd[i] / d[i]);
    }
    float prev_max = 0;
    for (;;) {
        int min_i = 0;
        for (int i = 1; i != 3; ++i)
            if (t_max[i] < t_max[min_i]) min_i = i;
        const uint32_t voxel_offset = a.voxel_index[(size_t)ind[0] * a.side * a.side + (size_t)ind[1] * a.side + ind[2]];
        const uint32_t num = a.voxel_index[voxel_offset];
        const uint32_t* list = a.voxel_index + voxel_offset + 1;
        const float max_dist = t_max[min_i];
        for (uint32_t i = 0; i != num; ++i) {
            const uint32_t* tri = a.triangles + 4 * (size_t)list[i];
            const float* q0 = a.vertices + 4 * (size_t)tri[1];
            const float* q1 = a.vertices + 4 * (size_t)tri[2];
            const float* q2 = a.vertices + 4 * (size_t)tri[3];
            const Inter in = triangle_hit({q0[0], q0[1], q0[2]}, {q1[0], q1[1], q1[2]}, {q2[0], q2[1], q2[2]}, pos, dir);
            if (in.t) {
                // is_degenerate (geometry.cpp:15-18)
                if (almost_equal(in.u, 0.0f, 10.0f) || almost_equal(in.v, 0.0f, 10.0f) || almost_equal(in.u + in.v, 1.0f, 10.0f))
                    return ~0u;
                if (prev_max < in.t && in.t <= max_dist) count += 1;
            }
        }
        ind[min_i] += step[min_i];
        if (ind[min_i] == just_out[min_i]) break;
        prev_max = t_max[min_i];
        t_max[min_i] += t_delta[min_i];
    }
    return count;
}

__global__ void __launch_bounds__(256) node_inside_kernel(const InsideArgs a) {
    const int64_t n = (int64_t)a.nx * a.ny * a.nz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.nx);
        const int64_t q = i / a.nx;
        const int y = (int)(q % a.ny), z = (int)((q / a.ny) % a.nz);
        // compute_node_position (src/waveguide/src/cl/utils.cpp:71-74)
        const f3 pos = {a.min_corner.x + (float)x * a.spacing, a.min_corner.y + (float)y * a.spacing,
                        a.min_corner.z + (float)z * a.spacing};
        uint8_t result = 0;  // voxel_inside (voxel.cpp:197-225): all rays unsure -> outside
        for (int k = 0; k < 32; ++k) {
            const uint32_t c = count_crossings(a, pos, {kDirections[k][0], kDirections[k][1], kDirections[k][2]});
            if (c != ~0u) {
                result = (uint8_t)(c % 2);
                break;
            }
        }
        a.inside[i] = result;
    }
}

// ---- host voxeliser ------------------------------------------------------------------------------
// Exact triangle / axis-aligned-box overlap by separating axes (3 box normals, the triangle
// normal, 9 edge cross products), in double.
bool tri_box_overlap(const double c[3], const double h[3], const double tv[3][3]) {
    double v[3][3];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) v[i][k] = tv[i][k] - c[k];
    for (int k = 0; k < 3; ++k) {
        const double lo = std::min({v[0][k], v[1][k], v[2][k]}), hi = std::max({v[0][k], v[1][k], v[2][k]});
        if (lo > h[k] || hi < -h[k]) return false;
    }
    const double e[3][3] = {{v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]},
                            {v[2][0] - v[1][0], v[2][1] - v[1][1], v[2][2] - v[1][2]},
                            {v[0][0] - v[2][0], v[0][1] - v[2][1], v[0][2] - v[2][2]}};
    auto separated = [&](const double ax[3]) {
        const double p0 = ax[0] * v[0][0] + ax[1] * v[0][1] + ax[2] * v[0][2];
        const double p1 = ax[0] * v[1][0] + ax[1] * v[1][1] + ax[2] * v[1][2];
        const double p2 = ax[0] * v[2][0] + ax[1] * v[2][1] + ax[2] * v[2][2];
        const double r = h[0] * std::fabs(ax[0]) + h[1] * std::fabs(ax[1]) + h[2] * std::fabs(ax[2]);
        return std::min({p0, p1, p2}) > r || std::max({p0, p1, p2}) < -r;
    };
    const double nrm[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2],
                           e[0][0] * e[1][1] - e[0][1] * e[1][0]};
    if (separated(nrm)) return false;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            double ax[3] = {0, 0, 0};  // unit_k x e_i
            ax[(k + 1) % 3] = -e[i][(k + 2) % 3];
            ax[(k + 2) % 3] = e[i][(k + 1) % 3];
            if (separated(ax)) return false;
        }
    return true;
}

}  // namespace

// Flattened voxel -> triangle lists, the array `get_flattened` builds
// (src/core/src/spatial_division/voxel_collection.cpp:9-37): words [0, side^3) = offset of voxel
// (x, y, z) at index x*side^2 + y*side + z; at each offset {count, tri_0, tri_1, ...}.  A triangle
// belongs to a voxel when it overlaps the voxel box padded by 0.001
// (src/core/include/core/spatial_division/voxelised_scene_data.h:28-44).
// Two-call protocol: out == nullptr or capacity too small -> *needed receives the word count.
extern "C" int wv_voxelise(const float* vertices, uint32_t n_vertices, const uint32_t* triangles, uint32_t n_triangles,
                           const float aabb_min[3], const float aabb_max[3], uint32_t side, uint32_t* out,
                           uint64_t capacity, uint64_t* needed) {
    if (!vertices || !triangles || !aabb_min || !aabb_max || side < 1 || !needed)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument");
    for (uint32_t t = 0; t < n_triangles; ++t)
        for (int k = 1; k < 4; ++k)
            if (triangles[4 * (size_t)t + k] >= n_vertices)
                return wv::fail_with(WV_E_INVALID_ARGUMENT, "triangle refers to a missing vertex");
    const size_t cells = (size_t)side * side * side;
    std::vector<std::vector<uint32_t>> lists(cells);
    double dim[3];
    for (int k = 0; k < 3; ++k) dim[k] = ((double)aabb_max[k] - (double)aabb_min[k]) / side;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n_threads = std::min<unsigned>(hw, side);
    auto work = [&](unsigned tid) {
        for (uint32_t x = tid; x < side; x += n_threads) {
            for (uint32_t t = 0; t < n_triangles; ++t) {
                double tv[3][3], lo[3], hi[3];
                for (int i = 0; i < 3; ++i)
                    for (int k = 0; k < 3; ++k) tv[i][k] = vertices[4 * (size_t)triangles[4 * (size_t)t + 1 + i] + k];
                for (int k = 0; k < 3; ++k) {
                    lo[k] = std::min({tv[0][k], tv[1][k], tv[2][k]});
                    hi[k] = std::max({tv[0][k], tv[1][k], tv[2][k]});
                }
                // candidate cells from the triangle's bounding box, then the exact test
                int r0[3], r1[3];
                for (int k = 0; k < 3; ++k) {
                    r0[k] = std::max(0, (int)std::floor((lo[k] - 0.001 - aabb_min[k]) / dim[k]));
                    r1[k] = std::min((int)side - 1, (int)std::floor((hi[k] + 0.001 - aabb_min[k]) / dim[k]));
                }
                if ((int)x < r0[0] || (int)x > r1[0]) continue;
                for (int y = r0[1]; y <= r1[1]; ++y)
                    for (int z = r0[2]; z <= r1[2]; ++z) {
                        const int idx[3] = {(int)x, y, z};
                        double c[3], h[3];
                        for (int k = 0; k < 3; ++k) {
                            c[k] = aabb_min[k] + (idx[k] + 0.5) * dim[k];
                            h[k] = 0.5 * dim[k] + 0.001;
                        }
                        if (tri_box_overlap(c, h, tv)) lists[((size_t)x * side + y) * side + z].push_back(t);
                    }
            }
        }
    };
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    uint64_t words = cells;
    for (const auto& l : lists) words += 1 + l.size();
    *needed = words;
    if (!out || capacity < words) return WV_OK;
    uint64_t cursor = cells;
    for (size_t v = 0; v < cells; ++v) {
        out[v] = (uint32_t)cursor;
        out[cursor++] = (uint32_t)lists[v].size();
        for (uint32_t t : lists[v]) out[cursor++] = t;
    }
    return WV_OK;
}

namespace wv {

// shared with scene_mesh.hip: the checks a scene must pass before a kernel follows its indices
int validate_scene(const uint32_t* voxel_index, uint64_t n_voxel_words, uint32_t side, const uint32_t* triangles,
                   uint32_t n_triangles, uint32_t n_vertices) {
    for (uint32_t t = 0; t < n_triangles; ++t)
        for (int k = 1; k < 4; ++k)
            if (triangles[4 * (size_t)t + k] >= n_vertices)
                return fail_with(WV_E_INVALID_ARGUMENT, "triangle refers to a missing vertex");
    if (!voxel_index) return WV_OK;
    if (n_voxel_words < (uint64_t)side * side * side) return fail_with(WV_E_INVALID_ARGUMENT, "voxel array too short");
    for (uint64_t cell = 0, cells = (uint64_t)side * side * side; cell < cells; ++cell) {
        const uint64_t off = voxel_index[cell];
        if (off >= n_voxel_words || off + 1 + voxel_index[off] > n_voxel_words)
            return fail_with(WV_E_INVALID_ARGUMENT, "voxel array: list outside the array");
        for (uint32_t k = 0; k < voxel_index[off]; ++k)
            if (voxel_index[off + 1 + k] >= n_triangles)
                return fail_with(WV_E_INVALID_ARGUMENT, "voxel array: list refers to a missing triangle");
    }
    return WV_OK;
}

// `set_node_inside` on arrays that already live on the device (stream-ordered, no synchronisation)
hipError_t nodes_inside_on_device(int nx, int ny, int nz, const float min_corner[3], float spacing,
                                  const uint32_t* d_voxel_index, const float aabb_min[3], const float aabb_max[3],
                                  uint32_t side, const uint32_t* d_triangles, const float* d_vertices, uint8_t* d_inside,
                                  hipStream_t stream) {
    InsideArgs a{};
    a.inside = d_inside;
    a.nx = nx;
    a.ny = ny;
    a.nz = nz;
    a.min_corner = {min_corner[0], min_corner[1], min_corner[2]};
    a.spacing = spacing;
    a.voxel_index = d_voxel_index;
    a.c0 = {aabb_min[0], aabb_min[1], aabb_min[2]};
    a.c1 = {aabb_max[0], aabb_max[1], aabb_max[2]};
    a.side = side;
    a.triangles = d_triangles;
    a.vertices = d_vertices;
    const size_t n = (size_t)nx * ny * nz;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(node_inside_kernel, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace wv

// `set_node_inside` over all nodes of the mesh described by (nx, ny, nz, min_corner, spacing).
extern "C" int wv_nodes_inside(int32_t nx, int32_t ny, int32_t nz, const float min_corner[3], float spacing,
                               const uint32_t* voxel_index, uint64_t n_voxel_words, const float aabb_min[3],
                               const float aabb_max[3], uint32_t side, const uint32_t* triangles, uint32_t n_triangles,
                               const float* vertices, uint32_t n_vertices, uint8_t* inside) {
    if (nx < 1 || ny < 1 || nz < 1 || !min_corner || !voxel_index || !aabb_min || !aabb_max || !triangles || !vertices ||
        !inside || side < 1 || n_voxel_words < (uint64_t)side * side * side)
        return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument");
    // the kernel follows these indices without checks: refuse a malformed scene here
    if (int rc = wv::validate_scene(voxel_index, n_voxel_words, side, triangles, n_triangles, n_vertices)) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return wv::fail_with(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    const size_t n = (size_t)nx * ny * nz;
    uint8_t* d_inside = nullptr;
    uint32_t *d_vox = nullptr, *d_tri = nullptr;
    float* d_vert = nullptr;
    hipError_t rc = hipMalloc((void**)&d_inside, n);
    if (rc == hipSuccess) rc = hipMalloc((void**)&d_vox, n_voxel_words * sizeof(uint32_t));
    if (rc == hipSuccess) rc = hipMalloc((void**)&d_tri, (size_t)std::max(n_triangles, 1u) * 16);
    if (rc == hipSuccess) rc = hipMalloc((void**)&d_vert, (size_t)std::max(n_vertices, 1u) * 16);
    if (rc == hipSuccess) rc = hipMemcpy(d_vox, voxel_index, n_voxel_words * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (rc == hipSuccess && n_triangles) rc = hipMemcpy(d_tri, triangles, (size_t)n_triangles * 16, hipMemcpyHostToDevice);
    if (rc == hipSuccess && n_vertices) rc = hipMemcpy(d_vert, vertices, (size_t)n_vertices * 16, hipMemcpyHostToDevice);
    if (rc == hipSuccess)
        rc = wv::nodes_inside_on_device(nx, ny, nz, min_corner, spacing, d_vox, aabb_min, aabb_max, side, d_tri, d_vert,
                                        d_inside, 0);
    if (rc == hipSuccess) rc = hipMemcpy(inside, d_inside, n, hipMemcpyDeviceToHost);
    (void)hipFree(d_inside);
    (void)hipFree(d_vox);
    (void)hipFree(d_tri);
    (void)hipFree(d_vert);
    if (rc != hipSuccess) return wv::fail_with(WV_E_HIP, hipGetErrorString(rc));
    return WV_OK;
}
