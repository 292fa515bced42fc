// mesh_setup.hip -- node classification for arbitrary (non-box) rooms: SURVEY.md 8(f) rank 1,
// first slice.
//
// Replaces the reference's `set_node_boundary_type` kernel
// (src/waveguide/src/mesh_setup_program.cpp:66-108,142-172) and the host numbering
// `set_boundary_index` as compute_boundary_index_data applies it
// (src/waveguide/src/boundary_coefficient_finder.cpp:11-19,44-54).  Input is the per-node
// inside flag (what `set_node_inside` produces, mesh_setup_program.cpp:110-140); output is the
// `condensed_node` array `wv_create` consumes.
//
// An outside node takes, in this order of preference, the single axial / edge-diagonal /
// corner-diagonal direction in which an inside node lies; several inside nodes at the same
// distance make it re-entrant; none leaves it id_none.  Integer work, one node per lane, the 26
// neighbour flags come from the byte mask through L2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wayverb_amd.h"

namespace {

__device__ __forceinline__ int probe(const uint8_t* inside, int nx, int ny, int nz, int x, int y, int z, int dir) {
    // relative_locator: one step per set direction bit, n* bits negative
    const int ax = x + ((dir >> 2) & 1) - ((dir >> 1) & 1);
    const int ay = y + ((dir >> 4) & 1) - ((dir >> 3) & 1);
    const int az = z + ((dir >> 6) & 1) - ((dir >> 5) & 1);
    if (ax < 0 || ay < 0 || az < 0 || ax >= nx || ay >= ny || az >= nz) return 0;
    return inside[(size_t)ax + (size_t)ay * nx + (size_t)az * nx * ny] != 0;
}

__global__ void __launch_bounds__(256) node_boundary_type_kernel(const uint8_t* inside, int32_t* type, int nx, int ny,
                                                                 int nz) {
    const int64_t n = (int64_t)nx * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (inside[i]) {
            type[i] = WV_ID_INSIDE;
            continue;
        }
        const int x = (int)(i % nx);
        const int64_t q = i / nx;
        const int y = (int)(q % ny), z = (int)(q / ny);
        const int bit[3][2] = {{WV_ID_NX, WV_ID_PX}, {WV_ID_NY, WV_ID_PY}, {WV_ID_NZ, WV_ID_PZ}};
        int result = WV_ID_NONE;
        // D = 1: the six axial neighbours
        int found = 0;
        for (int a = 0; a < 3; ++a)
            for (int s = 0; s < 2; ++s)
                if (probe(inside, nx, ny, nz, x, y, z, bit[a][s])) {
                    result = found ? WV_ID_REENTRANT : bit[a][s];
                    ++found;
                }
        if (!found) {  // D = 2: the twelve edge diagonals
            for (int a = 0; a < 3; ++a)
                for (int b = a + 1; b < 3; ++b)
                    for (int s = 0; s < 4; ++s) {
                        const int d = bit[a][s >> 1] | bit[b][s & 1];
                        if (probe(inside, nx, ny, nz, x, y, z, d)) {
                            result = found ? WV_ID_REENTRANT : d;
                            ++found;
                        }
                    }
        }
        if (!found) {  // D = 3: the eight corner diagonals
            for (int s = 0; s < 8; ++s) {
                const int d = bit[0][(s >> 2) & 1] | bit[1][(s >> 1) & 1] | bit[2][s & 1];
                if (probe(inside, nx, ny, nz, x, y, z, d)) {
                    result = found ? WV_ID_REENTRANT : d;
                    ++found;
                }
            }
        }
        type[i] = result;
    }
}

}  // namespace

namespace wv {
int fail_with(int code, const std::string& msg);  // engine.hip

// `set_node_boundary_type` on device arrays (stream-ordered); shared with scene_mesh.hip
hipError_t node_types_on_device(const uint8_t* d_inside, int32_t* d_type, int nx, int ny, int nz, hipStream_t stream) {
    const size_t n = (size_t)nx * ny * nz;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(node_boundary_type_kernel, dim3(grid), dim3(256), 0, stream, d_inside, d_type, nx, ny, nz);
    return hipGetLastError();
}
}  // namespace wv

extern "C" int wv_classify_nodes(int32_t nx, int32_t ny, int32_t nz, const uint8_t* inside, wv_condensed_node* nodes,
                                 uint64_t counts[3]) {
    if (nx < 1 || ny < 1 || nz < 1 || !inside || !nodes) return wv::fail_with(WV_E_INVALID_ARGUMENT, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return wv::fail_with(WV_E_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    const size_t n = (size_t)nx * ny * nz;
    uint8_t* d_in = nullptr;
    int32_t* d_type = nullptr;
    if (hipMalloc((void**)&d_in, n) != hipSuccess) return wv::fail_with(WV_E_HIP, "hipMalloc failed");
    if (hipMalloc((void**)&d_type, n * sizeof(int32_t)) != hipSuccess) {
        (void)hipFree(d_in);
        return wv::fail_with(WV_E_HIP, "hipMalloc failed");
    }
    std::vector<int32_t> type(n);
    hipError_t rc = hipMemcpy(d_in, inside, n, hipMemcpyHostToDevice);
    if (rc == hipSuccess) {
        const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 65536);
        hipLaunchKernelGGL(node_boundary_type_kernel, dim3(grid), dim3(256), 0, 0, d_in, d_type, nx, ny, nz);
        rc = hipGetLastError();
    }
    if (rc == hipSuccess) rc = hipMemcpy(type.data(), d_type, n * sizeof(int32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_type);
    if (rc != hipSuccess) return wv::fail_with(WV_E_HIP, hipGetErrorString(rc));

    // running counts in node order: (1-D boundary or re-entrant), 2-D, 3-D
    // (boundary_coefficient_finder.h:16-27)
    const int64_t plane = (int64_t)nx * ny;
    std::vector<uint64_t> per_plane((size_t)nz * 3, 0);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int n_threads = (int)std::min<unsigned>(hw, (unsigned)nz);
    auto dim_of = [](int32_t t) -> int {
        if (t == WV_ID_REENTRANT) return 0;
        if (t == WV_ID_NONE || (t & (WV_ID_INSIDE | WV_ID_REENTRANT))) return -1;
        const int bits = __builtin_popcount((uint32_t)t);
        return bits >= 1 && bits <= 3 ? bits - 1 : -1;
    };
    auto pass = [&](int t, bool assign, const std::vector<uint64_t>& start) {
        for (int z = t; z < nz; z += n_threads) {
            uint64_t c[3] = {0, 0, 0};
            if (assign)
                for (int d = 0; d < 3; ++d) c[d] = start[(size_t)z * 3 + d];
            for (int64_t i = (int64_t)z * plane; i < (int64_t)(z + 1) * plane; ++i) {
                const int d = dim_of(type[i]);
                if (assign) {
                    nodes[i].boundary_type = type[i];
                    nodes[i].boundary_index = d >= 0 ? (uint32_t)c[d]++ : 0u;
                } else if (d >= 0) {
                    ++c[d];
                }
            }
            if (!assign)
                for (int d = 0; d < 3; ++d) per_plane[(size_t)z * 3 + d] = c[d];
        }
    };
    std::vector<uint64_t> start((size_t)nz * 3, 0);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(pass, t, false, std::cref(start));
        for (auto& th : pool) th.join();
    }
    uint64_t run[3] = {0, 0, 0};
    for (int z = 0; z < nz; ++z)
        for (int d = 0; d < 3; ++d) {
            start[(size_t)z * 3 + d] = run[d];
            run[d] += per_plane[(size_t)z * 3 + d];
        }
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(pass, t, true, std::cref(start));
        for (auto& th : pool) th.join();
    }
    if (counts)
        for (int d = 0; d < 3; ++d) counts[d] = run[d];
    return WV_OK;
}
