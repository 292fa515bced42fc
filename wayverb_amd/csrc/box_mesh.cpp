// box_mesh.cpp -- host helper: the synthetic box mesh of SURVEY.md 8(d).
//
// What `compute_mesh` (src/waveguide/src/mesh.cpp:53-141) yields for a `geo::box` scene up to
// padding thickness: one id_none layer, one boundary shell whose type is the OR of the direction
// bits pointing at the adjacent inside node, id_inside elsewhere; boundary_index = running count
// per dimensionality in increasing node index (set_boundary_index,
// src/waveguide/src/boundary_coefficient_finder.cpp:11-19).  Slab-aware so that each rank of a
// z-decomposed run can build just its planes (plus ghosts) of a mesh too large for one index space.
#include <cstdint>
#include <thread>
#include <vector>

#include "../../include/wayverb_amd.h"

namespace {

inline int32_t axis_bits(int c, int n, int32_t p_bit, int32_t n_bit) {
    int32_t b = 0;
    if (c == 1) b |= p_bit;      // inside neighbour lies at c+1
    if (c == n - 2) b |= n_bit;  // inside neighbour lies at c-1
    return b;
}

inline int32_t box_type(int x, int y, int z, int nx, int ny, int nz) {
    if (x == 0 || y == 0 || z == 0 || x == nx - 1 || y == ny - 1 || z == nz - 1) return WV_ID_NONE;
    const int32_t t = axis_bits(x, nx, WV_ID_PX, WV_ID_NX) | axis_bits(y, ny, WV_ID_PY, WV_ID_NY) |
                      axis_bits(z, nz, WV_ID_PZ, WV_ID_NZ);
    return t ? t : WV_ID_INSIDE;
}

}  // namespace

extern "C" int wv_make_box_nodes(int32_t nx, int32_t ny, int32_t nz_global, int32_t z_begin, int32_t z_count,
                                 int32_t number_from, int32_t number_to, wv_condensed_node* nodes,
                                 uint64_t counts[3]) {
    if (nx < 5 || ny < 5 || nz_global < 5 || z_begin < 0 || z_count < 1 || z_begin + z_count > nz_global || !nodes)
        return WV_E_INVALID_ARGUMENT;
    const int64_t plane = (int64_t)nx * ny;
    // pass 1 (parallel over planes): types + per-plane boundary counts
    std::vector<uint64_t> per_plane((size_t)z_count * 3, 0);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int n_threads = (int)std::min<unsigned>(hw, (unsigned)z_count);
    auto pass1 = [&](int t) {
        for (int zi = t; zi < z_count; zi += n_threads) {
            const int z = z_begin + zi;
            uint64_t c[3] = {0, 0, 0};
            wv_condensed_node* out = nodes + (int64_t)zi * plane;
            for (int y = 0; y < ny; ++y)
                for (int x = 0; x < nx; ++x) {
                    const int32_t ty = box_type(x, y, z, nx, ny, nz_global);
                    out[(int64_t)y * nx + x].boundary_type = ty;
                    out[(int64_t)y * nx + x].boundary_index = 0;
                    if (ty != WV_ID_NONE && ty != WV_ID_INSIDE) c[__builtin_popcount((uint32_t)ty) - 1]++;
                }
            for (int d = 0; d < 3; ++d) per_plane[(size_t)zi * 3 + d] = c[d];
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(pass1, t);
        for (auto& th : pool) th.join();
    }
    // exclusive prefix over the numbered planes
    std::vector<uint64_t> start((size_t)z_count * 3, 0);
    uint64_t run[3] = {0, 0, 0};
    for (int zi = 0; zi < z_count; ++zi) {
        const int z = z_begin + zi;
        const bool numbered = z >= number_from && z < number_to;
        for (int d = 0; d < 3; ++d) {
            start[(size_t)zi * 3 + d] = run[d];
            if (numbered) run[d] += per_plane[(size_t)zi * 3 + d];
        }
    }
    if (counts)
        for (int d = 0; d < 3; ++d) counts[d] = run[d];
    // pass 2: boundary_index within the numbered planes
    auto pass2 = [&](int t) {
        for (int zi = t; zi < z_count; zi += n_threads) {
            const int z = z_begin + zi;
            if (z < number_from || z >= number_to) continue;
            uint64_t c[3] = {start[(size_t)zi * 3], start[(size_t)zi * 3 + 1], start[(size_t)zi * 3 + 2]};
            wv_condensed_node* out = nodes + (int64_t)zi * plane;
            for (int64_t i = 0; i < plane; ++i) {
                const int32_t ty = out[i].boundary_type;
                if (ty != WV_ID_NONE && ty != WV_ID_INSIDE)
                    out[i].boundary_index = (uint32_t)c[__builtin_popcount((uint32_t)ty) - 1]++;
            }
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(pass2, t);
        for (auto& th : pool) th.join();
    }
    return WV_OK;
}
