/* oracle/boundary_surfaces_oracle.c -- CPU restatement of the reference's boundary coefficient
 * finder: which scene surface each boundary filter of the mesh takes.  TEST INFRASTRUCTURE ONLY,
 * built into liboracle.so.
 *
 *   point_triangle_distance_squared, slow_closest_triangle
 *                              src/waveguide/src/boundary_coefficient_program.cpp:12-143,218-235
 *   boundary_coefficient_finder_1d / _2d / _3d     same file :310-338, :356-413, :429-484
 *   compute_boundary_index_data (host)     src/waveguide/src/boundary_coefficient_finder.cpp:38-131
 *
 * Pinned against oracle/_ref/libwvref_bcf.so (those kernels compiled for the host) in
 * tests/test_mesh_setup.py.
 *
 * Two things the kernels do as written, both kept:
 *  - "1-D" is `popcount(boundary_type) == 1`, which is also true of id_inside (1) and
 *    id_reentrant (128) nodes.  The 2-D / 3-D kernels therefore accept an inside or re-entrant
 *    neighbour as the donor of a surface, and do not look at which direction it lies in: the
 *    first qualifying neighbour in table order donates to every port.
 *  - in the 1-D kernel every inside node (boundary_index 0) writes entry 0 of the 1-D array, racing
 *    with that entry's owner.  `entry0_last_writer` = 1 reproduces what a serial in-order
 *    execution leaves there (what _ref does); 0 gives entry 0 to its owner only, which is the
 *    deterministic choice the product makes.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t boundary_type;
    uint32_t boundary_index;
} node_t;

typedef struct {
    float x, y, z;
} v3;

static v3 sub(v3 a, v3 b) { return (v3){a.x - b.x, a.y - b.y, a.z - b.z}; }
static float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* parameter on one triangle edge: 0 at the near vertex, 1 at the far one */
static float along_edge(float b, float a) {
    if (0 <= b) return 0;
    if (a <= -b) return 1;
    return -b / a;
}

/* Squared distance from p to triangle (v0, v1, v2): closest point v0 + t0 e0 + t1 e1 by region of
 * the (t0, t1) plane.  boundary_coefficient_program.cpp:16-143 */
float wvo_point_triangle_dist2(const float* pv0, const float* pv1, const float* pv2, const float* pp) {
    const v3 v0 = {pv0[0], pv0[1], pv0[2]}, v1 = {pv1[0], pv1[1], pv1[2]}, v2 = {pv2[0], pv2[1], pv2[2]};
    const v3 p = {pp[0], pp[1], pp[2]};
    const v3 diff = sub(p, v0), e0 = sub(v1, v0), e1 = sub(v2, v0);
    const float a00 = dot(e0, e0), a01 = dot(e0, e1), a11 = dot(e1, e1);
    const float b0 = -dot(diff, e0), b1 = -dot(diff, e1);
    const float det = a00 * a11 - a01 * a01;
    float t0 = a01 * b1 - a11 * b0;
    float t1 = a01 * b0 - a00 * b1;

    if (t0 + t1 <= det) {
        if (t0 < 0) {
            if (t1 < 0 && b0 < 0) { /* behind v0, nearer edge e0 */
                t1 = 0;
                t0 = a00 <= -b0 ? 1 : -b0 / a00;
            } else { /* edge e1 */
                t0 = 0;
                t1 = along_edge(b1, a11);
            }
        } else if (t1 < 0) { /* edge e0 */
            t1 = 0;
            t0 = along_edge(b0, a00);
        } else { /* interior */
            const float inv = 1 / det;
            t0 *= inv;
            t1 *= inv;
        }
    } else if (t0 < 0) {
        const float m0 = a01 + b0, m1 = a11 + b1;
        if (m0 < m1) { /* hypotenuse */
            const float numer = m1 - m0, denom = a00 - 2 * a01 + a11;
            if (denom <= numer) {
                t0 = 1;
                t1 = 0;
            } else {
                t0 = numer / denom;
                t1 = 1 - t0;
            }
        } else {
            t0 = 0;
            t1 = m1 <= 0 ? 1 : along_edge(b1, a11);
        }
    } else if (t1 < 0) {
        const float m0 = a01 + b1, m1 = a00 + b0;
        if (m0 < m1) {
            const float numer = m1 - m0, denom = a00 - 2 * a01 + a11;
            if (denom <= numer) {
                t1 = 1;
                t0 = 0;
            } else {
                t1 = numer / denom;
                t0 = 1 - t1;
            }
        } else {
            t1 = 0;
            t0 = m1 <= 0 ? 1 : along_edge(b0, a00);
        }
    } else {
        const float numer = a11 + b1 - a01 - b0;
        if (numer <= 0) {
            t0 = 0;
            t1 = 1;
        } else {
            const float denom = a00 - 2 * a01 + a11;
            if (denom <= numer) {
                t0 = 1;
                t1 = 0;
            } else {
                t0 = numer / denom;
                t1 = 1 - t0;
            }
        }
    }
    const v3 closest = {v0.x + e0.x * t0 + e1.x * t1, v0.y + e0.y * t0 + e1.y * t1, v0.z + e0.z * t0 + e1.z * t1};
    const v3 d = sub(p, closest);
    return dot(d, d);
}

/* first triangle at the minimum distance, over the whole list */
static uint32_t nearest_triangle(v3 p, const uint32_t* triangles, uint32_t n_triangles, const float* vertices) {
    uint32_t best = 0;
    float best_d = INFINITY;
    const float pp[3] = {p.x, p.y, p.z};
    for (uint32_t i = 0; i != n_triangles; ++i) {
        const uint32_t* t = triangles + 4 * (size_t)i;
        const float d = wvo_point_triangle_dist2(vertices + 4 * (size_t)t[1], vertices + 4 * (size_t)t[2],
                                                 vertices + 4 * (size_t)t[3], pp);
        if (d < best_d) {
            best = i;
            best_d = d;
        }
    }
    return best;
}

static int popcount32(int32_t v) { return __builtin_popcount((uint32_t)v); }

static const int k_face_offsets[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
static const int k_edge_offsets[12][3] = {{-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 0, -1}, {-1, 0, 1},
                                          {1, 0, -1},  {1, 0, 1},  {0, -1, -1}, {0, -1, 1}, {0, 1, -1}, {0, 1, 1}};

/* Surfaces for the D-dimensional boundary nodes (D = 2, 3) from neighbouring "1-D" nodes. */
static void gather_from_neighbours(const node_t* nodes, int nx, int ny, int nz, int dim, const int (*offsets)[3],
                                   int n_offsets, const uint32_t* out1, uint32_t* out) {
    const size_t n = (size_t)nx * ny * nz;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const int32_t bt = nodes[i].boundary_type;
        if (popcount32(bt) != dim) continue;
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((size_t)nx * ny));
        uint32_t* row = out + (size_t)nodes[i].boundary_index * dim;
        int count = 0;
        for (int port = 0; port != 6; ++port) {
            if (!(bt & (1 << (port + 1)))) continue;
            for (int j = 0; j != n_offsets; ++j) {
                const int ax = x + offsets[j][0], ay = y + offsets[j][1], az = z + offsets[j][2];
                if (ax < 0 || ay < 0 || az < 0 || nx <= ax || ny <= ay || nz <= az) continue;
                const node_t* a = nodes + ((size_t)az * ny + ay) * nx + ax;
                if (popcount32(a->boundary_type) != 1) continue;
                row[count++] = out1[a->boundary_index];
                break;
            }
        }
    }
}

/* The three kernels.  `nodes` carry the first numbering (1-D index also counts re-entrant nodes);
 * out1 [n1], out2 [n2][2], out3 [n3][3]. */
void wvo_boundary_coefficient_finder(const node_t* nodes, int nx, int ny, int nz, float spacing,
                                     const float* min_corner, const uint32_t* triangles, uint32_t n_triangles,
                                     const float* vertices, uint32_t* out1, size_t n1, uint32_t* out2, size_t n2,
                                     uint32_t* out3, size_t n3, int entry0_last_writer) {
    const size_t n = (size_t)nx * ny * nz;
    memset(out1, 0, n1 * sizeof(uint32_t));
    memset(out2, 0, n2 * 2 * sizeof(uint32_t));
    memset(out3, 0, n3 * 3 * sizeof(uint32_t));
    size_t last_inside = n; /* last node that is inside: the serial winner of the race for entry 0 */
    size_t owner0 = n;      /* the boundary / re-entrant node numbered 0 */
#pragma omp parallel for schedule(dynamic, 1024)
    for (size_t i = 0; i < n; ++i) {
        const int32_t bt = nodes[i].boundary_type;
        if (popcount32(bt) != 1 || bt == 1) continue;
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((size_t)nx * ny));
        const v3 p = {min_corner[0] + (float)x * spacing, min_corner[1] + (float)y * spacing,
                      min_corner[2] + (float)z * spacing};
        out1[nodes[i].boundary_index] = triangles[4 * (size_t)nearest_triangle(p, triangles, n_triangles, vertices)];
    }
    if (entry0_last_writer) {
        for (size_t i = 0; i < n; ++i) {
            if (nodes[i].boundary_type == 1) last_inside = i;
            if (popcount32(nodes[i].boundary_type) == 1 && nodes[i].boundary_type != 1 && nodes[i].boundary_index == 0)
                owner0 = i;
        }
        if (last_inside != n && (owner0 == n || owner0 < last_inside) && n1) {
            const size_t i = last_inside;
            const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((size_t)nx * ny));
            const v3 p = {min_corner[0] + (float)x * spacing, min_corner[1] + (float)y * spacing,
                          min_corner[2] + (float)z * spacing};
            out1[0] = triangles[4 * (size_t)nearest_triangle(p, triangles, n_triangles, vertices)];
        }
    }
    gather_from_neighbours(nodes, nx, ny, nz, 2, k_face_offsets, 6, out1, out2);
    gather_from_neighbours(nodes, nx, ny, nz, 3, k_edge_offsets, 12, out1, out3);
}

/* compute_boundary_index_data: kernels, then drop the re-entrant slots from the 1-D array and
 * renumber the true 1-D nodes (boundary_coefficient_finder.cpp:91-103,128).  `nodes` in: types
 * set, indices ignored; out: final indices (re-entrant nodes keep their first-numbering index,
 * as in the reference).  b1/b2/b3 must hold counts_first[0], [1]*2, [2]*3 words where
 * counts_first is the first numbering's counts; counts[3] receives the final row counts. */
void wvo_boundary_index_data(node_t* nodes, int nx, int ny, int nz, float spacing, const float* min_corner,
                             const uint32_t* triangles, uint32_t n_triangles, const float* vertices, uint32_t* b1,
                             uint32_t* b2, uint32_t* b3, uint64_t counts[3], int entry0_last_writer) {
    const size_t n = (size_t)nx * ny * nz;
    uint32_t c[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        const int32_t bt = nodes[i].boundary_type;
        int d = -1;
        if (bt == 128)
            d = 0;
        else if (bt != 0 && !(bt & (1 | 128))) {
            const int bits = popcount32(bt);
            if (bits >= 1 && bits <= 3) d = bits - 1;
        }
        nodes[i].boundary_index = d >= 0 ? c[d]++ : 0u;
    }
    uint32_t* first = (uint32_t*)malloc(sizeof(uint32_t) * (c[0] ? c[0] : 1));
    wvo_boundary_coefficient_finder(nodes, nx, ny, nz, spacing, min_corner, triangles, n_triangles, vertices, first,
                                    c[0], b2, c[1], b3, c[2], entry0_last_writer);
    uint32_t kept = 0;
    for (size_t i = 0; i < n; ++i) {
        const int32_t bt = nodes[i].boundary_type;
        if (bt != 128 && !(bt & 1) && popcount32(bt) == 1) {
            b1[kept] = first[nodes[i].boundary_index];
            nodes[i].boundary_index = kept++;
        }
    }
    free(first);
    counts[0] = kept;
    counts[1] = c[1];
    counts[2] = c[2];
}
