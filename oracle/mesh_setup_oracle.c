/* oracle/mesh_setup_oracle.c -- CPU restatement of the node classification half of wayverb's
 * mesh set-up.  TEST INFRASTRUCTURE ONLY, built into liboracle.so.
 *
 * Pinned against oracle/_ref/libwvref_setup.so (the reference's own `set_node_boundary_type`
 * kernel text compiled for the host) in tests/test_mesh_setup.py.
 */
#include <stddef.h>
#include <stdint.h>

typedef struct {
    int32_t boundary_type;
    uint32_t boundary_index;
} wvo_node;

enum { ID_INSIDE = 1, ID_REENTRANT = 128 };

/* Unit offset of a direction-bit set: one step per set bit, n* bits negative
 * (relative_locator, src/waveguide/src/mesh_setup_program.cpp:14-36). */
static void relative_locator(int dir, int rel[3]) {
    rel[0] = ((dir >> 2) & 1) - ((dir >> 1) & 1);
    rel[1] = ((dir >> 4) & 1) - ((dir >> 3) & 1);
    rel[2] = ((dir >> 6) & 1) - ((dir >> 5) & 1);
}

/* The direction tables of mesh_setup_program.cpp:38-64, generated: D set bits on D distinct axes,
 * enumerated in the reference's order (lowest axis pair first; n before p on each axis). */
static int directions(int D, int out[12]) {
    static const int axis_bits[3][2] = {{1 << 1, 1 << 2}, {1 << 3, 1 << 4}, {1 << 5, 1 << 6}};
    int n = 0;
    if (D == 1) {
        for (int a = 0; a < 3; ++a)
            for (int s = 0; s < 2; ++s) out[n++] = axis_bits[a][s];
    } else if (D == 2) {
        static const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int p = 0; p < 3; ++p)
            for (int s0 = 0; s0 < 2; ++s0)
                for (int s1 = 0; s1 < 2; ++s1) out[n++] = axis_bits[pairs[p][0]][s0] | axis_bits[pairs[p][1]][s1];
    } else {
        for (int s0 = 0; s0 < 2; ++s0)
            for (int s1 = 0; s1 < 2; ++s1)
                for (int s2 = 0; s2 < 2; ++s2) out[n++] = axis_bits[0][s0] | axis_bits[1][s1] | axis_bits[2][s2];
    }
    return n;
}

/* test_directions (mesh_setup_program.cpp:66-108): the single direction whose neighbour is an
 * inside node, id_reentrant if there are several, 0 if none. */
static int test_directions(const wvo_node* nodes, int nx, int ny, int nz, int x, int y, int z, const int* dirs, int n) {
    int ret = 0;
    for (int i = 0; i < n; ++i) {
        int rel[3];
        relative_locator(dirs[i], rel);
        const int ax = x + rel[0], ay = y + rel[1], az = z + rel[2];
        if (ax < 0 || ay < 0 || az < 0 || ax >= nx || ay >= ny || az >= nz) continue;
        if (nodes[(size_t)ax + (size_t)ay * nx + (size_t)az * nx * ny].boundary_type == ID_INSIDE) {
            if (ret != 0) return ID_REENTRANT;
            ret = dirs[i];
        }
    }
    return ret;
}

/* `set_node_boundary_type` over every node (mesh_setup_program.cpp:142-172): nodes arrive with
 * boundary_type = id_inside or 0; outside nodes receive their 1-D, else 2-D, else 3-D type. */
void wvo_set_node_boundary_type(wvo_node* nodes, int nx, int ny, int nz) {
    int d1[12], d2[12], d3[12];
    const int n1 = directions(1, d1), n2 = directions(2, d2), n3 = directions(3, d3);
    /* in place, like the reference: neighbours are only ever compared against id_inside, which
     * no outside node becomes, so already-classified neighbours cannot be mistaken for inputs */
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                wvo_node* node = nodes + ((size_t)x + (size_t)y * nx + (size_t)z * nx * ny);
                if (node->boundary_type & ID_INSIDE) continue;
                int t = test_directions(nodes, nx, ny, nz, x, y, z, d1, n1);
                if (!t) t = test_directions(nodes, nx, ny, nz, x, y, z, d2, n2);
                if (!t) t = test_directions(nodes, nx, ny, nz, x, y, z, d3, n3);
                if (t) node->boundary_type = t;
            }
}

static int popcount32(uint32_t v) { return __builtin_popcount(v); }

/* set_boundary_index as compute_boundary_index_data applies it
 * (src/waveguide/src/boundary_coefficient_finder.cpp:11-19,44-54;
 * predicates include/waveguide/boundary_coefficient_finder.h:16-27): running count, in node order,
 * of (1-D boundary OR re-entrant), of 2-D boundary, of 3-D boundary nodes.  counts[3] out. */
void wvo_set_boundary_indices(wvo_node* nodes, size_t n, uint64_t counts[3]) {
    counts[0] = counts[1] = counts[2] = 0;
    for (size_t i = 0; i < n; ++i) {
        const int32_t t = nodes[i].boundary_type;
        const int is_b = !((t & ID_REENTRANT) || (t & ID_INSIDE));
        const int bits = popcount32((uint32_t)t);
        if (t == ID_REENTRANT || (is_b && bits == 1)) {
            nodes[i].boundary_index = (uint32_t)counts[0]++;
        } else if (is_b && bits == 2) {
            nodes[i].boundary_index = (uint32_t)counts[1]++;
        } else if (is_b && bits == 3) {
            nodes[i].boundary_index = (uint32_t)counts[2]++;
        }
    }
}
