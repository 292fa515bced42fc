/* oracle/node_inside_oracle.c -- CPU restatement of the reference's `set_node_inside` kernel and
 * the voxel / geometry helpers it calls.  TEST INFRASTRUCTURE ONLY, built into liboracle.so.
 *
 *   set_node_inside            src/waveguide/src/mesh_setup_program.cpp:110-140
 *   voxel_inside, single_ray_inside, count_intersections, the traversal macro
 *                              src/core/src/cl/voxel.cpp:16-66,98-225
 *   triangle_vert_intersection, is_degenerate, almost_equal
 *                              src/core/src/cl/geometry.cpp:7-56
 *
 * Pinned against oracle/_ref/libwvref_setup.so (that kernel text compiled for the host, with
 * dot / cross evaluated left to right as in oracle/ref_shim_setup.cpp) in tests/test_mesh_setup.py.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef struct {
    float x, y, z;
} v3;

static v3 sub(v3 a, v3 b) { return (v3){a.x - b.x, a.y - b.y, a.z - b.z}; }
static float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 cross(v3 a, v3 b) { return (v3){a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

static int almost_equal(float x, float y, float ulp) {
    const float abs_diff = fabsf(x - y);
    return abs_diff < FLT_EPSILON * fabsf(x + y) * ulp || abs_diff < FLT_MIN;
}

typedef struct {
    float t, u, v;
} inter_t;

static inter_t triangle_hit(v3 v0, v3 v1, v3 v2, v3 pos, v3 dir) {
    const inter_t none = {0, 0, 0};
    const v3 e0 = sub(v1, v0), e1 = sub(v2, v0);
    const v3 pvec = cross(dir, e1);
    const float det = dot(e0, pvec);
    if (almost_equal(det, 0.0f, 10.0f)) return none;
    const float invdet = 1.0f / det;
    const v3 tvec = sub(pos, v0);
    const float u = invdet * dot(tvec, pvec);
    if (u < 0.0f || 1.0f < u) return none;
    const v3 qvec = cross(tvec, e0);
    const float v = invdet * dot(dir, qvec);
    if (v < 0.0f || 1.0f < v + u) return none;
    const float t = invdet * dot(e1, qvec);
    if (t < 0 || almost_equal(t, 0.0f, 10.0f)) return none;
    return (inter_t){t, u, v};
}

typedef struct {
    const uint32_t* voxel_index;
    float c0[3], c1[3];
    uint32_t side;
    const uint32_t* triangles; /* {surface, v0, v1, v2} */
    const float* vertices;     /* 4 floats per vertex */
} scene_t;

/* data of src/core/src/cl/voxel.cpp:156-189 */
static const float k_directions[32][3] = {
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

static uint32_t count_crossings(const scene_t* s, v3 pos, v3 dir) {
    const float side_f = (float)s->side;
    const float p[3] = {pos.x, pos.y, pos.z}, d[3] = {dir.x, dir.y, dir.z};
    float vd[3];
    int ind[3];
    for (int i = 0; i < 3; ++i) {
        vd[i] = (s->c1[i] - s->c0[i]) / side_f;
        ind[i] = (int)floorf((p[i] - s->c0[i]) / vd[i]);
    }
    const int side = (int)s->side;
    if (ind[0] < 0 || ind[1] < 0 || ind[2] < 0 || ind[0] >= side || ind[1] >= side || ind[2] >= side) return 0;
    int step[3], just_out[3];
    float t_max[3], t_delta[3];
    for (int i = 0; i < 3; ++i) {
        const float lo = s->c0[i] + (float)(ind[i] + 0) * vd[i];
        const float hi = s->c0[i] + (float)(ind[i] + 1) * vd[i];
        const int neg = signbit(d[i]) != 0;
        step[i] = neg ? -1 : 1;
        just_out[i] = neg ? -1 : side;
        const float tm = fabsf(((neg ? lo : hi) - p[i]) / d[i]);
        t_max[i] = isnan(tm) ? INFINITY : tm;
        t_delta[i] = fabsf(vd[i] / d[i]);
    }
    uint32_t count = 0;
    float prev_max = 0;
    for (;;) {
        int min_i = 0;
        for (int i = 1; i != 3; ++i)
            if (t_max[i] < t_max[min_i]) min_i = i;
        const uint32_t off = s->voxel_index[(size_t)ind[0] * s->side * s->side + (size_t)ind[1] * s->side + ind[2]];
        const uint32_t num = s->voxel_index[off];
        const uint32_t* list = s->voxel_index + off + 1;
        const float max_dist = t_max[min_i];
        for (uint32_t i = 0; i != num; ++i) {
            const uint32_t* tri = s->triangles + 4 * (size_t)list[i];
            const float* a = s->vertices + 4 * (size_t)tri[1];
            const float* b = s->vertices + 4 * (size_t)tri[2];
            const float* c = s->vertices + 4 * (size_t)tri[3];
            const inter_t in = triangle_hit((v3){a[0], a[1], a[2]}, (v3){b[0], b[1], b[2]}, (v3){c[0], c[1], c[2]}, pos, dir);
            if (in.t) {
                if (almost_equal(in.u, 0.0f, 10.0f) || almost_equal(in.v, 0.0f, 10.0f) || almost_equal(in.u + in.v, 1.0f, 10.0f))
                    return ~(uint32_t)0;
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

/* inside[i] = 1 when node i lies inside the triangle soup, else 0 */
void wvo_nodes_inside(int nx, int ny, int nz, const float* min_corner, float spacing, const uint32_t* voxel_index,
                      const float* aabb_min, const float* aabb_max, uint32_t side, const uint32_t* triangles,
                      const float* vertices, uint8_t* inside) {
    scene_t s;
    s.voxel_index = voxel_index;
    for (int k = 0; k < 3; ++k) {
        s.c0[k] = aabb_min[k];
        s.c1[k] = aabb_max[k];
    }
    s.side = side;
    s.triangles = triangles;
    s.vertices = vertices;
    const size_t n = (size_t)nx * ny * nz;
#pragma omp parallel for schedule(dynamic, 256)
    for (size_t i = 0; i < n; ++i) {
        const int x = (int)(i % nx);
        const size_t q = i / nx;
        const int y = (int)(q % ny), z = (int)((q / ny) % nz);
        const v3 pos = {min_corner[0] + (float)x * spacing, min_corner[1] + (float)y * spacing,
                        min_corner[2] + (float)z * spacing};
        uint8_t result = 0;
        for (int k = 0; k < 32; ++k) {
            const uint32_t c = count_crossings(&s, pos, (v3){k_directions[k][0], k_directions[k][1], k_directions[k][2]});
            if (c != ~(uint32_t)0) {
                result = (uint8_t)(c % 2);
                break;
            }
        }
        inside[i] = result;
    }
}
