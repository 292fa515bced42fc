/* oracle/waveguide_oracle.c -- CPU restatement of wayverb's `waveguide::run` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker and the `cpu_baseline` leg of
 * bench.py; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it.
 * The product library (wayverb_amd/csrc) never links, loads or calls anything in oracle/.
 *
 * Pinning: there are no numeric golden vectors for this path in the reference's own tests
 * (SURVEY.md 8(c)); the restatement is pinned instead against the reference's own kernel
 * text, compiled for the host by oracle/build_ref.py (oracle/_ref), bit for bit in fp32 and
 * in the fp64-promoted variant -- tests/test_oracle_vs_ref.py -- and against the committed
 * fixtures under tests/golden/ that oracle/_ref generated.
 *
 * Data contract = the reference's device structs
 * (src/waveguide/include/waveguide/cl/structs.h:19-58, cl/filter_structs.h:39-66):
 *   condensed_node {int32 boundary_type; uint32 boundary_index}            8 B
 *   boundary_data  {double filter_memory[6]; uint32 coefficient_index}    56 B
 *   coefficients   {double b[7]; double a[7]}                            112 B
 * Node index = x + y*nx + z*nx*ny (src/waveguide/src/cl/utils.cpp:33-36).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define WVO_CAT2(a, b) a##b
#define WVO_CAT(a, b) WVO_CAT2(a, b)

/* boundary_type bits, src/waveguide/include/waveguide/cl/utils.h:11-21 */
enum {
    WVO_ID_NONE = 0,
    WVO_ID_INSIDE = 1 << 0,
    WVO_ID_NX = 1 << 1,
    WVO_ID_PX = 1 << 2,
    WVO_ID_NY = 1 << 3,
    WVO_ID_PY = 1 << 4,
    WVO_ID_NZ = 1 << 5,
    WVO_ID_PZ = 1 << 6,
    WVO_ID_REENTRANT = 1 << 7,
};

/* error_code bits, src/waveguide/include/waveguide/cl/structs.h:8-15 */
enum {
    WVO_ERR_INF = 1 << 0,
    WVO_ERR_NAN = 1 << 1,
    WVO_ERR_OUTSIDE_RANGE = 1 << 2,
    WVO_ERR_OUTSIDE_MESH = 1 << 3,
    WVO_ERR_SUSPICIOUS_BOUNDARY = 1 << 4,
};

typedef struct {
    int32_t boundary_type;
    uint32_t boundary_index;
} wvo_condensed_node;

typedef struct {
    double filter_memory[6];
    uint32_t coefficient_index;
    uint32_t pad_;
} wvo_boundary_data;

typedef struct {
    double b[7];
    double a[7];
} wvo_coefficients;

typedef struct {
    int nx, ny, nz;
} wvo_dims;

_Static_assert(sizeof(wvo_condensed_node) == 8, "condensed_node is 8 bytes");
_Static_assert(sizeof(wvo_boundary_data) == 56, "boundary_data is 56 bytes");
_Static_assert(sizeof(wvo_coefficients) == 112, "coefficients_canonical is 112 bytes");

/* Inner-node directions of a boundary type: one port per set direction bit, x before y before
 * z; the port index of bit (1 << (p+1)) is p (utils.h:23-25).  Any type that is not exactly D
 * direction bits on D distinct axes gets -1 in every slot, like the reference's `default:`
 * (src/waveguide/src/program.cpp:19-87). */
static void wvo_inner_directions(int D, int32_t btype, int ind[3]) {
    int n = 0, axes = 0;
    ind[0] = ind[1] = ind[2] = -1;
    int tmp[3];
    for (int axis = 0; axis < 3; ++axis) {
        const int nbit = 1 << (1 + 2 * axis), pbit = 1 << (2 + 2 * axis);
        const int has_n = (btype & nbit) != 0, has_p = (btype & pbit) != 0;
        if (has_n && has_p) return; /* both sides of one axis: not a valid boundary type */
        if (has_n || has_p) {
            if (n < 3) tmp[n] = 2 * axis + (has_p ? 1 : 0);
            ++n;
            ++axes;
        }
    }
    if (n != D || (btype & ~0x7e)) return;
    for (int i = 0; i < D; ++i) ind[i] = tmp[i];
    (void)axes;
}

/* Ports whose pressures are summed un-doubled around a boundary node
 * (on_boundary_1 program.cpp:112-130, on_boundary_2 :132-143, none for D=3 :214-227). */
static int wvo_surrounding_ports(int D, const int ind[3], int ports[4]) {
    if (D == 1) {
        const int axis = ind[0] < 0 ? -1 : ind[0] / 2;
        if (axis < 0) {
            ports[0] = ports[1] = ports[2] = ports[3] = -1;
            return 4;
        }
        int n = 0;
        for (int a = 0; a < 3; ++a) {
            if (a == axis) continue;
            ports[n++] = 2 * a;
            ports[n++] = 2 * a + 1;
        }
        return 4;
    }
    if (D == 2) {
        const int has_x = (ind[0] == 0 || ind[0] == 1 || ind[1] == 0 || ind[1] == 1);
        const int has_y = (ind[0] == 2 || ind[0] == 3 || ind[1] == 2 || ind[1] == 3);
        int axis;
        if (has_x) {
            axis = has_y ? 2 : 1;
        } else {
            axis = 0;
        }
        ports[0] = 2 * axis;
        ports[1] = 2 * axis + 1;
        return 2;
    }
    return 0;
}

/* Order-6 transposed-direct-form-II step with zero-coefficient guards
 * (src/waveguide/src/cl/filters.cpp:17-36 instantiated at order 6, :39). */
static double wvo_filter_step_6(double input, double* m, const wvo_coefficients* c) {
    const double output = (input * c->b[0] + m[0]) / c->a[0];
    for (int i = 0; i < 5; ++i) {
        const double b = c->b[i + 1] == 0 ? 0 : c->b[i + 1] * input;
        const double a = c->a[i + 1] == 0 ? 0 : c->a[i + 1] * output;
        m[i] = b - a + m[i + 1];
    }
    const double b = c->b[6] == 0 ? 0 : c->b[6] * input;
    const double a = c->a[6] == 0 ? 0 : c->a[6] * output;
    m[5] = b - a;
    return output;
}

/* `filter_test_2` (filters.cpp:66-75): one canonical filter per work-item, float in / float out. */
void wvo_filter_test_2(const float* input, float* output, double* memory /* [n][6] */,
                       const wvo_coefficients* coeffs, int n) {
    for (int i = 0; i < n; ++i) {
        output[i] = (float)wvo_filter_step_6((double)input[i], memory + 6 * (size_t)i, coeffs + i);
    }
}

/* Order-2 step and the 3-section cascade behind `filter_test` (filters.cpp:17-36 at order 2,
 * :44-54, :56-64).  memory: [n][3][2] doubles; coeffs: [n][3]{b[3],a[3]}. */
static double wvo_filter_step_2(double input, double* m, const double* cb, const double* ca) {
    const double output = (input * cb[0] + m[0]) / ca[0];
    {
        const double b = cb[1] == 0 ? 0 : cb[1] * input;
        const double a = ca[1] == 0 ? 0 : ca[1] * output;
        m[0] = b - a + m[1];
    }
    {
        const double b = cb[2] == 0 ? 0 : cb[2] * input;
        const double a = ca[2] == 0 ? 0 : ca[2] * output;
        m[1] = b - a;
    }
    return output;
}

void wvo_filter_test(const float* input, float* output, double* memory, const double* coeffs, int n) {
    for (int i = 0; i < n; ++i) {
        double v = (double)input[i];
        for (int s = 0; s < 3; ++s) {
            double* m = memory + ((size_t)i * 3 + s) * 2;
            const double* c = coeffs + ((size_t)i * 3 + s) * 6;
            v = wvo_filter_step_2(v, m, c, c + 3);
        }
        output[i] = (float)v; /* biquad_cascade returns float, filters.cpp:44 */
    }
}

/* Host arithmetic of postprocessor::directional_receiver
 * (src/waveguide/src/postprocessor/directional_receiver.cpp:29-67), applied to a recorded
 * trace p7[step][7] = {centre, nx, px, ny, py, nz, pz} of float pressures.
 * out[step][4] = {intensity.x, intensity.y, intensity.z, pressure} as floats (glm::vec3 + float,
 * directional_receiver.h:30-33). */
void wvo_directional_receiver(const float* p7, int64_t steps, double mesh_spacing, double sample_rate,
                              double ambient_density, float* out) {
    double vel[3] = {0, 0, 0};
    for (int64_t s = 0; s < steps; ++s) {
        const float p = p7[s * 7];
        float d[6];
        for (int i = 0; i < 6; ++i) {
            /* (float - float) promoted to double, divided by the double spacing, stored as float */
            d[i] = (float)((double)(p7[s * 7 + 1 + i] - p) / mesh_spacing);
        }
        const double m[3] = {(double)(d[1] - d[0]) * 0.5, (double)(d[3] - d[2]) * 0.5,
                             (double)(d[5] - d[4]) * 0.5};
        const double k = ambient_density * sample_rate;
        for (int a = 0; a < 3; ++a) vel[a] -= m[a] / k;
        for (int a = 0; a < 3; ++a) out[s * 4 + a] = (float)(vel[a] * (double)p);
        out[s * 4 + 3] = p;
    }
}

/* ---- the two precision instantiations -------------------------------------------------- */
#define REAL float
#define SFX f32
#define SQRT_REAL sqrtf
#include "waveguide_oracle_body.h"
#undef REAL
#undef SFX
#undef SQRT_REAL

#define REAL double
#define SFX f64
#define SQRT_REAL sqrt
#include "waveguide_oracle_body.h"
#undef REAL
#undef SFX
#undef SQRT_REAL
