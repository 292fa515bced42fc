/* examples/box_run.c -- the C ABI on its own (plain C99, no C++ mirror, no Python):
 * a 64^3 box room with one flat wall material, a hard-source impulse in the middle, one receiver,
 * 200 steps on the GPU, receiver trace printed.
 *
 *   gcc -std=c99 -Iinclude examples/box_run.c -Lwayverb_amd -lwayverb_amd -Wl,-rpath,$PWD/wayverb_amd -lm -o box_run
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "wayverb_amd.h"

#define CHECK(call)                                                         \
    do {                                                                    \
        if ((call) != WV_OK) {                                              \
            fprintf(stderr, "%s failed: %s\n", #call, wv_last_error());     \
            return 2;                                                       \
        }                                                                   \
    } while (0)

int main(void) {
    enum { N = 64, STEPS = 200 };
    const size_t n_nodes = (size_t)N * N * N;
    wv_condensed_node* nodes = malloc(n_nodes * sizeof *nodes);
    uint64_t counts[3];
    CHECK(wv_make_box_nodes(N, N, N, 0, N, 0, N, nodes, counts));

    /* to_flat_coefficients(0.1): reflectance sqrt(1 - a) -> impedance form (fitted_boundary.h:21-75) */
    wv_coefficients_canonical refl = {{0}, {0}}, wall;
    refl.b[0] = sqrt(1.0 - 0.1);
    refl.a[0] = 1.0;
    CHECK(wv_impedance_coefficients(&refl, &wall));

    uint32_t* b1 = calloc(counts[0] ? counts[0] : 1, sizeof *b1);      /* every filter: surface 0 */
    uint32_t* b2 = calloc(counts[1] ? counts[1] * 2 : 1, sizeof *b2);
    uint32_t* b3 = calloc(counts[2] ? counts[2] * 3 : 1, sizeof *b3);
    wv_mesh mesh = {N, N, N, nodes, &wall, 1, b1, b2, b3, counts[0], counts[1], counts[2]};

    wv_options opt;
    wv_default_options(&opt);
    opt.precision = WV_PRECISION_F64;
    wv_engine* e = NULL;
    CHECK(wv_create(&mesh, &opt, &e));

    double signal[STEPS] = {1.0};
    const uint64_t centre = (uint64_t)(N / 2) * N * N + (uint64_t)(N / 2) * N + N / 2;
    const uint64_t receiver = centre + 5;
    CHECK(wv_set_source(e, WV_SOURCE_HARD, centre, signal, STEPS));
    CHECK(wv_set_receivers(e, &receiver, 1));

    uint64_t done = 0;
    int32_t flag = 0;
    CHECK(wv_run(e, STEPS, &done, &flag));
    if (done != STEPS || flag != WV_FLAG_SUCCESS) {
        fprintf(stderr, "stopped after %llu steps, flag %d\n", (unsigned long long)done, flag);
        return 1;
    }
    double trace[STEPS];
    CHECK(wv_fetch_receivers(e, 0, STEPS, trace));
    double peak = 0;
    int first = -1;
    for (int i = 0; i < STEPS; ++i) {
        if (first < 0 && trace[i] != 0) first = i;
        if (fabs(trace[i]) > peak) peak = fabs(trace[i]);
    }
    /* the wave front moves one node per step along an axis: 5 nodes away -> first arrival at step 5 */
    printf("first arrival at step %d, peak |p| = %.6f, p[%d] = %.6e\n", first, peak, STEPS - 1, trace[STEPS - 1]);
    wv_destroy(e);
    free(nodes);
    free(b1);
    free(b2);
    free(b3);
    return first == 5 ? 0 : 1;
}
