/* examples/slab_chain.c -- a box room cut into K z-slabs that live in ONE process (plain C99): one engine per slab, on the
 * GPUs that are there (slab r on device r % n_devices), joined by wv_comm_init_local and stepped together by
 * wv_run_group -- the step code of the one-rank-per-GPU RCCL chain with device-to-device copies for the face
 * planes.  The receiver trace must equal the single-domain run's exactly.
 *
 *   gcc -std=c99 -Iinclude examples/slab_chain.c -Lwayverb_amd -lwayverb_amd -Wl,-rpath,$PWD/wayverb_amd -lm -o slab_chain
 *   ./slab_chain [K = 4] [n_devices = 1]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wayverb_amd.h"

#define CHECK(call)                                                         \
    do {                                                                    \
        if ((call) != WV_OK) {                                              \
            fprintf(stderr, "%s failed: %s\n", #call, wv_last_error());     \
            return 2;                                                       \
        }                                                                   \
    } while (0)

enum { NX = 48, NY = 40, NZ = 64, STEPS = 150, MAX_SLABS = 16 };

/* planes [z0, z0 + planes) of the global box as one engine; ghost planes are part of the range */
static int make_engine(int z0, int planes, int own0, int own1, int ghost_lo, int ghost_hi, int device,
                       const wv_coefficients_canonical* wall, wv_engine** out) {
    const size_t n = (size_t)NX * NY * planes;
    wv_condensed_node* nodes = malloc(n * sizeof *nodes);
    uint64_t counts[3];
    /* boundary nodes are numbered over the OWNED planes only: ghost-plane walls belong to the neighbour */
    CHECK(wv_make_box_nodes(NX, NY, NZ, z0, planes, own0, own1, nodes, counts));
    uint32_t* b1 = calloc(counts[0] ? counts[0] : 1, sizeof *b1); /* every filter: surface 0 */
    uint32_t* b2 = calloc(counts[1] ? counts[1] * 2 : 1, sizeof *b2);
    uint32_t* b3 = calloc(counts[2] ? counts[2] * 3 : 1, sizeof *b3);
    wv_mesh mesh = {NX, NY, planes, nodes, wall, 1, b1, b2, b3, counts[0], counts[1], counts[2]};
    wv_options opt;
    wv_default_options(&opt);
    opt.precision = WV_PRECISION_F64;
    opt.device = device;
    opt.ghost_lo = ghost_lo;
    opt.ghost_hi = ghost_hi;
    const int rc = wv_create(&mesh, &opt, out);
    free(nodes);
    free(b1);
    free(b2);
    free(b3);
    return rc;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4, n_devices = argc > 2 ? atoi(argv[2]) : 1;
    if (K < 1 || K > MAX_SLABS || K > NZ || n_devices < 1) {
        fprintf(stderr, "usage: slab_chain [slabs 1..%d] [devices]\n", MAX_SLABS);
        return 2;
    }
    wv_coefficients_canonical refl = {{0}, {0}}, wall;
    refl.b[0] = sqrt(1.0 - 0.2);
    refl.a[0] = 1.0;
    CHECK(wv_impedance_coefficients(&refl, &wall));
    double signal[STEPS] = {1.0};
    const uint64_t plane = (uint64_t)NX * NY;
    const uint64_t source = (uint64_t)(NZ / 2) * plane + (uint64_t)(NY / 2) * NX + NX / 2;
    const uint64_t receiver = (uint64_t)(NZ / 2 + 9) * plane + (uint64_t)(NY / 2 + 3) * NX + NX / 2 - 4;

    /* the single domain */
    wv_engine* whole = NULL;
    CHECK(make_engine(0, NZ, 0, NZ, 0, 0, 0, &wall, &whole));
    CHECK(wv_set_source(whole, WV_SOURCE_HARD, source, signal, STEPS));
    CHECK(wv_set_receivers(whole, &receiver, 1));
    uint64_t done = 0;
    int32_t flag = 0;
    CHECK(wv_run(whole, STEPS, &done, &flag));
    double want[STEPS], got[STEPS];
    CHECK(wv_fetch_receivers(whole, 0, STEPS, want));
    wv_destroy(whole);
    if (done != STEPS || flag != WV_FLAG_SUCCESS) return 1;

    /* the chain: slab r owns planes [z0, z1), plus one ghost plane towards each neighbour */
    wv_engine* slab[MAX_SLABS] = {0};
    int holder = -1;
    for (int r = 0; r < K; ++r) {
        const int base = NZ / K, extra = NZ % K;
        const int z0 = r * base + (r < extra ? r : extra), z1 = z0 + base + (r < extra ? 1 : 0);
        const int lo = r > 0, hi = r + 1 < K;
        CHECK(make_engine(z0 - lo, z1 - z0 + lo + hi, z0, z1, lo, hi, r % n_devices, &wall, &slab[r]));
        const uint64_t first = (uint64_t)(z0 - lo) * plane, last = (uint64_t)(z1 + hi) * plane; /* planes held */
        /* the source is injected by every slab that HOLDS its plane (owner and ghost copy alike) ... */
        if (source >= first && source < last) CHECK(wv_set_source(slab[r], WV_SOURCE_HARD, source - first, signal, STEPS));
        /* ... a receiver is recorded by the slab that OWNS it */
        if (receiver >= (uint64_t)z0 * plane && receiver < (uint64_t)z1 * plane) {
            const uint64_t local = receiver - first;
            CHECK(wv_set_receivers(slab[r], &local, 1));
            holder = r;
        }
    }
    CHECK(wv_comm_init_local(slab, K));
    CHECK(wv_run_group(slab, K, STEPS, &done, &flag));
    if (done != STEPS || flag != WV_FLAG_SUCCESS || holder < 0) return 1;
    CHECK(wv_fetch_receivers(slab[holder], 0, STEPS, got));
    for (int r = 0; r < K; ++r) CHECK(wv_comm_destroy(slab[r]));
    for (int r = 0; r < K; ++r) wv_destroy(slab[r]);
    const int same = memcmp(want, got, sizeof want) == 0;
    double peak = 0;
    int first = -1;
    for (int i = 0; i < STEPS; ++i) {
        if (first < 0 && got[i] != 0) first = i;
        if (fabs(got[i]) > peak) peak = fabs(got[i]);
    }
    printf("%d slabs on %d device(s), %d steps: receiver trace %s the single domain's (first arrival at step %d, peak |p| = %.6e)\n",
           K, n_devices, STEPS, same ? "identical to" : "DIFFERS from", first, peak);
    return same && first == 9 + 3 + 4 ? 0 : 1;
}
