/* oracle/waveguide_oracle_body.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Type-generic body of the CPU restatement; included twice by waveguide_oracle.c with
 *   REAL = float  , SFX = f32   (bit-exact restatement of the reference as written)
 *   REAL = double , SFX = f64   (the reference with its pressure type promoted to double)
 *
 * Written from the arithmetic description in SURVEY.md Appendix A; each function cites the
 * reference lines whose behaviour it restates (paths relative to /root/reference/).
 * FILT = double everywhere (`filt_real`, src/waveguide/include/waveguide/cl/filter_structs.h:14).
 */

#define FN(name) WVO_CAT(WVO_CAT(name, _), SFX)

/* courant = 1/sqrt(3), courant_sq = 1/3, both evaluated in the pressure type
 * (src/waveguide/src/program.cpp:12-13). */
static inline REAL FN(courant)(void) { return (REAL)1 / (REAL)SQRT_REAL((REAL)3); }
static inline REAL FN(courant_sq)(void) { return (REAL)1 / (REAL)3; }

/* Pressure at the neighbour of `loc` through `port`; port -1 (the reference's "no such
 * direction" value, program.cpp:30,63,86) leaves the locator unchanged
 * (src/waveguide/src/cl/utils.cpp:38-69).  Returns 0 and sets *missing when off-grid. */
static inline int64_t FN(nbr)(const wvo_dims* d, int x, int y, int z, int port) {
    switch (port) {
        case 0: x -= 1; break;
        case 1: x += 1; break;
        case 2: y -= 1; break;
        case 3: y += 1; break;
        case 4: z -= 1; break;
        case 5: z += 1; break;
        default: break;
    }
    if (x < 0 || y < 0 || z < 0 || x >= d->nx || y >= d->ny || z >= d->nz) return -1;
    return (int64_t)x + (int64_t)y * d->nx + (int64_t)z * d->nx * (int64_t)d->ny;
}

/* Interior update (program.cpp:393-412): sum in-grid neighbours in port order, true
 * division by 3, subtract previous. */
static inline REAL FN(interior)(const REAL* cur, REAL prev, const wvo_dims* d, int x, int y, int z) {
    REAL s = 0;
    for (int port = 0; port < 6; ++port) {
        const int64_t n = FN(nbr)(d, x, y, z, port);
        if (n >= 0) s += cur[n];
    }
    s /= (REAL)3;
    s -= prev;
    return s;
}

/* One boundary node of dimensionality D (program.cpp:331-387 and its helpers :150-327).
 * `bd` points at this node's D consecutive boundary_data records. */
static REAL FN(boundary)(int D, const REAL* cur, REAL prev, int32_t btype,
                         const wvo_condensed_node* nodes, const wvo_dims* d, int x, int y, int z,
                         wvo_boundary_data* bd, const wvo_coefficients* coeffs, int* flag) {
    int ind[3];
    wvo_inner_directions(D, btype, ind);

    /* 2 * inner pressures (program.cpp:268-276, get_inner_pressure :231-249) */
    REAL sum = 0;
    for (int i = 0; i < D; ++i) {
        const int64_t n = FN(nbr)(d, x, y, z, ind[i]);
        REAL p = 0;
        if (n < 0) {
            *flag |= WVO_ERR_OUTSIDE_MESH;
        } else {
            p = cur[n];
        }
        sum += 2 * p;
    }

    /* surrounding ports (program.cpp:178-227): early-out with 0 on a missing neighbour */
    REAL surr = 0;
    {
        int ports[4];
        const int np = wvo_surrounding_ports(D, ind, ports);
        for (int i = 0; i < np; ++i) {
            const int64_t n = FN(nbr)(d, x, y, z, ports[i]);
            if (n < 0) {
                *flag |= WVO_ERR_OUTSIDE_MESH;
                surr = 0;
                break;
            }
            const int32_t nt = nodes[n].boundary_type;
            if (nt == WVO_ID_NONE || nt == WVO_ID_INSIDE) *flag |= WVO_ERR_SUSPICIOUS_BOUNDARY;
            surr += cur[n];
        }
    }
    const REAL csw = FN(courant_sq)() * (sum + surr);

    /* filter weighting (program.cpp:286-305): REAL accumulator, double addends */
    REAL facc = 0;
    for (int i = 0; i < D; ++i) {
        const double m0 = bd[i].filter_memory[0];
        facc = (REAL)((double)facc + m0 / coeffs[bd[i].coefficient_index].b[0]);
    }
    const REAL fw = FN(courant_sq)() * facc;

    /* coefficient weighting (program.cpp:309-327) */
    REAL cacc = 0;
    for (int i = 0; i < D; ++i) {
        const wvo_coefficients* c = coeffs + bd[i].coefficient_index;
        cacc = (REAL)((double)cacc + c->a[0] / c->b[0]);
    }
    const REAL cw = cacc * FN(courant)();

    /* program.cpp:363-366 */
    const REAL pw = (cw - 1) * prev;
    const REAL next = (csw + fw + pw) / (1 + cw);

    /* ghost-point filter update (program.cpp:367-381, :150-174) */
    for (int i = 0; i < D; ++i) {
        const wvo_coefficients* c = coeffs + bd[i].coefficient_index;
        /* the reference evaluates get_inner_pressure again here; its value is unused but
         * its off-grid check still raises the flag */
        if (FN(nbr)(d, x, y, z, ind[i]) < 0) *flag |= WVO_ERR_OUTSIDE_MESH;
        const double m0 = bd[i].filter_memory[0];
        const double b0 = c->b[0];
        const double a0 = c->a[0];
        const double diff = (a0 * (double)(REAL)(prev - next)) / (b0 * (double)FN(courant)()) + (m0 / b0);
        wvo_filter_step_6(-diff, bd[i].filter_memory, c);
    }
    return next;
}

/* One node (program.cpp:414-487 dispatch, :494-530 kernel tail). */
static inline void FN(node)(int64_t index, REAL* previous, const REAL* cur,
                            const wvo_condensed_node* nodes, const wvo_dims* d,
                            wvo_boundary_data* b1, wvo_boundary_data* b2, wvo_boundary_data* b3,
                            const wvo_coefficients* coeffs, int* flag) {
    const wvo_condensed_node node = nodes[index];
    /* to_locator, src/waveguide/src/cl/utils.cpp:25-31 */
    const int x = (int)(index % d->nx);
    const int64_t q = index / d->nx;
    const int y = (int)(q % d->ny);
    const int z = (int)((q / d->ny) % d->nz);
    const REAL prev = previous[index];
    REAL next = 0;
    const int bits = __builtin_popcount((uint32_t)node.boundary_type);
    if (bits == 1) {
        if (node.boundary_type & (WVO_ID_INSIDE | WVO_ID_REENTRANT)) {
            next = FN(interior)(cur, prev, d, x, y, z);
        } else {
            next = FN(boundary)(1, cur, prev, node.boundary_type, nodes, d, x, y, z,
                                b1 + (size_t)node.boundary_index, coeffs, flag);
        }
    } else if (bits == 2) {
        next = FN(boundary)(2, cur, prev, node.boundary_type, nodes, d, x, y, z,
                            b2 + 2 * (size_t)node.boundary_index, coeffs, flag);
    } else if (bits == 3) {
        next = FN(boundary)(3, cur, prev, node.boundary_type, nodes, d, x, y, z,
                            b3 + 3 * (size_t)node.boundary_index, coeffs, flag);
    }
    if (isinf(next)) *flag |= WVO_ERR_INF;
    if (isnan(next)) *flag |= WVO_ERR_NAN;
    previous[index] = next;
}

/* One launch over the nodes of planes [z_begin, z_end): previous <- next in place
 * (src/waveguide/include/waveguide/waveguide.h:85-97 covers all planes; the sub-range form
 * serves the z-slab tests, where ghost planes are read but never updated).
 * threads>1: static z-chunk partition. */
int FN(wvo_step_range)(REAL* previous, const REAL* current, const wvo_condensed_node* nodes, int nx,
                       int ny, int nz, wvo_boundary_data* b1, wvo_boundary_data* b2,
                       wvo_boundary_data* b3, const wvo_coefficients* coeffs, int z_begin, int z_end,
                       int threads) {
    const wvo_dims d = {nx, ny, nz};
    int flag = 0;
    if (threads < 1) threads = 1;
    /* static partition over x-rows (z-major), so thin slabs still use every thread */
    const int64_t row0 = (int64_t)z_begin * ny, row1 = (int64_t)z_end * ny;
#pragma omp parallel for schedule(static) num_threads(threads) reduction(| : flag)
    for (int64_t row = row0; row < row1; ++row) {
        int local = 0;
        for (int64_t i = row * nx; i < (row + 1) * nx; ++i) {
            FN(node)(i, previous, current, nodes, &d, b1, b2, b3, coeffs, &local);
        }
        flag |= local;
    }
    return flag;
}

int FN(wvo_step)(REAL* previous, const REAL* current, const wvo_condensed_node* nodes, int nx, int ny,
                 int nz, wvo_boundary_data* b1, wvo_boundary_data* b2, wvo_boundary_data* b3,
                 const wvo_coefficients* coeffs, int threads) {
    return FN(wvo_step_range)(previous, current, nodes, nx, ny, nz, b1, b2, b3, coeffs, 0, nz, threads);
}

/* The run loop (waveguide.h:80-125) with the single-node source / node-gather receivers that
 * every in-repo caller uses (SURVEY.md 8(b) usage census):
 *   source_kind 0: none, 1: hard (preprocessor/hard_source.h:17-23), 2: soft (soft_source.h:17-25)
 *   recv[r]: node indices whose pre-update `current` value is recorded each step
 *            (postprocessor/node.cpp:14-18; Q1 in SURVEY.md App. D)
 * Runs while step < n_steps (for kinds 1/2 n_steps should equal the signal length: an exhausted
 * source ends the run).  Returns the number of completed steps; *flag_out receives the error
 * bits of the failing step (0 if none).  buf0 is `previous`, buf1 is `current` on entry; on
 * return the roles are as after the last swap. */
int64_t FN(wvo_run)(REAL* buf0, REAL* buf1, const wvo_condensed_node* nodes, int nx, int ny, int nz,
                    wvo_boundary_data* b1, wvo_boundary_data* b2, wvo_boundary_data* b3,
                    const wvo_coefficients* coeffs, int source_kind, int64_t source_node,
                    const double* signal, int64_t n_steps, const int64_t* recv, int n_recv,
                    REAL* out, int threads, int* flag_out) {
    REAL* previous = buf0;
    REAL* current = buf1;
    *flag_out = 0;
    int64_t step = 0;
    for (; step < n_steps; ++step) {
        if (source_kind == 1) {
            current[source_node] = (REAL)signal[step];
        } else if (source_kind == 2) {
            current[source_node] = current[source_node] + (REAL)signal[step];
        }
        const int flag = FN(wvo_step)(previous, current, nodes, nx, ny, nz, b1, b2, b3, coeffs, threads);
        if (flag) {
            *flag_out = flag;
            return step;
        }
        for (int r = 0; r < n_recv; ++r) out[step * n_recv + r] = current[recv[r]];
        REAL* t = previous;
        previous = current;
        current = t;
    }
    return step;
}

#undef FN
