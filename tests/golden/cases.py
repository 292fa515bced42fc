"""Definitions of the golden cases (inputs only).  Shared by make_golden.py (which runs them
through oracle/_ref, the reference's own kernel) and by the tests (which run them through the
C restatement and the HIP engine)."""
import numpy as np

from wayverb_amd import mesh as M


def case_impulse(walls):
    """G1: 32^3 box, unit hard-source impulse at the centre, 256 steps (BASELINE configs[0])."""
    if walls == "flat":
        coeffs = np.array([M.flat_coefficients(0.1)], dtype=M.coefficients_dtype)
    else:
        coeffs = np.array([M.rigid_coefficients()], dtype=M.coefficients_dtype)
    mesh = M.box_mesh(32, 32, 32, coefficients=coeffs)
    steps = 256
    signal = np.zeros(steps)
    signal[0] = 1.0
    ci = mesh.compute_index
    recv = [ci(16, 16, 16), ci(19, 16, 16), ci(16, 19, 16), ci(16, 16, 19), ci(2, 2, 2),
            ci(1, 16, 16), ci(16, 1, 16), ci(16, 16, 30), ci(1, 1, 16), ci(1, 1, 1)]
    return dict(mesh=mesh, steps=steps, source_kind=1, source_node=ci(16, 16, 16), signal=signal,
                recv=recv, init=None)


def case_random(seed=1234):
    """G2/G3: 24x20x28 box, seeded U(-0.25,0.25) field on inside nodes in both buffers
    (tests/rectangular_kernel.cpp:50 range), six different passive order-6 wall filters plus a
    soft source; 64 steps."""
    rng = np.random.default_rng(seed)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 3),
                             M.passive_peak_filter_coefficients(rng, 1, sections=2),
                             np.array([M.flat_coefficients(0.3), M.rigid_coefficients()],
                                      dtype=M.coefficients_dtype)])
    mesh = M.box_mesh(24, 20, 28, coefficients=coeffs, surface_of_face=[0, 1, 2, 3, 4, 5])
    inside = (mesh.nodes["boundary_type"] & M.ID_INSIDE) != 0
    prev = np.zeros(mesh.num_nodes)
    cur = np.zeros(mesh.num_nodes)
    prev[inside] = rng.uniform(-0.25, 0.25, int(inside.sum()))
    cur[inside] = rng.uniform(-0.25, 0.25, int(inside.sum()))
    steps = 64
    signal = rng.uniform(-0.1, 0.1, steps)
    ci = mesh.compute_index
    recv = [ci(12, 10, 14), ci(2, 2, 2), ci(1, 10, 14), ci(12, 1, 14), ci(12, 10, 26), ci(22, 18, 26),
            ci(1, 1, 14), ci(1, 18, 26)]
    return dict(mesh=mesh, steps=steps, source_kind=2, source_node=ci(7, 9, 11), signal=signal,
                recv=recv, init=(prev, cur))


def case_filters(quiet, seed=99):
    """G4: `filter_test_2` -- 256 independent order-6 filters, noise in [-r, r]
    (tests/rectangular_kernel.cpp:54-101,242-305): r = 0.25, or 1e-35 for the quiet variant."""
    rng = np.random.default_rng(seed)
    n = 256
    coeffs = M.random_peak_filter_coefficients(rng, n)
    samples = 1000
    r = 1e-35 if quiet else 0.25
    x = rng.uniform(-r, r, (samples, n)).astype(np.float32)
    return dict(coeffs=coeffs, input=x)


CASES = {
    "impulse_flat": lambda: case_impulse("flat"),
    "impulse_rigid": lambda: case_impulse("rigid"),
    "random": lambda: case_random(),
}
