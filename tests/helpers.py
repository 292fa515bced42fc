"""Shared test plumbing: run a golden case through the oracle or the engine."""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def set_tuning(**fields):
    """Tuning (wv_tuning fields of include/wayverb_amd.h, plus stream_variant) of every engine created from now on, through
    wayverb_amd.engine.default_tuning -> wv_options::tuning.  No arguments: the library's defaults."""
    from wayverb_amd import engine as E
    E.default_tuning.clear()
    E.default_tuning.update({k: int(v) for k, v in fields.items()})


def initial_fields(case, dtype):
    n = case["mesh"].num_nodes
    if case["init"] is None:
        return np.zeros(n, dtype=dtype), np.zeros(n, dtype=dtype)
    return case["init"][0].astype(dtype), case["init"][1].astype(dtype)


def run_oracle(oracle, case, dtype, threads=1):
    mesh = case["mesh"]
    prev, cur = initial_fields(case, dtype)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    steps, flag, out = oracle.run(prev, cur, mesh, bd, case["source_kind"], case["source_node"],
                                  case["signal"], case["steps"], case["recv"], threads=threads)
    final_cur, final_prev = (cur, prev) if steps % 2 == 0 else (prev, cur)
    return dict(steps=steps, flag=flag, trace=out, current=final_cur, previous=final_prev, bd=bd)


def run_engine(case, precision, **engine_kw):
    from wayverb_amd import engine as E
    mesh = case["mesh"]
    dtype = np.float32 if precision == "f32" else np.float64
    eng = E.Engine(mesh, precision=precision, **engine_kw)
    try:
        if case["init"] is not None:
            prev, cur = initial_fields(case, dtype)
            eng.write_field(prev, E.BUF_PREVIOUS)
            eng.write_field(cur, E.BUF_CURRENT)
        steps, out = E.run_fast(eng, case["source_kind"], case["source_node"], case["signal"], case["recv"])
        return dict(steps=steps, flag=0, trace=out.astype(dtype),
                    current=eng.read_field(E.BUF_CURRENT), previous=eng.read_field(E.BUF_PREVIOUS),
                    bd=[eng.read_boundary_data(d) for d in (1, 2, 3)],
                    passes=eng.query(E.Engine.QUERY_PASSES), triple_passes=eng.query(E.Engine.QUERY_TRIPLE_PASSES),
                    xwall_entries=eng.query(E.Engine.QUERY_XWALL_ENTRIES))
    finally:
        eng.close()


def box_boundary_rows_below(nx, ny, nz, z, d):
    """Number of D-dimensional boundary nodes of an nx*ny*nz box in planes [0, z) = the boundary_index
    of the first such node of plane z (set_boundary_index numbers them in node order)."""
    per_face_plane = {1: (nx - 4) * (ny - 4), 2: 2 * (nx - 4) + 2 * (ny - 4), 3: 4}[d]
    per_mid_plane = {1: 2 * (nx - 4) + 2 * (ny - 4), 2: 4, 3: 0}[d]
    total = 0
    if z > 1:
        total += per_face_plane
    total += per_mid_plane * max(0, min(z, nz - 2) - 2)
    if z > nz - 2:
        total += per_face_plane
    return total
