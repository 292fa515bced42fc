#!/usr/bin/env python3
"""Wall time of the scene -> mesh stages on one GPU, next to the CPU restatement (all host cores).

    python tools/setup_bench.py [--n 256] [--subdivisions 4] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E  # noqa: E402
from wayverb_amd import scene as S  # noqa: E402


def timed(f, *a, **k):
    t0 = time.perf_counter()
    r = f(*a, **k)
    return r, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256, help="approximate nodes per axis")
    ap.add_argument("--subdivisions", type=int, default=4)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    v, t = S.icosphere_scene((0.0, 0.0, 0.0), 1.0, args.subdivisions)
    t = t.copy()
    t[:, 0] = np.arange(t.shape[0]) % 5
    spacing = 2.2 / args.n
    lo, hi = S.padded_aabb(v, 0.1)
    dims = tuple(int(d) for d in ((hi - lo) / np.float32(spacing)).astype(np.int32))
    out = {"dims": dims, "nodes": int(np.prod(dims)), "triangles": int(t.shape[0])}
    E.nodes_inside((8, 8, 8), lo, 0.3, E.voxelise(v, t, (lo, hi), 4), (lo, hi), 4, t, v)   # warm the runtime up
    vox, out["voxelise_host_s"] = timed(E.voxelise, v, t, (lo, hi), 32)
    mask, out["nodes_inside_gpu_s"] = timed(E.nodes_inside, dims, lo, spacing, vox, (lo, hi), 32, t, v)
    (nodes, counts), out["classify_gpu_s"] = timed(E.classify_nodes, mask)
    b, out["boundary_index_data_gpu_s"] = timed(E.boundary_index_data, dims, lo, spacing, nodes, t, v)
    out["boundary_nodes"] = [int(x.shape[0]) for x in b]
    # the same chain resident on the device, and an engine built on it without a host round trip
    from wayverb_amd import mesh as M
    sm, out["scene_mesh_resident_gpu_s"] = timed(E.SceneMesh, dims, lo, spacing, vox, (lo, hi), 32, t, v)
    coeffs = M.bench_materials()
    eng, out["engine_from_resident_nodes_s"] = timed(sm.engine, np.concatenate([coeffs, coeffs[:1]]))
    eng.close()
    mesh = M.Mesh(dims, nodes, np.concatenate([coeffs, coeffs[:1]]), b[0], b[1], b[2])
    eng, out["engine_from_host_nodes_s"] = timed(E.Engine, mesh)
    eng.close()
    if not args.no_cpu:
        r_nodes, rb = sm.fetch()
        out["resident_identical_to_staged"] = bool(r_nodes.tobytes() == nodes.tobytes()
                                                   and all(np.array_equal(x, y) for x, y in zip(rb, b)))
    sm.close()
    if not args.no_cpu:
        from oracle.oracle import Oracle
        o = Oracle()
        o_mask, out["nodes_inside_cpu_s"] = timed(o.nodes_inside, dims, lo, spacing, vox, (lo, hi), 32, t, v)
        (o_nodes, o_counts), out["classify_cpu_s"] = timed(o.classify, o_mask.astype(bool))
        ob, out["boundary_index_data_cpu_s"] = timed(o.boundary_index_data, o_nodes, dims, lo, spacing, t, v)
        out["cpu_threads"] = os.cpu_count()
        out["identical"] = bool(np.array_equal(mask, o_mask) and nodes.tobytes() == o_nodes.tobytes()
                                and all(np.array_equal(x, y) for x, y in zip(b, ob)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
