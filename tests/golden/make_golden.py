#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the cases of cases.py through oracle/_ref -- the
reference's own OpenCL C kernel compiled for the host (oracle/build_ref.py).  Run in the build
container (needs /root/reference to build oracle/_ref); the .npz files are committed, this
script documents how they were made.  Nothing from the repo's own oracle or engine is used.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import build_ref  # noqa: E402
from oracle.oracle import Reference  # noqa: E402
import cases  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(ref, case):
    mesh = case["mesh"]
    dt = ref.dtype
    n = mesh.num_nodes
    if case["init"] is None:
        prev = np.zeros(n, dtype=dt)
        cur = np.zeros(n, dtype=dt)
    else:
        prev = case["init"][0].astype(dt)
        cur = case["init"][1].astype(dt)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    steps, flag, out = ref.run(prev, cur, mesh, bd, case["source_kind"], case["source_node"],
                               case["signal"], case["steps"], case["recv"])
    assert steps == case["steps"] and flag == 0, (steps, flag)
    # after an even number of swaps buf1 (`cur`) is `current` again
    final_cur, final_prev = (cur, prev) if steps % 2 == 0 else (prev, cur)
    return dict(trace=out, sha_current=sha(final_cur), sha_previous=sha(final_prev),
                sha_bd=[sha(b) for b in bd], final_current=final_cur)


def main():
    build_ref.build()
    for name, make in cases.CASES.items():
        case = make()
        out = {}
        for tag in ("f32", "f64"):
            r = run_case(Reference(tag), case)
            out["trace_" + tag] = r["trace"]
            out["sha_current_" + tag] = r["sha_current"]
            out["sha_previous_" + tag] = r["sha_previous"]
            out["sha_bd_" + tag] = np.array(r["sha_bd"])
            if name == "random":
                out["final_current_" + tag] = r["final_current"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items()
                     if k.startswith("trace")})
    for quiet in (False, True):
        c = cases.case_filters(quiet)
        ref = Reference("f32")
        mem = np.zeros((256, 6), dtype=np.float64)
        outs = np.zeros_like(c["input"])
        for s in range(c["input"].shape[0]):
            outs[s] = ref.filter_test_2(c["input"][s], mem, c["coeffs"])
        name = "filters_quiet" if quiet else "filters_noise"
        np.savez_compressed(os.path.join(HERE, name + ".npz"), last_output=outs[-1],
                            sha_outputs=sha(outs), final_memory=mem)
        print(name, "finite:", bool(np.isfinite(outs).all()))


if __name__ == "__main__":
    main()
