#!/usr/bin/env python3
"""How the two-step march's units look on a real room (the concert hall at a given cutoff): per unit (a strip of 4 rows
through a chunk of 32 planes) the span of live waves, as wayverb_amd/csrc/engine_pair.hip.h (build_pair_units) computes it.
    python tools/hall_units.py [cutoff_hz=1600]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayverb_amd import simulation as sim, wayfile as W  # noqa: E402


def main():
    cutoff = float(sys.argv[1]) if len(sys.argv) > 1 else 1600.0
    cfg, v, t, absorptions = W.read_way(os.path.join(ROOT, "tests", "golden", "concert.way"))
    receiver = cfg["receivers"][0]["position"]
    vm = sim.compute_voxels_and_mesh(v, t, absorptions, receiver, sim.compute_sampling_frequency(cutoff, 0.6), 340.0)
    nx, ny, nz = vm.mesh.dims
    live = (vm.mesh.nodes["boundary_type"] != 0).reshape(nz, ny, nx)
    upd = ((vm.mesh.nodes["boundary_type"] & 0x81) != 0).reshape(nz, ny, nx)   # inside / re-entrant: what the march updates
    wave_cols, rows, planes = 128, 4, 32
    nw = (nx + wave_cols - 1) // wave_cols
    strips = (ny + rows - 1) // rows
    chunks = (nz + planes - 1) // planes
    pad = np.zeros((nz, strips * rows, nw * wave_cols), dtype=bool)
    pad[:, :ny, :nx] = live
    raw = pad.reshape(nz, strips, rows, nw, wave_cols).any(axis=(2, 4))          # [z, strip, wave]
    padu = np.zeros_like(pad)
    padu[:, :ny, :nx] = upd
    active = padu.reshape(nz, strips, rows, nw * wave_cols).any(axis=(2, 3))     # [z, strip]
    hist = np.zeros(nw + 1, dtype=np.int64)
    live_nodes = 0
    for s in range(strips):
        for c in range(chunks):
            zb, ze = c * planes, min((c + 1) * planes, nz)
            if not active[zb:ze, s].any():
                continue
            bits = raw[max(0, zb - 2):min(nz, ze + 2), max(0, s - 1):min(strips, s + 2)].any(axis=(0, 1))
            idx = np.nonzero(bits)[0]
            span = int(idx[-1] - idx[0] + 1) if idx.size else 1
            hist[span] += 1
            live_nodes += span * wave_cols * rows * (ze - zb)
    total_units = strips * chunks
    print("mesh %dx%dx%d, %d waves per row, %d strips x %d chunks = %d units, %d listed (%.1f %%)" % (nx, ny, nz, nw, strips, chunks, total_units, hist.sum(), 100.0 * hist.sum() / total_units))
    for k in range(1, nw + 1):
        print("   units with %d live wave(s): %6d (%.1f %% of the listed)" % (k, hist[k], 100.0 * hist[k] / max(1, hist.sum())))
    print("   mean live waves per listed unit %.2f of %d; visited nodes %.1f %% of the mesh; room %.1f %%" % (
        (hist * np.arange(nw + 1)).sum() / max(1, hist.sum()), nw, 100.0 * live_nodes / (nx * ny * nz), 100.0 * live.mean()))


if __name__ == "__main__":
    main()
