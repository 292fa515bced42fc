#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 `--pmc` passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) of the same
command, as tools/gpu_call.sh leaves them under <dir>/pmc_FETCH_SIZE and <dir>/pmc_WRITE_SIZE -> <dir>/pmc_summary.json.

gfx950 corrections (the guide's HBM / rocprofv3 section; calibrated for this engine's access widths in round 2, tools/fetch_calib.hip):
both counters are in KiB; FETCH_SIZE tallies a 128-byte line at 64 B, so fetched bytes = 2 x FETCH_SIZE x 1024.

    python tools/pmc_summary.py gpurun_out/r04
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    where = sys.argv[1]
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(where, "pmc_%s" % c, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    agg[r["Kernel_Name"].replace("void wv::", "").split("(")[0][:60]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                out.setdefault(k, {})[c] = {"mean_KiB": sum(v) / len(v), "n": len(v)}
    json.dump(out, open(os.path.join(where, "pmc_summary.json"), "w"), indent=1)
    for k, v in sorted(out.items()):
        if "boundary_kernel" in k or "pair_march" in k or "stream_sweep" in k:
            f, w = v.get("FETCH_SIZE", {}).get("mean_KiB", 0), v.get("WRITE_SIZE", {}).get("mean_KiB", 0)
            print("%-50s fetched 2 x %.0f KiB = %.3f GB, written %.3f GB, total %.3f GB (%d launches)"
                  % (k, f, 2 * f * 1024 / 1e9, w * 1024 / 1e9, (2 * f + w) * 1024 / 1e9, v.get("FETCH_SIZE", {}).get("n", 0)))


if __name__ == "__main__":
    main()
