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
import re
import sys


DEFAULT_WORKLOAD = "1024x1024x1024 f64"


def workload_of(where):
    """The workload the counted command ran, from the bench line its log holds ("768x768x768 f64"); the bench default when there is no log."""
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            for line in reversed(open(os.path.join(where, "pmc_%s.log" % c)).read().splitlines()):
                if line.startswith("{") and '"config"' in line:
                    rec = json.loads(line)
                    m = re.match(r"(\d+x\d+x\d+) ", rec["config"]["workload"])
                    return "%s %s" % (m.group(1), rec["dtype"])
        except (OSError, ValueError, KeyError, AttributeError):
            pass
    return DEFAULT_WORKLOAD


def write_traffic_record(where, out, workload=DEFAULT_WORKLOAD):
    """profiles/traffic.json (+ a copy beside the summary): the measured HBM bytes per launch of the dominant kernel and of the two
    boundary launches of a pass, stamped with the device code they were measured on -- what bench.py quotes as roofline.traffic."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    total = lambda v: int(2 * v["FETCH_SIZE"]["mean_KiB"] * 1024 + v["WRITE_SIZE"]["mean_KiB"] * 1024)  # noqa: E731
    # the dominant kernel: the three-step march where the run took three-step passes (more launches of it than of the two-step march)
    march = []
    for name in ("triple_march_kernel", "pair_march_kernel"):
        march += [(name, k, v) for k, v in out.items() if k.startswith(name + "<") and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
    if not march:
        return
    march.sort(key=lambda m: -m[2]["FETCH_SIZE"]["n"])
    kernel, k, v = march[0]
    tag = os.path.basename(os.path.normpath(where))
    rec = {"workload": workload, "kernel": kernel, "kernel_full_name": "wv::" + k, "kernel_sources": bench.kernel_sources_hash(),
           "measured": tag, "files": "%s/pmc_summary.json" % tag,
           "source": "profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, kernel trace only, "
                     "launches of `bench.py --steps 30 --warmup 6`)" % tag,
           "FETCH_SIZE_KiB_raw": v["FETCH_SIZE"]["mean_KiB"], "WRITE_SIZE_KiB_raw": v["WRITE_SIZE"]["mean_KiB"],
           "corrections": "gfx950: FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads -> x2 (calibrated on the in-place triad "
                          "in tools/stream_bench.hip: 2*FETCH_SIZE*1024 = bytes read, exactly); WRITE_SIZE*1024 = bytes written, exactly; "
                          "Infinity-Cache hits are counted (fabric-side counter)",
           "hbm_bytes_per_launch": total(v)}
    # the boundary launches of a pass: boundary_kernel<.., false> steps the boundary nodes to t+1 (and, in a three-step pass, to t+3: the
    # mean is over both), <.., true> to t+2 and finishes the nodes its 1-D entries face
    level = {}
    for name, c in out.items():
        if name.startswith("boundary_kernel<") and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            level[1 if name.rstrip(">").rstrip().endswith("true") else 0] = total(c)
    if len(level) == 2:
        rec["boundary_hbm_bytes_per_launch"] = [level[0], level[1]]
    fix = [total(c) for name, c in out.items() if name.startswith("pair_fixup_kernel<") and "FETCH_SIZE" in c and "WRITE_SIZE" in c]
    if fix:
        rec["fixup_list_hbm_bytes_per_launch"] = fix[0]  # (a three-step pass: the mean over the second level's short list and the third's long one)
    # the record bench.py reads (profiles/traffic.json) is the default workload's; another size leaves its record beside its summary only
    paths = [os.path.join(where, "traffic.json")]
    if workload == DEFAULT_WORKLOAD:
        paths.append(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"))
    for path in paths:
        json.dump(rec, open(path, "w"), indent=1)
    print("traffic record:", json.dumps({k: rec[k] for k in ("kernel_sources", "hbm_bytes_per_launch") if k in rec}), rec.get("boundary_hbm_bytes_per_launch"))


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
    write_traffic_record(where, out, workload_of(where))
    for k, v in sorted(out.items()):
        if "boundary_kernel" in k or "pair_march" in k or "stream_sweep" in k or "triple_" in k or "pair_fixup" in k:
            f, w = v.get("FETCH_SIZE", {}).get("mean_KiB", 0), v.get("WRITE_SIZE", {}).get("mean_KiB", 0)
            print("%-50s fetched 2 x %.0f KiB = %.3f GB, written %.3f GB, total %.3f GB (%d launches)"
                  % (k, f, 2 * f * 1024 / 1e9, w * 1024 / 1e9, (2 * f + w) * 1024 / 1e9, v.get("FETCH_SIZE", {}).get("n", 0)))


if __name__ == "__main__":
    main()
