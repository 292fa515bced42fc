#!/usr/bin/env python3
"""What one two-step pass looks like on the device: from a rocprofv3 `--kernel-trace` (+ `--memory-copy-trace`) run, the
kernels and copies between two consecutive launches of the march, with start offsets, durations and the idle gaps
between them -- per hardware queue, for the queue that ran the most marches (one slab's compute stream).

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o t -- python tools/slab_overhead.py ...
    python tools/pass_timeline.py <dir> [which_pass]
"""
import csv
import glob
import os
import sys


def main():
    where = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    for f in glob.glob(os.path.join(where, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void wv::", "").replace("wv::", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%s" % r.get("Queue_Id", "?"), name))
    for f in glob.glob(os.path.join(where, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", "memcpy %s" % r.get("Direction", "")))
    rows.sort()
    short = [(s, e) for s, e, q, name in rows if name.startswith(("pair_march", "triple_march")) and e - s < 2_000_000]
    if len(short) >= 2 * which_slabs(rows):
        # a chain of slabs on one device: from the first slab's march of one pass to its march of the next
        n = which_slabs(rows)
        k = min(which, len(short) // n - 2)
        t0, t1 = short[k * n][0], short[(k + 1) * n][0]
        print("chain of %d slabs: pass %d, %.1f us from the first slab's march to its next; everything on the device in that window:" % (n, k, (t1 - t0) / 1e3))
    else:
        marches = {}
        for s, e, q, name in rows:
            if name.startswith(("pair_march", "triple_march")):
                marches.setdefault(q, []).append(s)
        if not marches:
            print("no march in the trace")
            return
        q = max(marches, key=lambda k: len(marches[k]))
        starts = marches[q]
        if len(starts) <= which + 1:
            which = max(0, len(starts) - 2)
        t0, t1 = starts[which], starts[which + 1]
        print("queue %s: pass %d of %d, %.1f us from march to march; everything on the device in that window:" % (q, which, len(starts), (t1 - t0) / 1e3))
    last_end = None
    busy = 0
    for s, e, qq, name in rows:
        if s < t0 or s >= t1:
            continue
        gap = (s - last_end) / 1e3 if last_end is not None else 0.0
        last_end = max(e, last_end or 0)
        busy += e - s
        print("  %-6s +%9.1f us  %8.1f us  gap %7.1f us  %s" % (qq, (s - t0) / 1e3, (e - s) / 1e3, gap, name[:70]))
    print("sum of kernel durations in the window: %.1f us" % (busy / 1e3))


def which_slabs(rows):
    """Number of slabs = short marches between two long gaps ... taken from the environment (PASS_TIMELINE_SLABS), default 8."""
    return int(os.environ.get("PASS_TIMELINE_SLABS", "8"))


if __name__ == "__main__":
    main()
