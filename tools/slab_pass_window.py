#!/usr/bin/env python3
"""Everything on the device between two consecutive marches of the SAME slab of an in-process chain, from a rocprofv3 kernel trace of
tools/slab_overhead.py: which launches a slab's pass costs beside its march, and where the device idles.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python tools/slab_overhead.py --world 2 --steps 12
    python tools/slab_pass_window.py <dir> <slabs>"""
import csv
import glob
import os
import sys


def main():
    where, slabs = sys.argv[1], int(sys.argv[2])
    rows = []
    for f in glob.glob(os.path.join(where, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void wv::", "").replace("wv::", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%s" % r.get("Queue_Id", "?"), name))
    rows.sort()
    marches = [(s, e, q) for s, e, q, n in rows if n.startswith(("pair_march", "triple_march"))]
    longest = max(e - s for s, e, _ in marches)
    chain = [(s, e, q) for s, e, q in marches if e - s < 0.75 * longest]   # (the single domain's marches are the long ones)
    if len(chain) < 3 * slabs:
        print("no chain marches in the trace")
        return
    k = len(chain) - 2 * slabs            # the last but one pass of the first slab
    t0, t1 = chain[k][0], chain[k + slabs][0]
    print("%d slabs: %.1f us from the first slab's march to its next; everything on the device in that window:" % (slabs, (t1 - t0) / 1e3))
    last_end, busy = None, 0
    for s, e, q, name in rows:
        if s < t0 or s >= t1:
            continue
        gap = (s - last_end) / 1e3 if last_end is not None else 0.0
        last_end = max(e, last_end or 0)
        busy += e - s
        print("  %-4s +%9.1f us  %8.1f us  gap %7.1f us  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, name[:72]))
    print("sum of kernel durations in the window: %.1f us" % (busy / 1e3))


if __name__ == "__main__":
    main()
