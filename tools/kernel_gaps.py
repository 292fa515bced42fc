#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 `--kernel-trace` run: per kernel name, the number of launches, the average
duration and the average gap between the previous kernel's end and this one's start (gaps above --max-gap-us -- host synchronisation,
set-up -- are left out).  What a pass costs beyond the sum of its kernels' durations.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python bench.py ...
    python tools/kernel_gaps.py <dir> [--max-gap-us 2000]
"""
import csv
import glob
import os
import sys


def main():
    where = sys.argv[1]
    max_gap = 2000.0
    if "--max-gap-us" in sys.argv:
        max_gap = float(sys.argv[sys.argv.index("--max-gap-us") + 1])
    rows = []
    for f in glob.glob(os.path.join(where, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void wv::", "").replace("wv::", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    acc = {}
    last_end = None
    for s, e, name in rows:
        a = acc.setdefault(name, [0, 0.0, 0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        if last_end is not None:
            gap = (s - last_end) / 1e3
            if gap <= max_gap:
                a[2] += 1
                a[3] += gap
        last_end = max(e, last_end or 0)
    print("%-62s %7s %11s %11s %9s" % ("kernel", "calls", "avg us", "gap before", "(counted)"))
    for name, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-62s %7d %11.1f %11.1f %9d" % (name[:62], a[0], a[1] / a[0], a[3] / max(1, a[2]), a[2]))


if __name__ == "__main__":
    main()
