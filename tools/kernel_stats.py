#!/usr/bin/env python3
"""Print the kernels of a rocprofv3 `--kernel-trace --stats` run (…kernel_stats.csv), names shortened:
    python tools/kernel_stats.py <dir or csv> [substring ...]"""
import csv
import glob
import os
import sys


def main():
    where = sys.argv[1]
    wanted = sys.argv[2:]
    files = [where] if where.endswith(".csv") else glob.glob(os.path.join(where, "**", "*kernel_stats.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void wv::", "").replace("wv::", "")
            name = name.split("(")[0]
            if wanted and not any(w in name for w in wanted):
                continue
            print("%-62s calls %6s  avg %10.1f us  total %10.3f ms  %5.1f %%" % (name[:62], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                              float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))


if __name__ == "__main__":
    main()
