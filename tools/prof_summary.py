#!/usr/bin/env python
"""Condense rocprofv3 CSV output into a small text summary for profiles/.

    python tools/prof_summary.py <rocprof output dir> [more dirs ...] > profiles/rNN_xxx.txt

Handles `*_kernel_stats.csv` (from --kernel-trace --stats) and `*_counter_collection.csv`
(from --pmc): per kernel the mean counter value per dispatch.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"HIP_vector_type<(\w+), (\d)u>", r"\1\2", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^()]*>)?)", name)
    s = m.group(1) if m else name
    return s[:90]


def main():
    for d in sys.argv[1:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)):
            print(f"# kernel stats: {f}")
            print(f"{'kernel':<60} {'calls':>6} {'avg_us':>12} {'min_us':>12} {'max_us':>12} {'pct':>7}")
            for r in csv.DictReader(open(f)):
                print(f"{short(r['Name']):<60} {r['Calls']:>6} {float(r['AverageNs'])/1e3:>12.1f} {float(r['MinNs'])/1e3:>12.1f} {float(r['MaxNs'])/1e3:>12.1f} {float(r['Percentage']):>7.2f}")
            print()
        for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
            print(f"# counters: {f}")
            acc = defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                k = (short(r["Kernel_Name"]), r["Counter_Name"])
                acc[k][0] += float(r["Counter_Value"])
                acc[k][1] += 1
            print(f"{'kernel':<60} {'counter':<24} {'dispatches':>10} {'mean per dispatch':>20}")
            for (k, c), (s, n) in sorted(acc.items()):
                print(f"{k:<60} {c:<24} {n:>10} {s/n:>20.1f}")
            print()


if __name__ == "__main__":
    main()
