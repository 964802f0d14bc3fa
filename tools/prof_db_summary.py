#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 run kept in its rocpd SQLite form (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`
writes DIR/NAME_results.db on this image): calls, total, average, share -- the text that goes under profiles/.

    python tools/prof_db_summary.py <dir or .db> [top N] > profiles/rNN_xxx_kernel_stats.txt
"""
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"HIP_vector_type<(\w+), (\d)u>", r"\1\2", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    src = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    dbs = [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
    for f in dbs:
        db = sqlite3.connect(f)
        rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        tot = sum(r[2] for r in rows)
        print(f"# kernel stats: {os.path.basename(f)}  (durations in microseconds; all kernels together {tot / 1e3:.1f} ms)")
        print(f"{'kernel':<70} {'calls':>7} {'total_ms':>11} {'avg_us':>12} {'pct':>7}")
        for name, calls, total, avg, pct in rows[:top]:
            print(f"{short(name):<70} {calls:>7} {total / 1e3:>11.2f} {avg:>12.1f} {pct:>7.2f}")


if __name__ == "__main__":
    main()
