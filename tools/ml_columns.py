#!/usr/bin/env python
"""Per-column cost of the Householder reduction from a `rocprofv3 --kernel-trace --output-format csv` run of an ML pass:
the first long run of alternating k_td_col / k_td_trail* launches (one chunk of order-Np matrices), sampled every
`step` columns: kernel times, and the idle time of the queue between them.

    python tools/ml_columns.py <rocprof output dir> [step]
"""
import csv
import re
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows]
q0 = next(q for n, s, e, q in ev if "k_td_col" in n)
main = [(n, s, e) for n, s, e, q in ev if q == q0]
# runs of reduction kernels
runs, cur = [], []
for n, s, e in main:
    if "k_td_col" in n or "k_td_trail" in n:
        cur.append((n, s, e))
    elif cur:
        runs.append(cur)
        cur = []
if cur:
    runs.append(cur)
runs.sort(key=lambda r: sum(e - s for _, s, e in r), reverse=True)
run = runs[0]
cols = [i for i, (n, _, _) in enumerate(run) if "k_td_col" in n]
print(f"{len(runs)} reduction runs; longest: {len(cols)} columns, {(run[-1][2] - run[0][1]) / 1e6:.1f} ms wall, "
      f"kernel time {sum(e - s for _, s, e in run) / 1e6:.1f} ms")
print("column   k_td_col us   sweeps us (n)    idle us")
tot_col = tot_sw = tot_idle = 0.0
for ci, i in enumerate(cols):
    j = cols[ci + 1] if ci + 1 < len(cols) else len(run)
    tcol = (run[i][2] - run[i][1]) / 1e3
    sw = run[i + 1:j]
    tsw = sum(e - s for _, s, e in sw) / 1e3
    end_next = run[j][1] if j < len(run) else run[-1][2]
    idle = (end_next - run[i][1]) / 1e3 - tcol - tsw
    tot_col += tcol
    tot_sw += tsw
    tot_idle += idle
    if ci % step == 0 or ci % step == 1 or ci % step == 2 or ci % step == 3:
        kinds = ",".join("r" if re.search(r"k_td_trail_tri<\d+, 0,", n) else "A" for n, _, _ in sw)
        print(f"{ci:6d} {tcol:12.1f} {tsw:12.1f} ({kinds}) {idle:10.1f}")
print(f"totals: k_td_col {tot_col / 1e3:.1f} ms, sweeps {tot_sw / 1e3:.1f} ms, idle {tot_idle / 1e3:.1f} ms")
