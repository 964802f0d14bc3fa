#!/usr/bin/env python
"""Timeline of an ML pass from a `rocprofv3 --kernel-trace --output-format csv` run: when the Gram launches (one per
chunk) and the serial QL launches start and end, and where the caller's queue sits idle for more than 2 ms.

    python tools/ml_timeline.py <rocprof output dir> <seconds of the timed pass (the end of the trace)>
"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows]
end = max(e[2] for e in ev)
t0 = end - int(float(sys.argv[2]) * 1e9)
ev = [e for e in ev if e[1] >= t0]


def short(n):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", n)
    return m.group(1)[:28] if m else n[:28]


mainq = next(q for n, s, e, q in ev if "k_td_col" in n or "k_nt" in n)
last = None
for n, s, e, q in ev:
    if q == mainq:
        if last and s - last[2] > 2e6:
            print(f"idle {(s - last[2]) / 1e6:7.1f} ms at t = {(last[2] - t0) / 1e6:8.1f} ms, after {short(last[0])}, before {short(n)}")
        last = (n, s, e)
for n, s, e, q in ev:
    if "k_td_solve<2>" in n:
        print(f"QL    queue {q}  {(s - t0) / 1e6:8.1f} .. {(e - t0) / 1e6:8.1f} ms")
    if re.search(r"k_nt<[03]>", n) and e - s > 5e6:
        print(f"Gram           {(s - t0) / 1e6:8.1f} .. {(e - t0) / 1e6:8.1f} ms")
print(f"caller's queue ends at {(last[2] - t0) / 1e6:.1f} ms, trace at {(end - t0) / 1e6:.1f} ms")
