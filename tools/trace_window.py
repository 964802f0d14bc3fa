#!/usr/bin/env python
"""Kernel time per name and queue inside the last `seconds` of a `rocprofv3 --kernel-trace --output-format csv` run
(the timed pass of the tools/*.py timing scripts is the last thing they do).

    python tools/trace_window.py <rocprof output dir> <seconds>
"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows]
end = max(e[2] for e in ev)
t0 = end - int(float(sys.argv[2]) * 1e9)
ev = [e for e in ev if e[1] >= t0]


def short(n):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", n)
    return m.group(1)[:44] if m else n[:44]


print(f"window {(end - t0) / 1e6:.1f} ms, {len(ev)} kernels")
byq = collections.defaultdict(list)
for n, s, e, q in ev:
    byq[q].append((s, e, short(n)))
for q, iv in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in iv)
    print(f"queue {q}: {len(iv)} kernels, busy {busy / 1e6:.1f} ms")
    tot, cnt = collections.Counter(), collections.Counter()
    for s, e, n in iv:
        tot[n] += e - s
        cnt[n] += 1
    for k, v in tot.most_common(10):
        print(f"    {k:46} {cnt[k]:6d} {v / 1e6:9.1f} ms  avg {v / cnt[k] / 1e3:8.1f} us")
