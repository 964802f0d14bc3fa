#!/usr/bin/env python
"""Busy time and gaps per queue of a `rocprofv3 --kernel-trace --output-format csv` run (second half of the run = the
timed repetition of the tools/*_prof.py scripts).

    python tools/trace_gaps.py <rocprof output dir>
"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows]
mid = (ev[0][1] + ev[-1][2]) // 2
ev = [e for e in ev if e[1] >= mid] if len(sys.argv) < 3 else ev
t0, t1 = ev[0][1], max(e[2] for e in ev)
print(f"window {(t1 - t0) / 1e6:.1f} ms, {len(ev)} kernels")
byq = collections.defaultdict(list)
for n, s, e, q in ev:
    byq[q].append((s, e, n))
for q, iv in byq.items():
    iv.sort()
    busy = sum(e - s for s, e, _ in iv)
    gaps = [(iv[i + 1][0] - iv[i][1], iv[i][2], iv[i + 1][2]) for i in range(len(iv) - 1) if iv[i + 1][0] > iv[i][1]]
    big = sorted(gaps, reverse=True)[:6]
    print(f"queue {q}: {len(iv)} kernels, busy {busy / 1e6:.1f} ms, gaps total {sum(g for g, _, _ in gaps) / 1e6:.1f} ms, gaps > 20 us: {sum(1 for g, _, _ in gaps if g > 20000)}")
    for g, a, b in big:
        short = lambda x: x[x.find("k_"):][:28] if "k_" in x else x[:28]
        print(f"    {g / 1e3:8.1f} us between {short(a)} and {short(b)}")
