#!/usr/bin/env python
"""Timeline digest of an ML run traced with `rocprofv3 --kernel-trace --output-format csv`: where the serial QL
launches (k_td_solve, second stream) sit relative to the rest of the pass.

    python tools/ml_trace.py <rocprof output dir>
"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows]
# the LAST pass of the run: from the last k_prior-less marker = last k_dirty<..., false, ...> (the sky-side right-hand sides) or first k_nt<0>
starts = [i for i, e in enumerate(ev) if "k_rowsum" in e[0]]
first = starts[len(starts) // 2] if starts else 0  # second half of the run = the timed pass
t0 = ev[first][1]
end = max(e[2] for e in ev[first:])
print(f"pass length {(end - t0) / 1e6:.1f} ms")
for n, s, e, q in ev[first:]:
    if "k_td_solve" in n:
        print(f"k_td_solve  {(s - t0) / 1e6:9.1f} .. {(e - t0) / 1e6:9.1f} ms  (queue {q})")
busy = {}
for n, s, e, q in ev[first:]:
    busy.setdefault(q, []).append((s, e))
for q, iv in busy.items():
    tot = sum(e - s for s, e in iv)
    print(f"queue {q}: {len(iv)} kernels, busy {tot / 1e6:.1f} ms, first {(min(s for s, _ in iv) - t0) / 1e6:.1f} last {(max(e for _, e in iv) - t0) / 1e6:.1f}")
import collections
tot = collections.Counter()
cnt = collections.Counter()
for n, s, e, q in ev[first:]:
    key = n.split("(")[0][-40:] if "<" not in n else n[n.find("k_"):][:40]
    tot[key] += e - s
    cnt[key] += 1
for k, v in tot.most_common(12):
    print(f"{k:42} {cnt[k]:6d} {v / 1e6:9.1f} ms")
