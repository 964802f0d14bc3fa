#!/usr/bin/env python
"""Twenty-five cfg-3 days issued back to back through the bench's task objects, no synchronisation and nothing reading
a map until the end: sampled maps must be bit-identical (the map's own stream wait, recycled allocations), the host
must never get more than `days_in_flight` days ahead of the GPU, and the caching allocator must reach a steady state
(no retry, no hipMalloc / hipFree after the first days).

    python tools/soak_days.py [days]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from draco_amd import workloads as wl

ndays = int(sys.argv[1]) if len(sys.argv) > 1 else 25
job = bench.Job(wl.CONFIGS[3], 0, 1, "weak", "complex128", 0)
first = job.step()
torch.cuda.synchronize()
keep = {0: first}
mem0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
issue = []
for d in range(1, ndays):
    m = job.step()  # no synchronisation, nothing reads
    issue.append(time.perf_counter() - t0)
    if d in (1, ndays // 2, ndays - 1):
        keep[d] = m
    del m
torch.cuda.synchronize()
wall = time.perf_counter() - t0
mem1 = torch.cuda.memory_stats()
a = first.map._dev
ok = all(bool(torch.equal(a, m.map._dev)) for d, m in keep.items() if d)
torch.cuda.synchronize()
ms_day = wall / (ndays - 1) * 1e3
# the host may lead the GPU by at most days_in_flight days: the issue time of day d is >= the GPU's finish of day d - 2
lead = max((d + 1) - issue[d] / (wall / (ndays - 1)) for d in range(len(issue)))
rec = {
    "days": ndays, "ms_per_day": ms_day, "maps_identical": ok, "finite": bool(torch.isfinite(a).all()),
    "max_host_lead_days": lead,
    "num_alloc_retries": mem1["num_alloc_retries"] - mem0["num_alloc_retries"],
    "num_device_alloc": mem1["num_device_alloc"] - mem0["num_device_alloc"],
    "num_device_free": mem1["num_device_free"] - mem0["num_device_free"],
    "reserved_peak_GB": mem1["reserved_bytes.all.peak"] / 1e9,
}
print(json.dumps(rec))
assert ok and rec["num_alloc_retries"] == 0 and rec["num_device_free"] == 0, rec
