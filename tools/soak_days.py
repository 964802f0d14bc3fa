#!/usr/bin/env python
"""Six cfg-3 days issued back to back through the bench's task objects, no synchronisation and nothing reading a map until
the end: all six maps must be bit-identical (the deferred stream wait of the map, recycled allocations).

    python tools/soak_days.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from draco_amd import workloads as wl
job = bench.Job(wl.CONFIGS[3], 0, 1, "weak", "complex128", 0)
first = job.step()
maps = [first]
for _ in range(5):
    maps.append(job.step())   # no synchronisation, nothing reads
a = first.map._dev
ok = all(bool(torch.equal(a, m.map._dev)) for m in maps[1:])
torch.cuda.synchronize()
print("soak cfg3: 6 back-to-back days, maps identical:", ok, "finite:", bool(torch.isfinite(a).all()))
