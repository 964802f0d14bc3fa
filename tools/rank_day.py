#!/usr/bin/env python
"""ONE rank's share of the strong-scaled headline job, timed alone on one GPU: what a rank of `bench.py --gpus N` does per
sidereal day when cfg 3's 256 frequencies are split over N = 8 / 4 / 2 ranks (no collective inside the timed region, so
the N-GPU figure is this one if every rank behaves alike).  Not a scaling measurement -- the projection DESIGN 7 quotes.

    python tools/rank_day.py > gpurun_out/rank_day.json
"""
import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, numpy as np
import bench
from draco_amd import workloads as wl
torch.cuda.set_device(0)
cfg = wl.CONFIGS[3]
res = {}
for world in (8, 4, 2):
    job = bench.Job(cfg, 0, world, "strong", "complex128", 0)
    for _ in range(3): job.step()
    torch.cuda.synchronize()
    K = 40
    t0 = time.perf_counter(); issue = []
    for _ in range(K):
        m = job.step(); issue.append(time.perf_counter())
    m.map._dev; torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    # host-only cost: time to issue when GPU is far behind? measure issue gaps of first few days
    gaps = np.diff([t0] + issue)
    alone, n = job.timed_launches(job.to_alm)
    res[world] = {"nfreq_rank": job.nfreq, "ms_per_day": el * 1e3, "value_if_all_ranks_alike": 513 / el, "dirty_alone_ms": alone, "launches": n, "host_issue_ms_median": float(np.median(gaps) * 1e3), "host_issue_ms_first": [round(g * 1e3, 2) for g in gaps[:4]]}
    print(world, json.dumps(res[world]), flush=True)
    del job, m
    from draco_amd.analysis import _solve
    _solve.release_pools(); torch.cuda.empty_cache()
