#!/usr/bin/env python
"""A/B of k_dirty variants INSIDE the bench step (alm2map on the side stream) and alone, interleaved in one process.

    python tools/step_ab.py [complex128|complex64] [variants=0,7] [rounds=5] [option=dirty_variant]

    `option` is any dmm_ctx_set_option name (dirty_variant, grid_mult ...); `variants` its values.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import bench
    from draco_amd import _lib
    from draco_amd import workloads as wl

    dtype = sys.argv[1] if len(sys.argv) > 1 else "complex128"
    variants = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,7").split(",")]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    option = (sys.argv[4] if len(sys.argv) > 4 else "dirty_variant").encode()
    job = bench.Job(wl.CONFIGS[3], 0, 1, "weak", dtype, 0)
    job.step()
    torch.cuda.synchronize()
    import time

    res = {v: {"step_ms": [], "launch_in_step_ms": [], "launch_alone_ms": []} for v in variants}
    for _ in range(rounds):
        for v in variants:
            _lib.check(_lib.lib.dmm_ctx_set_option(job.ctx.handle, option, v))
            ms, n = job.timed_launches(job.to_alm)
            res[v]["launch_alone_ms"].append(ms)
            eng = job.dm._get_engine()
            eng.launch_events = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            job.step()
            torch.cuda.synchronize()
            res[v]["step_ms"].append((time.perf_counter() - t0) * 1e3)
            res[v]["launch_in_step_ms"].append(float(np.mean([a.elapsed_time(b) for a, b, _, _ in eng.launch_events])))
            eng.launch_events = None
    for v in variants:
        r = {k: float(np.median(x)) for k, x in res[v].items()}
        r["variant"] = v
        r["alone_TBs"] = job.dirty_bytes / r["launch_alone_ms"] / 1e9
        r["in_step_TBs"] = job.dirty_bytes / r["launch_in_step_ms"] / 1e9
        print(json.dumps(r))


if __name__ == "__main__":
    main()
