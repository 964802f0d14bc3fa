#!/usr/bin/env python
"""Interleaved A/B of k_dirty tuning variants in ONE process (guide rule 24): median/min ms and TB/s."""
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
    from draco_amd.device import ptr
    from draco_amd import workloads as osyn

    cfg = osyn.CONFIGS[3]
    dtype = sys.argv[1] if len(sys.argv) > 1 else "complex128"
    job = bench.Job(cfg, 0, dtype, 16)
    ctx = job.ctx
    from draco_amd.analysis.transform import mmode_forward

    mv, mw = mmode_forward(ctx, job.vis, job.weight, job.lmax)
    lib = _lib.lib

    def run():
        _lib.check(lib.dmm_dirty_run(job.slab.plan, ptr(job.slab.pool), mv.data_ptr(), mw.data_ptr(), job.alm.data_ptr()))

    variants = [(v, g, st) for v in (0, 2, 3, 4) for g in (1, 2, 3) for st in (0, 1)]  # st = 1: static striding
    times = {k: [] for k in variants}
    for rnd in range(6):
        for v, g, st in variants:
            _lib.check(lib.dmm_ctx_set_option(ctx.handle, b"dirty_variant", v))
            _lib.check(lib.dmm_ctx_set_option(ctx.handle, b"grid_mult", g))
            _lib.check(lib.dmm_ctx_set_option(ctx.handle, b"dirty_static", st))
            if rnd == 0:
                run()
                ctx.sync()
            ctx.timer_start()
            run()
            times[(v, g, st)].append(ctx.timer_stop())
    res = []
    for (v, g, st), ts in times.items():
        ts = np.array(ts[1:])
        res.append({"variant": v, "grid_mult": g, "static": st, "median_ms": float(np.median(ts)), "min_ms": float(ts.min()),
                    "TBs_median": job.dirty_bytes / np.median(ts) / 1e9})
    res.sort(key=lambda r: r["median_ms"])
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
