#!/usr/bin/env python
"""Interleaved A/B of k_dirty tuning variants in ONE process (guide rule 24): median/min ms and TB/s per launch.

    python tools/tune_dirty.py [complex128|complex64] [pool_freqs=16] [variants=0,2,3,4] [grid_mults=1,2]

One launch = all 513 m of `pool_freqs` cfg-3 frequencies through DirtyMapMaker.make_alm (HIP events on the launch stream).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from draco_amd import _lib
    from draco_amd import workloads as wl
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context

    dtype = sys.argv[1] if len(sys.argv) > 1 else "complex128"
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    variants = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,2,3,4").split(",")]
    gms = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,2").split(",")]
    cfg = wl.CONFIGS[3]
    lmax = cfg["lmax"]
    ctx = Context.get()
    tel = TransitTelescope(wl.frequencies(nf), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    gen = torch.Generator(device=ctx.device).manual_seed(5)
    shape = (lmax + 1, 2, nf, tel.npairs)
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
    mm.attach("vis", torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen))
    mm.attach("vis_weight", torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5)
    es = 16 if dtype == "complex128" else 8
    per_freq = sum(2 * tel.npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * es
    dm = DirtyMapMaker(b_dtype=dtype, pool_bytes=nf * per_freq + (1 << 20))
    dm.setup(SyntheticProvider(tel, seed=3003))
    dm.make_alm(mm)
    eng = dm._get_engine()
    nbytes = nf * per_freq + nf * (lmax + 1) * 2 * tel.npairs * 24 + nf * sum(4 * (lmax + 1 - m) * 16 for m in range(lmax + 1))
    lib = _lib.lib
    combos = [(v, g, st) for v in variants for g in gms for st in (0,)]
    times = {k: [] for k in combos}
    for rnd in range(7):
        for v, g, st in combos:
            _lib.check(lib.dmm_ctx_set_option(ctx.handle, b"dirty_variant", v))
            _lib.check(lib.dmm_ctx_set_option(ctx.handle, b"grid_mult", g))
            _lib.check(lib.dmm_ctx_set_option(ctx.handle, b"dirty_static", st))
            eng.launch_events = []
            dm.make_alm(mm)
            torch.cuda.synchronize()
            times[(v, g, st)].append(sum(a.elapsed_time(b) for a, b, _, _ in eng.launch_events))
            eng.launch_events = None
    res = []
    for (v, g, st), ts in times.items():
        ts = np.array(ts[1:])
        res.append({"dtype": dtype, "variant": v, "grid_mult": g, "static": st, "median_ms": float(np.median(ts)), "min_ms": float(ts.min()),
                    "TBs_median": nbytes / np.median(ts) / 1e9, "frac_of_8TBs": nbytes / np.median(ts) / 1e9 / 8.0})
    res.sort(key=lambda r: r["median_ms"])
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
