#!/usr/bin/env python
"""Timing of the forward projection v = B a (k_project) at a config: HIP events, algorithmic GB/s."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context
    from draco_amd import workloads as osyn

    cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfg = osyn.CONFIGS[cfgn]
    ctx = Context.get()
    lmax = cfg["lmax"]
    tel = TransitTelescope(osyn.frequencies(nf), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    eng = SolveEngine(SyntheticProvider(tel, seed=5), ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, cache=True)
    gen = torch.Generator(device=ctx.device).manual_seed(7)
    alm = torch.randn((nf, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device, generator=gen)
    eng.project(alm, list(range(nf)), lmax)
    ctx.sync()
    for var, gm in [(v, g) for v in (0, 5) for g in (1, 2, 3)]:  # 4 = the row-per-wave form shipped before; 0 = row groups of 8
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"project_grid_mult", gm))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"project_variant", var))
        ts = []
        for _ in range(4):
            ctx.timer_start()
            eng.project(alm, list(range(nf)), lmax)
            ts.append(ctx.timer_stop())
        t = float(np.median(ts[1:]))
        print(json.dumps({"config": cfgn, "nfreq": nf, "variant": var, "grid_mult": gm, "ms": t, "B_GB": eng.last_b_bytes / 1e9, "TBs": eng.last_b_bytes / t / 1e9}))


if __name__ == "__main__":
    main()
