#!/usr/bin/env python
"""`dmm_dirty_run_multi` alone at cfg-3 tile sizes: D days per read of a resident slab (8 frequencies' tiles, 51 GB),
HIP-event time per launch for the variants behind the "dirty_variant" / "grid_mult" options.

    python tools/multi_tune.py > gpurun_out/multi_tune.json
"""
import ctypes as C
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
    from draco_amd.analysis._solve import Slab
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    cfg = wl.CONFIGS[3]
    nf, lmax = 8, cfg["lmax"]
    tel = TransitTelescope(wl.frequencies(nf), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    bt = SyntheticProvider(tel, seed=3003)
    ms = np.tile(np.arange(lmax + 1, dtype=np.int32), nf)
    fs = np.repeat(np.arange(nf, dtype=np.int32), lmax + 1)
    c64 = len(sys.argv) > 1 and sys.argv[1] == "complex64"
    slab = Slab(ctx, bt, ms, fs, fs, _lib.DMM_C64 if c64 else _lib.DMM_C128, _lib.DMM_B_PACKED, nf, lmax + 1)
    gen = torch.Generator(device=ctx.device).manual_seed(3)
    shape = (lmax + 1, 2, nf, tel.npairs)
    Dmax = 8
    mv = [torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen) for _ in range(Dmax)]
    mw = [torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) for _ in range(Dmax)]
    al = [torch.empty((nf, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device) for _ in range(Dmax)]
    out = {"b_GB": slab.b_bytes / 1e9, "rows": []}

    def run(D, reps=5):
        PA = C.c_void_p * D
        pv, pw, pa = PA(*[ptr(x) for x in mv[:D]]), PA(*[ptr(x) for x in mw[:D]]), PA(*[ptr(x) for x in al[:D]])
        _lib.check(_lib.lib.dmm_dirty_run_multi(slab.plan, ptr(slab.pool), pv, pw, pa, D))
        ctx.sync()
        best = 1e9
        for _ in range(3):
            ctx.timer_start()
            for _ in range(reps):
                _lib.check(_lib.lib.dmm_dirty_run_multi(slab.plan, ptr(slab.pool), pv, pw, pa, D))
            best = min(best, ctx.timer_stop() / reps)
        return best

    for variant in ((0,) if c64 else (0, 1, 2, 3, 4)):
        for gm in ((1,) if c64 else (1, 2)):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"dirty_variant", variant))
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"grid_mult", gm))
            for D in (1, 2, 4, 8):
                if variant and D == 1:
                    continue
                if gm == 2 and D == 8:
                    continue  # (two blocks of 97 KB do not fit a CU's LDS)
                ms_ = run(D)
                row = {"variant": variant, "grid_mult": gm, "D": D, "ms": ms_, "ms_per_day": ms_ / D, "hbm_frac": slab.b_bytes / ms_ / 1e6 / 8000.0,
                       "f64_frac": 8.0 * D * slab.b_bytes / (8 if c64 else 16) / ms_ / 1e9 / 78.6}
                out["rows"].append(row)
                print(json.dumps(row), file=sys.stderr, flush=True)
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"dirty_variant", 0))
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"grid_mult", 0))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
