#!/usr/bin/env python
"""ML timing over all m of a few frequencies (accuracy against the oracle's SVD is the tests' job: tests/test_gpu_dense.py).

    python tools/ml_tune.py [config [nfreq [modes [ml_eigen]]]]
    modes (ml_shortcut): 0 certified shortcut (default), 2 eigen path always, 3 telescope side only;
    ml_eigen: 0 by batch size (default), 4 tridiagonalisation + QL, 1 blocked Jacobi;
    a fifth argument `ill` spreads the noise weights over eight decades, so that no tile passes the certificate
    (the situation with real, ill-conditioned beam transfers) while the default options stay in force
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.analysis.transform import mmode_forward
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context
    from draco_amd import workloads as osyn

    cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    modes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 3, 2]
    eig = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    cfg = osyn.CONFIGS[cfgn]
    ctx = Context.get()
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", eig))
    lmax = cfg["lmax"]
    tel = TransitTelescope(osyn.frequencies(nf), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    eng = SolveEngine(SyntheticProvider(tel, seed=5), ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, cache=True)
    gen = torch.Generator(device=ctx.device).manual_seed(7)
    vis = torch.randn((nf, tel.npairs, cfg["nra"]), dtype=torch.complex64, device=ctx.device, generator=gen)
    w = torch.rand((nf, tel.npairs, cfg["nra"]), dtype=torch.float32, device=ctx.device, generator=gen) * 40 + 10
    if len(sys.argv) > 5 and sys.argv[5] == "ill":  # one weight per baseline, 1 ... 1e-8
        w = w * torch.pow(10.0, -8.0 * torch.rand((nf, tel.npairs, 1), dtype=torch.float32, device=ctx.device, generator=gen))
    mv, mw = mmode_forward(ctx, vis, w, lmax)
    fl = list(range(nf))
    eng.solve("ml", mv, mw, fl, lmax, acond=1e-4, rcond=1e-3)
    import ctypes as C

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    labels = {0: "shortcut", 3: "telescope-side only", 2: "eigen always"}
    for mode in modes:
        label = labels[mode]
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", mode))
        d0, e0 = counter(b"ml_tiles_direct"), counter(b"ml_tiles_eigen")
        ctx.sync()
        t0 = time.perf_counter()
        alm = eng.solve("ml", mv, mw, fl, lmax, acond=1e-4, rcond=1e-3)
        ctx.sync()
        dt = time.perf_counter() - t0
        print(json.dumps({"cfg": cfgn, "mode": label, "ml_eigen": eig, "nfreq": nf, "ms_per_tile": dt * 1e3 / (nf * (lmax + 1)), "total_s": dt,
                          "tiles_direct": counter(b"ml_tiles_direct") - d0, "tiles_eigen": counter(b"ml_tiles_eigen") - e0}), flush=True)
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))


if __name__ == "__main__":
    main()
