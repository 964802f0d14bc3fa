#!/usr/bin/env python
"""ML (blocked Jacobi) sweep-cap tuning: time and accuracy vs the oracle SVD on sampled tiles."""
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
    from oracle import mapmaker as omm
    from oracle import synth as osyn

    cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cfg = osyn.CONFIGS[cfgn]
    ctx = Context.get()
    lmax = cfg["lmax"]
    tel = TransitTelescope(osyn.frequencies(1), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    eng = SolveEngine(SyntheticProvider(tel, seed=5), ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, cache=True)
    gen = torch.Generator(device=ctx.device).manual_seed(7)
    vis = torch.randn((1, tel.npairs, cfg["nra"]), dtype=torch.complex64, device=ctx.device, generator=gen)
    w = torch.rand((1, tel.npairs, cfg["nra"]), dtype=torch.float32, device=ctx.device, generator=gen) * 40 + 10
    mv, mw = mmode_forward(ctx, vis, w, lmax)
    mvh, mwh = mv.cpu().numpy(), mw.cpu().numpy()
    ms = [0, lmax // 3, (2 * lmax) // 3, lmax - 2]
    refs = {m: omm.ml_solve(osyn.beam_tile(5, m, 0, tel.npairs, 4, lmax), mvh[m, :, 0], mwh[m, :, 0]) for m in ms}
    eng.solve("ml", mv, mw, [0], lmax, acond=1e-4, rcond=1e-3)
    for inner, outer in ((1, 60), (2, 60)):
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_inner_sweeps", inner))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_outer_sweeps", outer))
        ctx.sync()
        t0 = time.perf_counter()
        alm = eng.solve("ml", mv, mw, [0], lmax, acond=1e-4, rcond=1e-3)
        ctx.sync()
        dt = time.perf_counter() - t0
        err = max(np.abs(alm[0, :, m, :].cpu().numpy() - refs[m]).max() / np.abs(refs[m]).max() for m in ms)
        print(json.dumps({"cfg": cfgn, "inner": inner, "outer": outer, "ms_per_tile": dt * 1e3 / (lmax + 1), "max_rel_err": err}))


if __name__ == "__main__":
    main()
