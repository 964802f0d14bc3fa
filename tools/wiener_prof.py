#!/usr/bin/env python
"""Wiener over all m of a few frequencies at a config (for rocprofv3 kernel breakdowns)."""
import os
import sys
import time

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

    cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    kind = sys.argv[3] if len(sys.argv) > 3 else "wiener"
    side = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # 3: telescope-side systems only
    cfg = osyn.CONFIGS[cfgn]
    ctx = Context.get()
    tel = TransitTelescope(osyn.frequencies(nf), lmax=cfg["lmax"], ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    bdt = _lib.DMM_C64 if (len(sys.argv) > 5 and sys.argv[5] == "c64") else _lib.DMM_C128
    eng = SolveEngine(SyntheticProvider(tel, seed=5), ctx, bdt, _lib.DMM_B_PACKED, cache=True)
    gen = torch.Generator(device=ctx.device).manual_seed(7)
    vis = torch.randn((nf, tel.npairs, cfg["nra"]), dtype=torch.complex64, device=ctx.device, generator=gen)
    w = torch.rand((nf, tel.npairs, cfg["nra"]), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
    mv, mw = mmode_forward(ctx, vis, w, cfg["lmax"])
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", side))
    for _ in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        eng.solve(kind, mv, mw, list(range(nf)), cfg["lmax"], prior_amp=1.0, prior_tilt=0.5)
        ctx.sync()
        dt = time.perf_counter() - t0
    print(f"{kind} cfg{cfgn} nf={nf}: {dt*1e3:.1f} ms, {dt*1e3/(nf*(cfg['lmax']+1)):.4f} ms per solve")


if __name__ == "__main__":
    main()
