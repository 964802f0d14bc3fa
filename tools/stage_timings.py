#!/usr/bin/env python
"""Per-stage timings on one MI355X (HIP events on the launch stream) for DESIGN.md.

    python tools/stage_timings.py [--config 3] > gpurun_out/stage_timings.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--dense-freqs", type=int, default=1, help="frequencies used for the Wiener sample")
    ap.add_argument("--ml-tiles", type=int, default=16)
    ap.add_argument("--ml-batched", action="store_true")
    args = ap.parse_args()
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import Slab, SolveEngine
    from draco_amd.analysis.transform import mmode_forward, mmode_inverse
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context, ptr
    from draco_amd import workloads as osyn

    cfg = osyn.CONFIGS[args.config]
    ctx = Context.get()
    nfreq, nra, lmax, nside = cfg["nfreq"], cfg["nra"], cfg["lmax"], cfg["nside"]
    tel = TransitTelescope(osyn.frequencies(nfreq), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    npairs = tel.npairs
    bt = SyntheticProvider(tel, seed=1)
    out = {"config": args.config, "npairs": npairs, "nfreq": nfreq, "nra": nra, "lmax": lmax, "nside": nside}

    def timed(fn, reps=3):
        fn()
        ctx.sync()
        ts = []
        for _ in range(reps):
            ctx.timer_start()
            fn()
            ts.append(ctx.timer_stop())
        return float(np.median(ts))

    gen = torch.Generator(device=ctx.device).manual_seed(0)
    vis = torch.randn((nfreq, npairs, nra), dtype=torch.complex64, device=ctx.device, generator=gen)
    w = torch.rand((nfreq, npairs, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
    hold = {}

    def fwd():
        hold["mv"], hold["mw"] = mmode_forward(ctx, vis, w, lmax)

    t = timed(fwd)
    b_in = vis.numel() * 8 + w.numel() * 4
    b_out = (lmax + 1) * 2 * nfreq * npairs * 24
    out["mmode_forward"] = {"ms": t, "GBs": (b_in + b_out) / t / 1e6, "bytes": b_in + b_out}
    mv, mw = hold["mv"], hold["mw"]
    t = timed(lambda: mmode_inverse(ctx, mv, nra))
    out["mmode_inverse"] = {"ms": t, "GBs": ((lmax + 1) * 2 * nfreq * npairs * 16 + vis.numel() * 8) / t / 1e6}

    # odd length (what SimulateSidereal produces): Bluestein path
    nodd = 2 * lmax + 1
    vis_o = torch.randn((max(nfreq // 8, 1), npairs, nodd), dtype=torch.complex64, device=ctx.device, generator=gen)
    t = timed(lambda: mmode_forward(ctx, vis_o, None, lmax))
    out["mmode_forward_odd"] = {"ms": t, "nra": nodd, "rows": vis_o.shape[0] * npairs, "GBs": (vis_o.numel() * 8 + (lmax + 1) * 2 * vis_o.shape[0] * npairs * 16) / t / 1e6}

    # SHT on a few frequencies
    nf_s = min(nfreq, 8)
    alm = torch.randn((nf_s, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device, generator=gen)
    maps = ctx.empty((nf_s, 4, 12 * nside * nside), np.float64)
    t = timed(lambda: _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(alm), nf_s, 4, lmax, lmax, nside, ptr(maps))))
    out["alm2map"] = {"ms_per_freq": t / nf_s, "nfreq_timed": nf_s}
    alm2 = ctx.empty((nf_s, 4, lmax + 1, lmax + 1), np.complex128)
    t = timed(lambda: _lib.check(_lib.lib.dmm_map2alm(ctx.handle, ptr(maps), nf_s, 4, lmax, lmax, nside, 0, ptr(alm2))), reps=2)
    out["map2alm_iter0"] = {"ms_per_freq": t / nf_s, "nfreq_timed": nf_s}
    del alm, maps, alm2

    # Wiener over all m of `dense_freqs` frequencies
    nf_d = args.dense_freqs
    eng = SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, cache=True)
    mv_d = mv[:, :, :nf_d].contiguous()
    mw_d = mw[:, :, :nf_d].contiguous()
    t0 = time.perf_counter()
    eng.solve("wiener", mv_d, mw_d, list(range(nf_d)), lmax, prior_amp=1.0, prior_tilt=0.5)
    ctx.sync()
    t1 = time.perf_counter()
    eng.solve("wiener", mv_d, mw_d, list(range(nf_d)), lmax, prior_amp=1.0, prior_tilt=0.5)
    ctx.sync()
    t = (time.perf_counter() - t1) * 1e3
    ntile = nf_d * (lmax + 1)
    ntel = 2 * npairs
    flops = sum(8.0 * ntel * ntel * 4 * (lmax + 1 - m) / 2 for m in range(lmax + 1)) * nf_d + ntile * (8.0 / 3.0) * ntel**3
    out["wiener"] = {"ms": t, "tiles": ntile, "ms_per_tile": t / ntile, "TFLOPs": flops / t / 1e9, "first_call_ms": (t1 - t0) * 1e3}

    # ML batched over all m of one frequency
    if args.ml_batched:
        mv1 = mv[:, :, :1].contiguous()
        mw1 = mw[:, :, :1].contiguous()
        t1 = time.perf_counter()
        eng.solve("ml", mv1, mw1, [0], lmax, acond=1e-4, rcond=1e-3)
        ctx.sync()
        t = (time.perf_counter() - t1) * 1e3
        out["ml_batched"] = {"ms": t, "tiles": lmax + 1, "ms_per_tile": t / (lmax + 1)}

    # ML on a handful of tiles (Jacobi is O(n^3 * sweeps) from global memory)
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker

    if args.ml_tiles <= 0:
        ml_skip = True
    else:
        ml_skip = False
    ml = MaximumLikelihoodMapMaker()
    ml.setup(bt)
    if not ml_skip:
        ms = np.linspace(0, lmax, args.ml_tiles).astype(int)
        v = mv[:, :, 0].cpu().numpy()
        Ni = mw[:, :, 0].cpu().numpy()
        ml._solve_m(int(ms[0]), 0, v[ms[0]], Ni[ms[0]])
        t0 = time.perf_counter()
        for m in ms[: max(2, args.ml_tiles // 4)]:
            ml._solve_m(int(m), 0, v[m], Ni[m])
        out["ml_single_tile_ms"] = (time.perf_counter() - t0) * 1e3 / max(2, args.ml_tiles // 4)

    # PCIe
    h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    d = torch.empty(1 << 30, dtype=torch.uint8, device=ctx.device)
    t = timed(lambda: d.copy_(h, non_blocking=True))
    out["h2d_GBs_pinned"] = (1 << 30) / t / 1e6
    t = timed(lambda: h.copy_(d, non_blocking=True))
    out["d2h_GBs_pinned"] = (1 << 30) / t / 1e6
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
