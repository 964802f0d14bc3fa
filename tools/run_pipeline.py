#!/usr/bin/env python
"""End-to-end example: sky -> SimulateSidereal -> (noise) -> MModeTransform -> {Dirty, Wiener, ML}MapMaker.

    python tools/run_pipeline.py [--config 1] [--makers dirty wiener ml] [--noise]

Mirrors the reference's tutorial pipeline (doc/pipeline_params.yaml:1-35) with this repo's
task classes.  Prints per-task wall time (device synchronised) and a few sanity numbers.
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
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--makers", nargs="*", default=["dirty", "wiener"])
    ap.add_argument("--noise", action="store_true")
    ap.add_argument("--nfreq", type=int, default=0, help="override the number of frequencies")
    ap.add_argument("--pool-gb", type=float, default=0.0, help="B pool budget in GB (default: 0.6 of the free HBM)")
    args = ap.parse_args()

    import torch

    from draco_amd.analysis.mapmaker import DirtyMapMaker, MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context
    from draco_amd.synthesis.noise import GaussianNoise
    from draco_amd.synthesis.stream import SimulateSidereal
    from draco_amd import workloads as osyn

    cfg = dict(osyn.CONFIGS[args.config])
    if args.nfreq:
        cfg["nfreq"] = args.nfreq
    ctx = Context.get()
    tel = TransitTelescope(osyn.frequencies(cfg["nfreq"]), lmax=cfg["lmax"], ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    bt = SyntheticProvider(tel, seed=3000 + args.config)
    nside = cfg["nside"]

    # band-limited Gaussian sky with C_l = (l+1)^-2, made on the device through alm2map
    from draco_amd import _lib
    from draco_amd.device import ptr

    gen = torch.Generator(device=ctx.device).manual_seed(4000 + args.config)
    lmax = tel.lmax
    alm = torch.randn((tel.nfreq, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device, generator=gen)
    cl = (torch.arange(lmax + 1, device=ctx.device, dtype=torch.float64) + 1.0) ** -1.0
    alm = alm * cl[None, None, None, :]
    alm[:, :, 0, :] = alm[:, :, 0, :].real.to(torch.complex128)  # m = 0 is real
    alm[:, 1:3, :, :2] = 0
    sky = ctx.empty((tel.nfreq, 4, 12 * nside * nside), np.float64)
    alm = alm.contiguous()
    _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(alm), tel.nfreq, 4, lmax, lmax, nside, ptr(sky)))
    mp = containers.Map(nside=nside, freq=tel.frequencies, allocate=False)
    mp.attach("map", sky)

    report = {"config": args.config, "nfeed": tel.nfeed, "npairs": tel.npairs, "nfreq": tel.nfreq, "lmax": lmax, "nside": nside}

    def timed(name, fn):
        torch.cuda.synchronize()  # device-wide: a map-maker's last alm2map may still be running on the side stream
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        report[name + "_s"] = round(time.perf_counter() - t0, 4)
        print(f"[run_pipeline] {name}: {report[name + '_s']} s", file=sys.stderr, flush=True)
        return out

    pool = int(args.pool_gb * 1e9) if args.pool_gb > 0 else None
    sim = SimulateSidereal(pool_bytes=pool)
    sim.setup(bt)
    timed("SimulateSidereal_first_call(incl. B block allocation)", lambda: sim.process(mp))
    ss = timed("SimulateSidereal", lambda: sim.process(mp))
    if args.noise:
        gn = GaussianNoise(seed=1, ndays=733.0, recv_temp=50.0)
        gn.setup(tel)
        ss = timed("GaussianNoise_host", lambda: gn.process(ss))
    tr = MModeTransform()
    tr.setup(bt)
    mm = timed("MModeTransform", lambda: tr.process(ss))
    makers = {"dirty": DirtyMapMaker, "wiener": WienerMapMaker, "ml": MaximumLikelihoodMapMaker}
    for name in args.makers:
        task = makers[name](nside=nside, pool_bytes=pool)
        task.setup(bt)
        timed(name + "_first_call(incl. B fill)", lambda: task.process(mm))
        out = timed(name, lambda: task.process(mm))
        m = out.map[:]
        report[name + "_map_rms"] = float(np.sqrt((m**2).mean()))
        assert np.all(np.isfinite(m))
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
