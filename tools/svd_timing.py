#!/usr/bin/env python
"""Timing of the m-mode SVD filter at a config's size (all m of one rank's MModes).

    python tools/svd_timing.py [--config 3] [--mask 0.02]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--mask", type=float, default=0.0, help="fraction of missing (zero-weight) entries")
    a = ap.parse_args()
    from draco_amd.analysis.svdfilter import _decompose
    from draco_amd.core.products import TransitTelescope
    from draco_amd.device import Context
    from draco_amd import workloads as osyn

    cfg = osyn.CONFIGS[a.config]
    ctx = Context.get()
    tel = TransitTelescope(osyn.frequencies(cfg["nfreq"]), lmax=cfg["lmax"], ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    n_m, nfreq, nbase = cfg["lmax"] + 1, cfg["nfreq"], tel.npairs
    gen = torch.Generator(device=ctx.device).manual_seed(3)
    smooth = torch.cos(0.02 * torch.arange(nfreq, device=ctx.device, dtype=torch.float64))[None, None, :, None]
    vis = 1e3 * torch.randn((n_m, 2, 1, nbase), dtype=torch.complex128, device=ctx.device, generator=gen) * smooth
    vis = vis + torch.randn((n_m, 2, nfreq, nbase), dtype=torch.complex128, device=ctx.device, generator=gen)
    w = torch.ones(vis.shape, dtype=torch.float64, device=ctx.device)
    if a.mask > 0:
        w[torch.rand(w.shape, device=ctx.device, generator=gen) < a.mask] = 0.0
    out = {"config": a.config, "n_m": n_m, "nfreq": nfreq, "columns": 2 * nbase, "mask": a.mask}
    for label, mode in (("spectrum", 0), ("filter", 1)):
        for rep in range(2):
            work = vis.clone()
            ctx.sync()
            t0 = time.perf_counter()
            spec = _decompose(ctx, work, w, 5, 5, mode, float(1e3 * nfreq), 1e-3, 1e-2)
            ctx.sync()
            dt = time.perf_counter() - t0
        out[label + "_ms"] = dt * 1e3
        out[label + "_ms_per_m"] = dt * 1e3 / n_m
    out["bytes_mmodes"] = vis.numel() * 16
    print(json.dumps(out))


if __name__ == "__main__":
    main()
