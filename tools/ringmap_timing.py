#!/usr/bin/env python
"""Timing of the deconvolving ring-map maker at a CHIME-like shape (HIP events)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from draco_amd import _lib
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    mmax = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nfreq = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    oddra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    nm, npol, new, nel = mmax + 1, 4, 4, 512
    nra = 2 * mmax + oddra
    gen = torch.Generator(device=ctx.device).manual_seed(0)
    shp = (nm, 2, npol, nfreq, new, nel)
    hv = torch.randn(shp, dtype=torch.complex64, device=ctx.device, generator=gen)
    bv = torch.randn(shp, dtype=torch.complex64, device=ctx.device, generator=gen)
    hw = torch.rand(shp[:-1], dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
    table = ctx.to_device(np.ones(new), np.float64)
    eps = ctx.to_device(np.full((nfreq, nm), 1e-3), np.float64)
    rmap = ctx.empty((1, npol, nfreq, nra, nel), np.float64)
    rwgt = ctx.empty((npol, nfreq, nra, nel), np.float64)
    rdbp = ctx.empty((1, npol, nfreq, nel), np.float64)

    def run():
        _lib.check(_lib.lib.dmm_ringmap_deconvolve(ctx.handle, nm, nm, npol, nfreq, new, nel, nra, 2, 0, 0, ptr(hv), ptr(hw), ptr(bv), ptr(table), ptr(eps), None, ptr(rmap), ptr(rwgt), ptr(rdbp), None))

    variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # 1: the three-kernel form even where the single pass applies; 2: the single pass with 8 elevations per block (round 3)
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ringmap_variant", variant))
    run()
    ctx.sync()
    ts = []
    for _ in range(3):
        ctx.timer_start()
        run()
        ts.append(ctx.timer_stop())
    t = float(np.median(ts))
    b_in = hv.numel() * 8 * 2 + hw.numel() * 4
    b_out = (rmap.numel() + rwgt.numel()) * 8
    print(json.dumps({"variant": {0: "single pass, 16 elevations per block (default where it applies)", 1: "three kernels", 2: "single pass, 8 elevations per block (round 3)"}.get(variant, str(variant)), "mmax": mmax, "nra": nra, "nfreq": nfreq, "nel": nel, "ms": t, "ms_per_freq": t / nfreq,
                      "algorithmic_GB": (b_in + b_out) / 1e9, "GBs": (b_in + b_out) / t / 1e6}))


if __name__ == "__main__":
    main()
