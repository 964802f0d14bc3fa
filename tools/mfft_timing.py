#!/usr/bin/env python
"""The forward m-mode transform at a configuration's size, kernel by kernel (HIP events on the library's stream):
`dmm_mfft_pack` (FFT + +/-m pack), `dmm_mmode_weight` (weight reduction + broadcast), and the two together.

    python tools/mfft_timing.py [config] > gpurun_out/mfft_timing.json
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
    from draco_amd.analysis.transform import mmode_forward
    from draco_amd.core.products import TransitTelescope
    from draco_amd.device import Context, ptr

    cfg = wl.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 3]
    ctx = Context.get()
    nfreq, nra, lmax = cfg["nfreq"], cfg["nra"], cfg["lmax"]
    tel = TransitTelescope(wl.frequencies(nfreq), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    nrow = nfreq * tel.npairs
    gen = torch.Generator(device=ctx.device).manual_seed(1)
    vis = torch.randn((nfreq, tel.npairs, nra), dtype=torch.complex64, device=ctx.device, generator=gen)
    w = torch.rand((nfreq, tel.npairs, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
    mv = ctx.empty((lmax + 1, 2, nfreq, tel.npairs), np.complex128)
    mw = ctx.empty((lmax + 1, 2, nfreq, tel.npairs), np.float64)

    def fft():
        _lib.check(_lib.lib.dmm_mfft_pack(ctx.handle, ptr(vis), nrow, nra, ptr(mv), lmax, _lib.DMM_C128, None))

    def wgt():
        _lib.check(_lib.lib.dmm_mmode_weight(ctx.handle, ptr(w), nrow, nra, ptr(mw), lmax, None))

    def both():
        fft()
        wgt()

    def timed(fn, reps=20):
        fn()
        ctx.sync()
        best = []
        for _ in range(3):
            ctx.timer_start()
            for _ in range(reps):
                fn()
            best.append(ctx.timer_stop() / reps)
        return min(best)

    b_fft = nrow * nra * 8 + (lmax + 1) * 2 * nrow * 16
    b_w = nrow * nra * 4 + (lmax + 1) * 2 * nrow * 8
    out = {"config": cfg.get("name", ""), "rows": nrow, "nra": nra, "mmax": lmax}
    for name, fn, by in (("mfft_pack", fft, b_fft), ("mmode_weight", wgt, b_w), ("both", both, b_fft + b_w)):
        ms = timed(fn)
        out[name] = {"ms": ms, "bytes": by, "GBs": by / ms / 1e6, "frac_of_8TBs": by / ms / 1e6 / 8000.0}
    a, b = mmode_forward(ctx, vis, w, lmax)
    ctx.sync()
    out["same_as_task_path"] = bool(torch.equal(a, mv) and torch.equal(b, mw))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
