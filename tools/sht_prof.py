#!/usr/bin/env python
"""Run the spherical-harmonic transforms alone (for rocprofv3 --kernel-trace --stats).

    python tools/sht_prof.py [--nside 256 --lmax 512 --nfreq 8 --niter 3 --reps 3]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from draco_amd import _lib
from draco_amd.device import Context, ptr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nside", type=int, default=256)
    ap.add_argument("--lmax", type=int, default=512)
    ap.add_argument("--nfreq", type=int, default=8)
    ap.add_argument("--npol", type=int, default=4)
    ap.add_argument("--niter", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variant", type=int, default=0, help="dmm_ctx_set_option sht_variant")
    a = ap.parse_args()
    ctx = Context.get()
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"sht_variant", int(a.variant)))
    gen = torch.Generator(device=ctx.device).manual_seed(5)
    lmax, nside, nf, npol = a.lmax, a.nside, a.nfreq, a.npol
    alm = torch.randn((nf, npol, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device, generator=gen)
    maps = ctx.empty((nf, npol, 12 * nside * nside), np.float64)
    alm2 = ctx.empty((nf, npol, lmax + 1, lmax + 1), np.complex128)

    def timed(fn):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / a.reps * 1e3

    out = {"variant": a.variant, "nside": nside, "lmax": lmax, "nfreq": nf, "npol": npol}
    t = timed(lambda: _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(alm), nf, npol, lmax, lmax, nside, ptr(maps))))
    out["alm2map_ms_per_freq"] = t / nf
    for it in sorted({0, a.niter}):
        t = timed(lambda: _lib.check(_lib.lib.dmm_map2alm(ctx.handle, ptr(maps), nf, npol, lmax, lmax, nside, it, ptr(alm2))))
        out[f"map2alm_iter{it}_ms_per_freq"] = t / nf
    print(json.dumps(out))


if __name__ == "__main__":
    main()
