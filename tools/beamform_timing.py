#!/usr/bin/env python
"""Timing of BeamformNS / BeamformEW at a CHIME-like shape (HIP events): 4 pol x 4 EW separations, 511 NS separations
(256 feeds per cylinder), 2048 RA samples, 512 elevations.

    python tools/beamform_timing.py [nfreq=4] [nra=2048]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    nfreq = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    nra = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    npol, nx, ny, npix = 4, 4, 511, 512
    gen = torch.Generator(device=ctx.device).manual_seed(0)
    gv = torch.randn((npol, nfreq, nx, ny, nra), dtype=torch.complex64, device=ctx.device, generator=gen)
    gw = torch.rand((npol, nfreq, nx, ny, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
    red = torch.randint(1, 200, (npol, nx, ny, nra), dtype=torch.int32, device=ctx.device, generator=gen)
    nspos = ctx.to_device(np.fft.fftfreq(ny, d=1.0 / (ny * 0.3048)))
    el = ctx.to_device(np.linspace(-1.0, 1.0, npix))
    iwv = np.ascontiguousarray(np.linspace(400.0, 800.0, nfreq) * 1e6 / 299792458.0)
    hv = ctx.empty((npol, nfreq, nx, npix, nra), np.complex64)
    hw = ctx.empty((npol, nfreq, nx, nra), np.float32)

    def run_ns():
        _lib.check(_lib.lib.dmm_beamform_ns(ctx.handle, npol, nfreq, nx, ny, nra, npix, 1, 0, ptr(gv), ptr(gw), ptr(red), None, ptr(nspos), ptr(el),
                                            C.c_void_p(iwv.ctypes.data), ptr(hv), ptr(hw), None))

    rot = np.eye(4, dtype=np.complex128)
    rot[1, 1:3] = [0.5, 0.5]
    rot[2, 1:3] = [-0.5j, 0.5j]
    rot_d, w_d = ctx.to_device(rot), ctx.to_device(np.array([4.0, 3.0, 2.0, 1.0]) / 10.0)
    nbeam = 2 * nx - 1
    rmm = ctx.empty((nbeam, 4, nfreq, nra, npix), np.float64)
    rmw = ctx.empty((4, nfreq, nra, npix), np.float64)
    rmr = ctx.empty((4, nfreq, nra), np.float64)

    def run_ew():
        _lib.check(_lib.lib.dmm_beamform_ew(ctx.handle, 4, 4, nfreq, nx, npix, nra, 0, ptr(hv), ptr(hw), None, ptr(rot_d), ptr(w_d), ptr(rmm), ptr(rmw), ptr(rmr), None))

    out = {"shape": {"npol": npol, "nfreq": nfreq, "new": nx, "nns": ny, "nra": nra, "npix": npix}}
    for name, fn in (("beamform_ns", run_ns), ("beamform_ew", run_ew)):
        fn()
        ctx.sync()
        ts = []
        for _ in range(3):
            ctx.timer_start()
            fn()
            ts.append(ctx.timer_stop())
        out[name + "_ms_per_freq"] = float(np.median(ts)) / nfreq
    flops = 8.0 * npix * ny * nra * npol * nx  # complex GEMM F [npix x nns] x X [nns x nra] per (pol, ew)
    out["beamform_ns_TFLOPs_f64"] = flops / (out["beamform_ns_ms_per_freq"] * 1e-3) / 1e12
    out["beamform_ns_frac_of_f64_mfma_peak"] = out["beamform_ns_TFLOPs_f64"] / 78.6
    b_ew = npol * nx * npix * nra * 8 + (nbeam + 1) * 4 * nra * npix * 8
    out["beamform_ew_GBs"] = b_ew / (out["beamform_ew_ms_per_freq"] * 1e-3) / 1e9
    print(json.dumps(out))


if __name__ == "__main__":
    main()
