#!/usr/bin/env python
"""The ring-map makers through their task classes at a CHIME-like shape (external beam and analytic beam).

    python tools/ringmap_tasks.py [mmax 1024] [nfreq 8] [oddra 0]
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

    from draco_amd.analysis.ringmapmaker import TikhonovRingMapMaker, WienerRingMapMakerAnalytical
    from draco_amd.core import containers
    from draco_amd.device import Context

    ctx = Context.get()
    mmax = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nfreq = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    oddra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    nm, new, nel = mmax + 1, 4, 512
    pol = np.array(["XX", "XY", "YX", "YY"])
    freq = np.linspace(400.0, 800.0, nfreq, endpoint=False)
    ew = 22.0 * np.arange(new)
    el = np.linspace(-0.95, 0.95, nel)
    gen = torch.Generator(device=ctx.device).manual_seed(0)
    shp = (nm, 2, 4, nfreq, new, nel)
    kw = dict(mmax=mmax, oddra=bool(oddra), pol=pol, freq=freq, ew=ew, el=el, allocate=False)
    hv = containers.HybridVisMModes(**kw)
    hv.attach("vis", torch.randn(shp, dtype=torch.complex64, device=ctx.device, generator=gen))
    hv.attach("vis_weight", torch.rand(shp[:-1], dtype=torch.float32, device=ctx.device, generator=gen) + 0.5)
    bm = containers.HybridVisMModes(**kw)
    bm.attach("vis", torch.randn(shp, dtype=torch.complex64, device=ctx.device, generator=gen))
    bm.attach("vis_weight", torch.ones(shp[:-1], dtype=torch.float32, device=ctx.device))

    class Tel:  # what io.get_telescope duck-types on
        latitude = 49.32
        lmax = mmax
        frequencies = freq

    Tel.mmax = mmax

    rep = {"mmax": mmax, "nra": 2 * mmax + oddra, "nfreq": nfreq, "nel": nel}

    def timed(name, fn):
        fn()  # tables, allocations
        ctx.sync()
        t0 = time.perf_counter()
        out = fn()
        ctx.sync()
        rep[name + "_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        return out

    t1 = TikhonovRingMapMaker(weight_ew="inverse_variance", window_type="nuttall")
    t1.setup(Tel())
    rm = timed("tikhonov_external_beam", lambda: t1.process(hv, bm))
    assert bool(torch.isfinite(rm.map._dev).all())
    t2 = WienerRingMapMakerAnalytical()
    t2.setup(Tel())
    timed("analytic_beam_mmodes", lambda: t2._get_beam_mmodes(hv))
    rm = timed("wiener_analytic_beam", lambda: t2.process(hv))
    assert bool(torch.isfinite(rm.map._dev).all())
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
