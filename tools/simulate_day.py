#!/usr/bin/env python
"""One cfg-3 day of `SimulateSidereal.process` (map2alm with the reference's three iterations -> v_m = B_m a_m for every
(m, freq) -> unpack + inverse FFT to the SiderealStream, stream.py:48-178) with B resident under the hbm-pool policy
(32 frequencies' distinct tiles): seconds per day after a warm-up day, and the stages alone.

    python tools/simulate_day.py
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from draco_amd import _lib
    from draco_amd import workloads as wl
    from draco_amd.core import containers
    from draco_amd.core.products import PoolCycledProvider, SyntheticProvider, TransitTelescope
    from draco_amd.device import Context, ptr
    from draco_amd.synthesis.stream import SimulateSidereal

    ctx = Context.get()
    cfg = wl.CONFIGS[3]
    nfreq, lmax, nside = cfg["nfreq"], cfg["lmax"], cfg["nside"]
    tel = TransitTelescope(wl.frequencies(nfreq), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    per_freq = sum(2 * tel.npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * 16
    bt = PoolCycledProvider(SyntheticProvider(tel, seed=3003), 32)
    gen = torch.Generator(device=ctx.device).manual_seed(2)
    mp = containers.Map(nside=nside, freq=tel.frequencies, allocate=False)
    mp.attach("map", torch.randn((nfreq, 4, 12 * nside * nside), dtype=torch.float64, device=ctx.device, generator=gen))
    task = SimulateSidereal(pool_bytes=32 * per_freq + (1 << 20)) if "pool_bytes" in getattr(SimulateSidereal, "_config_names", ()) else SimulateSidereal()
    task.setup(bt)
    task.process(mp)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ss = task.process(mp)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out = {"config": "cfg3", "SimulateSidereal_s_per_day": min(ts), "all": ts}
    alm = ctx.empty((nfreq, 4, lmax + 1, lmax + 1), np.complex128)
    ctx.timer_start()
    _lib.check(_lib.lib.dmm_map2alm(ctx.handle, ptr(mp.map._dev if hasattr(mp.map, "_dev") else mp.map), nfreq, 4, lmax, lmax, nside, 3, ptr(alm)))
    out["map2alm_iter3_ms"] = ctx.timer_stop()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
