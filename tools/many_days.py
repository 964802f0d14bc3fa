#!/usr/bin/env python
"""D sidereal days against ONE pass over the beam transfers (`BaseMapMaker.process_many`, `dmm_dirty_run_multi`) at
cfg-3 size: (a) B resident in HBM (hbm-pool policy, 16 frequencies' distinct tiles), the whole days through
MModeTransform.process + DirtyMapMaker.process_many for D = 1, 2, 4, 8, with HIP-event times of the multi-day Dirty
launches; (b) B streamed from pinned host memory (4 frequencies' tiles, scaled to the 256-frequency day) for
D = 1, 4, 16.  Reported per DAY-EQUIVALENT: (mmax + 1) D / T.

    python tools/many_days.py [resident|host|both] > gpurun_out/many_days.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK, F64_PEAK = 8000.0, 78.6


def main():
    import torch

    from draco_amd import workloads as wl
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.hoststage import HostStager
    from draco_amd.core.products import PackedStoreProvider, PoolCycledProvider, SyntheticProvider, TransitTelescope
    from draco_amd.device import Context

    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    ctx = Context.get()
    cfg = wl.CONFIGS[3]
    nfreq, nra, lmax, nside = cfg["nfreq"], cfg["nra"], cfg["lmax"], cfg["nside"]
    out = {"config": "cfg3", "unit": "m-modes/s per day-equivalent = (mmax+1) D / T"}
    gen = torch.Generator(device=ctx.device).manual_seed(5)

    if what in ("resident", "both"):
        pool_freqs = 16
        tel = TransitTelescope(wl.frequencies(nfreq), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
        npairs = tel.npairs
        per_freq = sum(2 * npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * 16
        bt = PoolCycledProvider(SyntheticProvider(tel, seed=3003), pool_freqs)
        mt = MModeTransform()
        mt.setup(bt)
        dm = DirtyMapMaker(nside=nside, pool_bytes=pool_freqs * per_freq + (1 << 20))
        dm.setup(bt)
        Dmax = 8
        streams = []
        for d in range(Dmax):
            ss = containers.SiderealStream(freq=tel.frequencies, ra=nra, stack=npairs, allocate=False)
            ss.attach("vis", torch.randn((nfreq, npairs, nra), dtype=torch.complex64, device=ctx.device, generator=gen))
            w = torch.rand((nfreq, npairs, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
            w[torch.rand(w.shape, dtype=torch.float32, device=ctx.device, generator=gen) < 0.01] = 0.0
            ss.attach("vis_weight", w)
            streams.append(ss)
        ntile = pool_freqs * (lmax + 1)
        tile_bytes = pool_freqs * per_freq
        res = {}
        eng = dm._get_engine()
        for D in (1, 2, 4, 8):
            def group():
                return dm.process_many([mt.process(s) for s in streams[:D]])

            for _ in range(3):  # (fills the pool the first time; lets the caching allocator reach its steady set of blocks --
                group()         # a group's maps stay with the side stream until its events have passed)
            torch.cuda.synchronize()
            eng.launch_events = []
            steps = 4
            mem0 = torch.cuda.memory_stats()
            t0 = time.perf_counter()
            for _ in range(steps):
                maps = group()
            maps[-1].map._dev
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / steps
            mem1 = torch.cuda.memory_stats()
            alloc = {k: int(mem1.get(k, 0) - mem0.get(k, 0)) for k in ("num_alloc_retries", "num_device_alloc", "num_device_free")}
            alloc["reserved_GB"] = mem1.get("reserved_bytes.all.current", 0) / 1e9
            ms = [a.elapsed_time(b) for a, b, _, _ in eng.launch_events]
            eng.launch_events = None
            launch_ms = float(np.mean(ms))
            by = tile_bytes + D * (ntile * 2 * npairs * 24 + sum(4 * (lmax + 1 - m) * 16 for m in range(lmax + 1)) * pool_freqs)
            fl = 8.0 * D * tile_bytes / 16  # 8 flop per element of B and day
            res[f"D={D}"] = {"seconds_per_group": el, "value": (lmax + 1) * D / el, "ms_per_day_equivalent": el / D * 1e3,
                            "dirty_launch_ms": launch_ms, "launches_per_group": len(ms) // steps,
                            "hbm_GBs": by / launch_ms / 1e6, "hbm_frac": by / launch_ms / 1e6 / HBM_PEAK,
                            "f64_TFLOPs": fl / launch_ms / 1e9, "f64_frac": fl / launch_ms / 1e9 / F64_PEAK, "allocator": alloc}
            print(f"resident D={D}", json.dumps(res[f"D={D}"]), file=sys.stderr, flush=True)
            del maps
        # the days' a_lm only (no alm2map): what the solves alone do
        for D in (1, 8):
            mms = [mt.process(s) for s in streams[:D]]
            dm.make_alm_many(mms)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                dm.make_alm_many(mms)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / 3
            res[f"D={D}"]["value_to_alm"] = (lmax + 1) * D / el
            del mms
        res["note"] = f"B resident: {pool_freqs} frequencies' distinct tiles ({tile_bytes/1e9:.1f} GB), 256-frequency days, MModeTransform.process per day + DirtyMapMaker.process_many; the Dirty launch serves D days per read of a slab"
        out["resident"] = res
        del streams, dm, mt
        _solve.release_pools()
        torch.cuda.empty_cache()

    if what in ("host", "both"):
        nf_h = 4
        tel_h = TransitTelescope(wl.frequencies(nf_h), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
        shape = (lmax + 1, 2, nf_h, tel_h.npairs)
        Dmax = 16
        days = []
        for d in range(Dmax):
            mm = containers.MModes(mmax=lmax, freq=tel_h.frequencies, stack=tel_h.npairs, allocate=False)
            mm.attach("vis", torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen))
            mm.attach("vis_weight", torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5)
            days.append(mm)
        hs = {}
        for b_dtype, npdt in (("complex128", np.complex128), ("complex64", np.complex64)):
            store = PackedStoreProvider.from_provider(SyntheticProvider(tel_h, seed=9), ctx, npdt, pin=True)
            per_f = store.per_freq * np.dtype(npdt).itemsize
            t_ = DirtyMapMaker(nside=64, b_dtype=b_dtype, pool_bytes=int(2 * 1.05 * per_f))
            t_.setup(store)
            rec = {}
            for D in (1, 4, 16):
                best = None
                for _ in range(2):
                    _solve.release_pools()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    maps = t_.process_many(days[:D])
                    maps[-1].map._dev
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                    del maps
                nb = t_._get_engine().last_b_bytes
                rec[f"D={D}"] = {"value": (lmax + 1) * D / (best * nfreq / nf_h), "seconds": best, "h2d_GBs": nb / best / 1e9, "b_GB": nb / 1e9}
                print(f"host {b_dtype} D={D}", json.dumps(rec[f"D={D}"]), file=sys.stderr, flush=True)
            hs[b_dtype] = rec
            del store, t_
        hs["note"] = f"DirtyMapMaker.process_many with a PackedStoreProvider over pinned host memory ({nf_h} frequencies' cfg-3 tiles cross PCIe ONCE per group of D days, double-buffered under the solves), alm2map to nside 64 included, scaled to the {nfreq}-frequency day"
        out["host_stream"] = hs
        HostStager.release()
        _solve.release_pools()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
