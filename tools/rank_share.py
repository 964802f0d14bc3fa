#!/usr/bin/env python
"""ONE rank's share of the 8-GPU jobs BASELINE.json names as cfg 4 and cfg 5, timed alone on one MI355X and scaled to the
rank's frequencies -- a PROJECTION (no collective inside the timed region; RCCL has never seen N > 1 ranks here).

  cfg 4 (256 feeds, 512 freq, 2048 RA, DirtyMapMaker over 8 GPUs): a rank owns 64 frequencies; one frequency's B is 51 GB,
        so `pool` frequencies' distinct tiles are resident (hbm-pool policy) and the rank's day cycles through them.
  cfg 5 (256 feeds, 1024 freq, 2047 RA: SimulateSidereal + noise + MModeTransform + WienerMapMaker over 8 GPUs): a rank owns
        128 frequencies.  The noise step is the host-side Gaussian one (the Wishart draw is host NumPy per (freq, ra):
        input generation, `tests/test_gpu_beamscreen.py` runs it inside the chain at one frequency) and is reported apart.

    python tools/rank_share.py [--timed-freqs 8] [--pool 4] > gpurun_out/rank_share.json
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timed-freqs", type=int, default=8)
    ap.add_argument("--pool", type=int, default=4)
    ap.add_argument("--configs", type=int, nargs="*", default=[4, 5])
    args = ap.parse_args()

    import torch

    from draco_amd import _lib
    from draco_amd import workloads as wl
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import DirtyMapMaker, WienerMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import PoolCycledProvider, SyntheticProvider, TransitTelescope
    from draco_amd.device import Context, ptr
    from draco_amd.synthesis.noise import GaussianNoise
    from draco_amd.synthesis.stream import SimulateSidereal

    ctx = Context.get()
    lib = _lib.lib
    HBM = 8000.0  # GB/s (guide)
    FP64 = 78.6e12

    def counter(name):
        v = C.c_int64()
        _lib.check(lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    def wall(fn, reps=2):
        ts = []
        out = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return out, min(ts)

    def launches(eng, fn):
        eng.launch_events = []
        out = fn()
        torch.cuda.synchronize()
        ev, eng.launch_events = eng.launch_events, None
        ms = sum(a.elapsed_time(b) for a, b, _, _ in ev)
        by = sum(b for _, _, b, _ in ev)
        return out, ms, by

    res = {"tool": "python tools/rank_share.py", "note": "projection, not a measurement: one rank's share alone on one GPU, scaled linearly in frequencies (they are independent)"}
    for cfgno in args.configs:
        cfg = wl.CONFIGS[cfgno]
        nf, pool = args.timed_freqs, args.pool
        rank_freqs = cfg["nfreq"] // 8
        lmax, nside, nra = cfg["lmax"], cfg["nside"], cfg["nra"]
        tel = TransitTelescope(wl.frequencies(cfg["nfreq"])[:nf], lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
        per_freq = sum(2 * tel.npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * 16
        bt = PoolCycledProvider(SyntheticProvider(tel, seed=3000 + cfgno), pool)
        pool_bytes = pool * per_freq + (1 << 20)
        scale = rank_freqs / nf
        r = {"rank_frequencies": rank_freqs, "frequencies_timed": nf, "pool_frequencies_resident": pool, "B_GB_per_frequency": per_freq / 1e9,
             "npairs": tel.npairs, "lmax": lmax, "nside": nside, "nra": nra}
        gen = torch.Generator(device=ctx.device).manual_seed(cfgno)
        if cfgno == 4:
            vis = torch.randn((nf, tel.npairs, nra), dtype=torch.complex64, device=ctx.device, generator=gen)
            w = torch.rand((nf, tel.npairs, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
            ss = containers.SiderealStream(freq=tel.frequencies, ra=nra, stack=tel.npairs, allocate=False)
            ss.attach("vis", vis)
            ss.attach("vis_weight", w)
            mt = MModeTransform()
            mt.setup(bt)
            dm = DirtyMapMaker(nside=nside, pool_bytes=pool_bytes)
            dm.setup(bt)
            dm.process(mt.process(ss))  # fills the pool
            _, t_day = wall(lambda: dm.process(mt.process(ss)))
            eng = dm._get_engine()
            mm = mt.process(ss)
            _, ms, by = launches(eng, lambda: dm.make_alm(mm))
            r.update({"day_s_timed": t_day, "rank_day_s": t_day * scale, "m_modes_per_s_if_all_ranks_alike": (lmax + 1) / (t_day * scale),
                      "k_dirty_ms_timed": ms, "k_dirty_hbm_frac": by / 1e9 / (ms * 1e-3) / HBM})
            del dm, mt, ss, vis, w, mm
        else:
            sky = torch.randn((nf, 4, 12 * nside * nside), dtype=torch.float64, device=ctx.device, generator=gen)
            mp = containers.Map(nside=nside, freq=tel.frequencies, allocate=False)
            mp.attach("map", sky)
            sim = SimulateSidereal(pool_bytes=pool_bytes)
            sim.setup(bt)
            sim.process(mp)
            ss, t_sim = wall(lambda: sim.process(mp))
            eng = sim._get_engine()
            _, ms_p, by_p = launches(eng, lambda: sim.process(mp))
            gn = GaussianNoise(seed=1, ndays=733.0, recv_temp=50.0)
            gn.setup(tel)
            t0 = time.perf_counter()
            ss = gn.process(ss)
            t_noise = time.perf_counter() - t0
            mt = MModeTransform()
            mt.setup(bt)
            mm, t_mt = wall(lambda: mt.process(ss))
            wm = WienerMapMaker(nside=nside, pool_bytes=pool_bytes)
            wm.setup(bt)
            wm.process(mm)
            _lib.check(lib.dmm_ctx_set_option(ctx.handle, b"profile", 1))
            _, t_w = wall(lambda: wm.process(mm), reps=1)
            span_ms = counter(b"prof_solve_us") / 1e3
            _lib.check(lib.dmm_ctx_set_option(ctx.handle, b"profile", 0))
            ntel = 2 * tel.npairs
            flops = 0.0
            for m in range(lmax + 1):  # Hermitian half of the smaller Gram matrix + Cholesky
                K = 4 * (lmax + 1 - m)
                k_, K_ = (ntel, K) if K >= ntel else (K, ntel)
                flops += 4.0 * k_ * k_ * K_ + (8.0 / 3.0) * k_**3
            flops *= nf
            r.update({"SimulateSidereal_s_timed": t_sim, "k_project_ms_timed": ms_p, "k_project_hbm_frac": by_p / 1e9 / (ms_p * 1e-3) / HBM,
                      "GaussianNoise_host_s_timed": t_noise, "MModeTransform_s_timed": t_mt, "WienerMapMaker_s_timed": t_w,
                      "wiener_span_ms_timed": span_ms, "wiener_span_frac_of_fp64_peak": flops / (span_ms * 1e-3) / FP64, "wiener_order": ((ntel + 63) // 64) * 64,
                      "rank_day_s_gpu_stages": (t_sim + t_mt + t_w) * scale, "rank_day_s_with_host_noise": (t_sim + t_noise + t_mt + t_w) * scale})
            del wm, mt, sim, ss, mm, sky, mp
        res[f"cfg{cfgno}"] = r
        print(json.dumps({f"cfg{cfgno}": r}), file=sys.stderr, flush=True)
        _solve.release_pools()
        torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
