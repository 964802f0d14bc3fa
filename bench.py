#!/usr/bin/env python
"""Headline benchmark: m-modes/sec through MModeTransform + DirtyMapMaker.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

One *step* = one sidereal day of synthetic CHIME-pathfinder-shaped data at BASELINE.json's metric configuration
(128 feeds -> 379 stacked baselines, 256 frequencies, 1024 RA samples, lmax = mmax = 512; cfg 3 of SURVEY.md 8d)
through the PRODUCT's task classes: ``MModeTransform.process`` (sidereal-time -> m FFT + pack + noise weights of all
97 024 (freq, baseline) rows) then ``DirtyMapMaker.process`` (all 131 328 (m, freq) solves a = B^H N^-1 v and the
inverse SHT to IQUV HEALPix maps, mapmaker.py:35-118).  The SiderealStream and the B pool are resident in HBM when
the clock starts; it stops when the maps of every frequency are resident in HBM.

B residency (stated with every number): all B_m[f] of cfg 3 are 1.64 TB in complex128 (l >= m columns only) and
cannot be resident on one GPU, so the job runs under SURVEY 8d's *hbm-pool* policy -- the provider
(``PoolCycledProvider``) serves frequency f with the tiles of f % pool_freqs, ``pool_freqs`` frequencies' worth of
DISTINCT tiles (default 32 -> 205 GB, >> 256 MiB Infinity Cache) stay resident across slabs and across days: every
byte of B is read from HBM exactly once per solve, tile contents repeat every pool_freqs frequencies.

Multi-GPU (one rank per GPU over RCCL; frequency is the path's shard axis, no collective inside the timed region).
``python bench.py --gpus N`` with N > 1 starts the N ranks ITSELF (a child ``python -m torch.distributed.run
--nproc-per-node N bench.py ...``; the parent never touches the GPU and relays rank 0's line); launched under torchrun
(WORLD_SIZE set) it is a rank.  ``--scaling strong`` (default for N > 1): the metric's own job -- cfg 3's 256
frequencies -- is split over the ranks (``parallel.split_local``, the rule of stream.py:73); at N = 8 every rank's 32
frequencies ARE the resident pool (no aliasing), value = (mmax+1) / T_max.  ``--scaling weak``: every rank owns its own
256 frequencies of a 256*N-frequency job, value = N * (mmax+1) / T_max.  After the timed region the rank-local Maps are
all-gathered (``parallel.allgather_map``, the north star's single RCCL collective) and its time is reported.

Prints ONE compact JSON line (< 4 KB) LAST on rank 0's stdout; the full record goes to bench_extra.json beside this file.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


EXTRA_FILE = "bench_extra.json"
LINE_LIMIT = 4096  # the driver keeps about 8 KB of stdout: the result line must fit with room to spare


def _sig(x, n=6):
    """Floats to n significant digits (the line is a record, not a checkpoint), containers recursively."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    if isinstance(x, (np.floating,)):
        return _sig(float(x), n)
    if isinstance(x, (np.integer,)):
        return int(x)
    return x


def _cut(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _dense_scalars(extra):
    """Four scalars per dense maker out of the `ml_day` / `wiener_day` records (default day: nothing resident but B)."""
    out = {}
    ml, wi = (extra or {}).get("ml_day"), (extra or {}).get("wiener_day")
    if isinstance(ml, dict) and "ms_per_step" in ml:
        out["ml_day_s"] = ml["ms_per_step"] * 1e-3
        out["ml_day_sample_freqs"] = (ml.get("config") or {}).get("frequencies_timed")  # (the day figures are this sample scaled to the config's frequencies)
        out["ml_gram_frac"] = (ml.get("roofline") or {}).get("frac")
        for r in ml.get("roofline_secondary") or []:
            if "stage 1" in r.get("kernel", ""):
                out["ml_stage1_hbm_frac"] = r.get("frac")
                out["ml_stage1_s"] = r.get("ms_per_day", 0.0) * 1e-3 * (ml.get("day_scale") or 1.0)
    elif isinstance(ml, dict) and "error" in ml:
        out["ml_day_error"] = _cut(ml["error"], 120)
    if isinstance(wi, dict) and "ms_per_step" in wi:
        out["wiener_day_s"] = wi["ms_per_step"] * 1e-3
        out["wiener_day_sample_freqs"] = (wi.get("config") or {}).get("frequencies_timed")
        out["wiener_span_frac"] = (wi.get("roofline") or {}).get("frac")
    elif isinstance(wi, dict) and "error" in wi:
        out["wiener_day_error"] = _cut(wi["error"], 120)
    return out


def compact_record(out, extra_file=EXTRA_FILE):
    """The ONE line the driver parses, from the full record `out`: every key of the bench contract, the roofline and
    cpu_baseline objects, a few scalars of the secondary measurements.  Everything else (per-arm tables, notes, the
    dense makers' full day records, the allocator's counters) stays in `bench_extra.json` beside bench.py."""
    rec = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "data"))
    rec["dtype"] = _cut(out.get("dtype"), 100)
    cfg = out.get("config") or {}
    rec["config"] = {k: (_cut(v, 400) if isinstance(v, str) else v) for k, v in _pick(cfg, ("workload", "tiles", "b_residency", "parallelism", "solves_per_s", "ml_tiles")).items()}
    rf = out.get("roofline") or {}
    rec["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms", "launches", "flops_per_day", "ms_per_day_timed"))
    rec["roofline"]["kernel"] = _cut(rf.get("kernel"), 90)
    if isinstance(rf.get("alone"), dict):
        rec["roofline"]["alone"] = _pick(rf["alone"], ("frac", "avg_launch_ms"))
    if out.get("roofline_secondary"):
        rec["roofline_secondary"] = [dict(_pick(r, ("bound", "achieved", "peak", "unit", "frac", "ms_per_day")), kernel=_cut(r.get("kernel"), 70)) for r in out["roofline_secondary"][:3]]
    cpu = out.get("cpu_baseline")
    if isinstance(cpu, dict):
        c = _pick(cpu, ("value", "unit", "cores", "kind", "repeats", "spread", "values", "blas"))
        share = cpu.get("cpu_share") or {}
        c["cgroup_quota_cores"] = share.get("cgroup_quota_cores")
        c["loadavg"] = share.get("loadavg")
        c["host_cores_visible"] = cpu.get("host_cores_visible")
        c["sample"] = _cut(cpu.get("sample"), 200)
        rec["cpu_baseline"] = c
    else:
        rec["cpu_baseline"] = None
    for k in ("stages_alone_ms", "value_to_alm"):
        if k in out:
            rec[k] = out[k]
    ex = out.get("extra")
    if isinstance(ex, dict):
        rec.update(_dense_scalars(ex))
        sec = {}
        if isinstance(ex.get("many_days"), dict):
            sec["many_days_D1"] = (ex["many_days"].get("D1") or {}).get("value")
            sec["many_days_D8"] = (ex["many_days"].get("D8") or {}).get("value")
        if isinstance(ex.get("b_complex64"), dict):
            sec["b_complex64"] = ex["b_complex64"].get("value")
            sec["b_complex64_frac"] = ex["b_complex64"].get("frac")
        if isinstance(ex.get("raw_abi_to_alm"), dict):
            sec["raw_abi_to_alm"] = ex["raw_abi_to_alm"].get("value")
        hs = ex.get("b_host_stream")
        if isinstance(hs, dict) and isinstance(hs.get("complex128"), dict):
            sec["b_host_stream"] = hs["complex128"].get("value")
            sec["b_host_stream_h2d_GBs"] = hs["complex128"].get("h2d_GBs")
        if "error" in ex:
            sec["error"] = _cut(ex["error"], 160)
        rec["secondary"] = sec
    if isinstance(out.get("allocator"), dict):
        rec["allocator"] = _pick(out["allocator"], ("num_alloc_retries", "num_device_alloc", "reserved_peak_GB"))
    if isinstance(out.get("ranks"), dict):
        rec["ranks"] = _pick(out["ranks"], ("world_size", "ranks_seen_by_all_gather_into_tensor", "distinct_devices", "backend"))
    if isinstance(out.get("allgather"), dict):
        rec["allgather"] = {k: _cut(v, 160) if isinstance(v, str) else v for k, v in _pick(out["allgather"], ("allgather_ms", "shard_GB", "gathered_GB", "frequencies_gathered", "GBs_per_link", "GBs_per_gpu_in", "backend", "skipped", "error")).items()}
    if isinstance(out.get("launcher"), dict):
        rec["launcher"] = _pick(out["launcher"], ("ranks_started", "devices_visible"))
    rec["extra_file"] = extra_file
    return _sig(rec)


def compact_line(out, extra_file=EXTRA_FILE):
    """`compact_record` serialised; if it still does not fit (it always has), optional objects go until it does."""
    rec = compact_record(out, extra_file)
    line = json.dumps(rec, separators=(", ", ": "))
    for k in ("secondary", "allocator", "roofline_secondary", "stages_alone_ms", "launcher"):
        if len(line) < LINE_LIMIT:
            break
        rec.pop(k, None)
        line = json.dumps(rec, separators=(", ", ": "))
    assert len(line) < LINE_LIMIT, len(line)
    return line


def emit(out):
    """Write the full record beside bench.py (and under gpurun_out/, the one directory a GPU box hands back), then print
    the compact line LAST on stdout."""
    full = json.dumps(_sig(out, 9), indent=1)
    for path in (os.path.join(ROOT, EXTRA_FILE), os.path.join(ROOT, "gpurun_out", EXTRA_FILE)):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as fh:
                fh.write(full + "\n")
        except OSError as e:  # a read-only tree must not cost the line
            print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    sys.stderr.flush()
    print(compact_line(out), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="sidereal days timed")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="SURVEY 8d config number (metric is quoted on 3)")
    ap.add_argument("--maker", default="dirty", choices=["dirty", "ml", "wiener"], help="map-maker of the timed day (the headline metric is quoted on dirty; cfg 3 of BASELINE.json names ml)")
    ap.add_argument("--tiles", default=None, choices=["random", "screen"], help="B tile source: counter-hash tiles (SyntheticProvider) or physically structured ones (BeamScreenProvider); default: random for dirty, screen for ml / wiener")
    ap.add_argument("--band", default="spread", choices=["spread", "low"], help="ml / wiener on structured tiles: the resident pool's frequencies span the config's band (default: a telescope reaches higher m and its Gram matrices have higher rank at the top of the band) or are its lowest channels (the sample of rounds 3-4's earlier records)")
    ap.add_argument("--basis-resident", action="store_true", help="ml: keep the singular bases (U, Sigma) of the resident telescope-side beam transfers beside the B block (MaximumLikelihoodMapMaker.cache_beam_basis: the day's eigenproblem has the order of the beam transfer's numerical rank; the warm-up day builds them) -- a labelled mode, not the default")
    ap.add_argument("--gram-resident", action="store_true", help="ml / wiener: keep the beam Gram products (B B^H, for Wiener B S B^H) of the resident telescope-side tiles beside the B block (task attribute cache_beam_gram: multi-day processing; the warm-up day fills them) -- a labelled mode, not the default")
    ap.add_argument("--freqs", type=int, default=0, help="ml / wiener: frequencies of the timed day (0 = all of the config's; fewer = a stated sample, scaled)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="N > 1: 'strong' (default) splits the metric's 256 frequencies over the ranks -- the job BASELINE.json names; 'weak' gives every rank its own 256")
    ap.add_argument("--b-dtype", default="complex128", choices=["complex128", "complex64"])
    ap.add_argument("--pool-freqs", type=int, default=0, help="frequencies' worth of distinct B tiles resident (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="CPU baseline: budget of timed windows (the process arm gets 3 windows of 0.22 x this)")
    ap.add_argument("--no-overlap", action="store_true", help="A/B switch: alm2map on the caller's stream between the slabs' solves (DirtyMapMaker.overlap_sht = False) instead of beside them")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--extra-resident", action="store_true", help="secondary measurements: also the labelled multi-day modes of ML / Wiener (beam Gram products / singular bases resident beside B); off by default")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --same-device: rehearse the N>1 code path on a one-GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    return ap.parse_args()


def _cpu_worker(rank, barrier, queue, seed, npairs, nra, lmax, ms, repeats, window):
    """One process of the multi-process CPU arm (1 BLAS thread; the reference's MPI decomposition over frequency): the
    m-mode transform of one frequency's rows, then `repeats` timed windows of Dirty solves on its own RAM pool of tiles
    -- every process enters each window together (barrier), so a window is the machine under the full arm --, then the
    same tiles against 16 days at once."""
    from oracle import mapmaker as omm
    from oracle import synth as osyn
    from oracle import transform as otr

    rng = np.random.default_rng(seed)
    vis = (rng.standard_normal((1, npairs, nra), dtype=np.float32) + 1j * rng.standard_normal((1, npairs, nra), dtype=np.float32)).astype(np.complex64)
    w = rng.uniform(0.5, 1.5, (1, npairs, nra)).astype(np.float32)
    otr.mmode_transform(vis, w, mmax=lmax)
    tiles = [osyn.beam_tile(3000, int(m), seed % 7, npairs, 4, lmax) for m in ms]
    v = rng.standard_normal((2, npairs)) + 1j * rng.standard_normal((2, npairs))
    Ni = rng.uniform(0.5, 1.5, (2, npairs))
    omm.dirty_solve(tiles[0], v, Ni)
    res = {"rank": rank, "solves": [], "seconds": []}
    barrier.wait()
    t0 = time.perf_counter()
    nfft = 0
    while nfft < 2 or time.perf_counter() - t0 < 0.25 * window:
        otr.mmode_transform(vis, w, mmax=lmax)
        nfft += 1
    res["t_fft"] = (time.perf_counter() - t0) / nfft  # seconds per frequency, this process
    barrier.wait()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5 * window:  # untimed warm-up: the first window of a cold pool is 30 % slow
        for bm in tiles:
            omm.dirty_solve(bm, v, Ni)
    for _ in range(repeats):
        barrier.wait()
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < window:
            for bm in tiles:
                omm.dirty_solve(bm, v, Ni)
                n += 1
        res["solves"].append(n)
        res["seconds"].append(time.perf_counter() - t0)
    # 16 days per read of B (one complex128 matrix product per tile: the CPU's way of sharing the read of B between
    # days, next to the GPU's process_many)
    D = 16
    vs = rng.standard_normal((D, 2, npairs)) + 1j * rng.standard_normal((D, 2, npairs))
    Nis = rng.uniform(0.5, 1.5, (D, 2, npairs))
    omm.dirty_solve_many(tiles[0], vs, Nis)
    barrier.wait()
    nm = 0
    t0 = time.perf_counter()
    while nm < 1 or time.perf_counter() - t0 < 0.25 * window:
        for bm in tiles:
            omm.dirty_solve_many(bm, vs, Nis)
            nm += 1
    res["many"] = (nm * D, time.perf_counter() - t0)
    queue.put(res)


def _process_arm(nproc, npairs, nra, lmax, ms, repeats, window):
    """`nproc` single-threaded worker processes at once.  Returns per timed window the aggregate seconds per solve, and
    the aggregate seconds per frequency of the transform, solves done, seconds per day-solve at 16 days per tile read."""
    import multiprocessing as mp

    keys = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")
    saved = {k: os.environ.get(k) for k in keys}
    os.environ.update({k: "1" for k in keys})
    ctx = mp.get_context("spawn")
    barrier, queue = ctx.Barrier(nproc), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(i, barrier, queue, 100 + i, npairs, nra, lmax, ms[i % len(ms) :: 3][:6], repeats, window), daemon=True) for i in range(nproc)]
    try:
        for p_ in procs:
            p_.start()
        deadline = time.time() + 60 + (repeats + 2) * window * 6
        res = []
        while len(res) < nproc:
            try:
                res.append(queue.get(timeout=1.0))
            except Exception:  # queue.Empty
                if time.time() > deadline or any(p_.exitcode not in (None, 0) for p_ in procs):
                    barrier.abort()
                    raise RuntimeError(f"{len(res)} of {nproc} workers reported")
        for p_ in procs:
            p_.join(timeout=10)
    finally:
        for p_ in procs:
            if p_.is_alive():
                p_.kill()  # (exact children of this call)
        for k, v_ in saved.items():
            if v_ is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v_
    per_window = [1.0 / sum(r["solves"][i] / r["seconds"][i] for r in res) for i in range(repeats)]  # seconds per solve, all processes together
    fft_rate = sum(1.0 / r["t_fft"] for r in res)  # frequencies per second, all processes together
    many_rate = sum(r["many"][0] / r["many"][1] for r in res)  # day-solves per second with 16 days per tile read
    return per_window, 1.0 / fft_rate, [sum(r["solves"][i] for r in res) for i in range(repeats)], 1.0 / many_rate


def _blas_vendor():
    """BLAS behind NumPy's dot ('print the core count and BLAS vendor')."""
    try:
        from threadpoolctl import threadpool_info

        return "; ".join(f"{d.get('internal_api')} {d.get('version')} ({d.get('threading_layer', d.get('user_api'))})" for d in threadpool_info() if d.get("user_api") == "blas") or "unknown"
    except Exception:
        return "unknown"


def _cpu_share():
    """What the box actually grants this process: the cgroup CPU quota (v2 `cpu.max`, v1 `cpu.cfs_quota_us` /
    `cpu.cfs_period_us`), next to the affinity mask -- a 256-thread host may schedule a one-GPU job on a 16-core share,
    which is why an arm with 256 processes can be slower than one with 16."""
    rec = {"cpu_count": os.cpu_count()}
    try:
        rec["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        rec["affinity"] = None

    def read(path):
        try:
            with open(path) as fh:
                return fh.read().strip()
        except Exception:
            return None

    quota = None
    v2 = None
    try:  # cgroup v2: the process's own group, then the root of the mount
        rel = [ln.split("::", 1)[1].strip() for ln in (read("/proc/self/cgroup") or "").splitlines() if ln.startswith("0::")]
        for base in ([os.path.join("/sys/fs/cgroup", rel[0].lstrip("/"))] if rel else []) + ["/sys/fs/cgroup"]:
            v2 = read(os.path.join(base, "cpu.max"))
            if v2:
                break
    except Exception:
        v2 = None
    if v2:
        rec["cgroup_cpu_max"] = v2
        parts = v2.split()
        if parts[0] != "max":
            quota = float(parts[0]) / float(parts[1] if len(parts) > 1 else 100000)
    else:
        q, per = read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        if q and per:
            rec["cgroup_cfs_quota_us"], rec["cgroup_cfs_period_us"] = q, per
            if float(q) > 0:
                quota = float(q) / float(per)
    rec["cgroup_quota_cores"] = quota  # None: no quota set (or not readable)
    rec["loadavg"] = read("/proc/loadavg")
    return rec


def cpu_baseline(cfg, seconds):
    """Oracle (NumPy restatement of the reference) timed on this box's host cores.

    Bounded sample of the SAME workload: the complex64 FFT + pack of whole frequencies, and Dirty solves (complex128
    np.dot on the FULL tile, like the reference) over an m-stratified set of tiles drawn from RAM pools; extrapolated
    linearly to the full job.  HDF5 I/O of B (dominant in real reference runs) is excluded, as on the GPU side.  The
    final alm2map (healpy's C++ in the reference; the NumPy oracle would overstate it by orders of magnitude) is NOT
    charged to the CPU time, although the GPU step includes it: the CPU figure is an upper bound.

    Arms (SURVEY 8d / BASELINE.md): (i) one process, one BLAS thread; (ii) one process, BLAS threads = the cores this
    job is granted; (iii) P single-threaded processes, P = the cgroup's CPU quota (16 on a one-GPU box; every core of the
    affinity mask when there is no quota) -- the reference's MPI-over-frequency decomposition, and the arm that wins.
    (i) and (ii) get a short window each; (iii) gets a warm-up and five barrier-synchronised windows of its own (>= 10 s
    together at the default budget): `value` is the BEST window (a loaded host only ever slows a window down), `values` lists all of
    them and `spread` = (max - min) / max says how much the host moved underneath.  In (iii) the
    transform is timed INSIDE the workers, all of them at once.  Runs BEFORE the GPU is touched (workers are spawned).
    """
    from oracle import mapmaker as omm
    from oracle import synth as osyn
    from oracle import transform as otr

    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    share = _cpu_share()
    quota = share.get("cgroup_quota_cores")
    granted = max(1, min(ncpu, int(quota + 0.5))) if quota else ncpu  # the cores this job may actually keep busy
    npairs = osyn.npairs_of(cfg["ncyl"], cfg["nfeed_cyl"])
    nfreq, nra, lmax = cfg["nfreq"], cfg["nra"], cfg["lmax"]
    rng = np.random.default_rng(0)
    short = max(0.5, seconds * 0.08)  # windows of arms (i) and (ii)
    repeats = 5
    window = max(0.5, seconds * 0.13)  # (iii): a warm-up window + 5 timed windows + transform and many-days (0.25 w each)

    # single-process arms: transform of a slice of frequencies, then solves on a RAM pool of stratified tiles
    nf_s = max(1, min(nfreq, 2))
    vis = (rng.standard_normal((nf_s, npairs, nra), dtype=np.float32) + 1j * rng.standard_normal((nf_s, npairs, nra), dtype=np.float32)).astype(np.complex64)
    w = rng.uniform(0.5, 1.5, (nf_s, npairs, nra)).astype(np.float32)
    otr.mmode_transform(vis[:1], w[:1], mmax=lmax)
    t0 = time.perf_counter()
    otr.mmode_transform(vis, w, mmax=lmax)
    t_fft_1 = (time.perf_counter() - t0) / nf_s
    ms = np.unique(np.linspace(0, lmax, 24).astype(int))
    tiles = [osyn.beam_tile(3000, int(m), 0, npairs, 4, lmax) for m in ms[::2]]
    v = rng.standard_normal((2, npairs)) + 1j * rng.standard_normal((2, npairs))
    Ni = rng.uniform(0.5, 1.5, (2, npairs))

    def solve_arm(nthreads, budget):
        import contextlib

        cm = threadpool_limits(limits=nthreads) if threadpool_limits is not None else contextlib.nullcontext()
        with cm:
            omm.dirty_solve(tiles[0], v, Ni)
            t_end = time.perf_counter() + budget
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() < t_end:
                for bm in tiles:
                    omm.dirty_solve(bm, v, Ni)
                    n += 1
            # the reference multiplies the FULL tile (zeros included): the cost does not depend on m
            return (time.perf_counter() - t0) / n, n

    def job_value(ms_per_solve, fft_ms_per_freq):
        return (lmax + 1) / (fft_ms_per_freq * 1e-3 * nfreq + ms_per_solve * 1e-3 * (lmax + 1) * nfreq)

    arms = {}
    t1, n1 = solve_arm(1, short)
    arms["1_thread"] = {"cores": 1, "ms_per_solve": t1 * 1e3, "fft_ms_per_freq": t_fft_1 * 1e3, "solves": n1, "values": [job_value(t1 * 1e3, t_fft_1 * 1e3)]}
    if threadpool_limits is not None and granted > 1:
        tn, nn = solve_arm(granted, short)
        arms[f"{granted}_blas_threads"] = {"cores": granted, "ms_per_solve": tn * 1e3, "fft_ms_per_freq": t_fft_1 * 1e3, "solves": nn, "values": [job_value(tn * 1e3, t_fft_1 * 1e3)]}
    del tiles
    nproc = min(granted, 64)
    if nproc > 1:
        try:
            tps, tf, nprs, tmany = _process_arm(nproc, npairs, nra, lmax, ms, repeats, window)
            best_i = int(np.argmin(tps))
            arms[f"{nproc}_processes"] = {"cores": nproc, "ms_per_solve": tps[best_i] * 1e3, "ms_per_solve_windows": [t * 1e3 for t in tps], "fft_ms_per_freq": tf * 1e3,
                                          "solves": int(sum(nprs)), "window_seconds": window, "values": [job_value(t * 1e3, tf * 1e3) for t in tps],
                                          "ms_per_day_solve_16_days_per_tile_read": tmany * 1e3}
        except Exception as e:  # the baseline must never break the bench line
            print(f"cpu_baseline: {nproc}-process arm failed: {e!r}", file=sys.stderr)
    for a in arms.values():
        a["m_modes_per_s"] = max(a["values"])
        a["job_seconds"] = (lmax + 1) / a["m_modes_per_s"]
        a["spread"] = (max(a["values"]) - min(a["values"])) / max(a["values"])
    for a in arms.values():
        if "ms_per_day_solve_16_days_per_tile_read" in a:  # 16 days per read of B (one zgemm per tile): per day-equivalent
            a["m_modes_per_s_16_days"] = (lmax + 1) / (a["fft_ms_per_freq"] * 1e-3 * nfreq + a["ms_per_day_solve_16_days_per_tile_read"] * 1e-3 * (lmax + 1) * nfreq)
    best_name = min(arms, key=lambda k: arms[k]["job_seconds"])
    best = arms[best_name]
    share_after = _cpu_share()
    return {
        "value": best["m_modes_per_s"],
        "unit": "m-modes/s",
        "cores": int(best["cores"]),
        "kind": "port",
        "repeats": len(best["values"]),
        "values": best["values"],
        "spread": best["spread"],
        "many_days": {"value": max((a.get("m_modes_per_s_16_days", 0.0) for a in arms.values()), default=0.0) or None, "unit": "m-modes/s per day-equivalent",
                      "note": "the process arm again with 16 days per tile read (oracle.dirty_solve_many: one complex128 matrix product B^H [N_d v_d] per tile), the CPU counterpart of extra.many_days / b_host_stream D=16"},
        "sample": f"best of {len(best['values'])} windows of arm '{best_name}' ({best['solves']} Dirty solves, np.dot c128 on full {2*npairs}x{4*(lmax+1)} tiles from RAM pools, + FFT+pack of whole frequencies), extrapolated linearly to {(lmax+1)*nfreq} solves + {nfreq} freq; alm2map not charged; arms {list(arms)}, {seconds:.0f} s budget",
        "host_cores_visible": ncpu,
        "cores_granted": granted,
        "cpu_share": share_after,
        "loadavg_before": share.get("loadavg"),
        "blas": _blas_vendor(),
        "arms": arms,
    }


def _cpu_dense_worker(args):
    """One single-threaded process of a multi-process CPU arm of the dense map-makers: ML (SVD pseudo-inverse) or
    Wiener (Hermitian-PD solve) on its own tiles until the budget is spent."""
    kind, seed, npairs, lmax, ms, budget = args
    from oracle import mapmaker as omm
    from oracle import synth as osyn

    rng = np.random.default_rng(seed)
    v = rng.standard_normal((2, npairs)) + 1j * rng.standard_normal((2, npairs))
    Ni = rng.uniform(0.5, 1.5, (2, npairs)) * 20
    n, spent = 0, 0.0
    while n < 1 or spent < budget:
        m = int(ms[n % len(ms)])
        bm = osyn.beam_tile(3000, m, seed % 7, npairs, 4, lmax)  # (tile generation is not charged)
        t1 = time.perf_counter()
        omm.ml_solve(bm, v, Ni) if kind == "ml" else omm.wiener_solve(bm, m, v, Ni, 1.0, 0.5)
        spent += time.perf_counter() - t1
        n += 1
    return n, spent


def cpu_baseline_dense(cfg, kind, seconds):
    """The oracle's ML (``pinv_svd`` restated: scipy.linalg.svd of the 758 x 2052 tile, mapmaker.py:184-201,287-300) or
    Wiener solve (Hermitian-PD solve of the smaller normal system, mapmaker.py:235-284) timed on this box's host cores,
    on an m-stratified sample of cfg-3 tiles, extrapolated linearly to the day's solves.  Arms as for the Dirty
    baseline: 1 thread; one process with every BLAS thread; 16 single-threaded processes.  Tile contents are the
    counter-hash tiles (LAPACK's cost does not depend on them; generating structured tiles on the host would cost more
    than the solves); generation time is excluded.  The m-mode transform is negligible next to these solves and is not
    charged."""
    from oracle import mapmaker as omm
    from oracle import synth as osyn

    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    import contextlib

    npairs = osyn.npairs_of(cfg["ncyl"], cfg["nfeed_cyl"])
    nfreq, lmax = cfg["nfreq"], cfg["lmax"]
    ms = np.unique(np.linspace(0, lmax, 9).astype(int))  # stratified in m: the cost of both solvers depends on it
    rng = np.random.default_rng(1)
    v = rng.standard_normal((2, npairs)) + 1j * rng.standard_normal((2, npairs))
    Ni = rng.uniform(0.5, 1.5, (2, npairs)) * 20
    budget = seconds / 3

    def arm(nthreads):
        cm = threadpool_limits(limits=nthreads) if threadpool_limits is not None else contextlib.nullcontext()
        spent, n = 0.0, 0
        with cm:
            while n < len(ms) and (n < 2 or spent < budget):  # one pass over the strata at most, at least two solves
                bm = osyn.beam_tile(3000, int(ms[n]), 0, npairs, 4, lmax)
                t0 = time.perf_counter()
                omm.ml_solve(bm, v, Ni) if kind == "ml" else omm.wiener_solve(bm, int(ms[n]), v, Ni, 1.0, 0.5)
                spent += time.perf_counter() - t0
                n += 1
        return spent / n, n

    arms = {}
    t1, n1 = arm(1)
    arms["1_thread"] = {"cores": 1, "ms_per_solve": t1 * 1e3, "solves": n1}
    if threadpool_limits is not None and ncpu > 1:
        tn, nn = arm(ncpu)
        arms[f"{ncpu}_blas_threads"] = {"cores": ncpu, "ms_per_solve": tn * 1e3, "solves": nn}
    nproc = min(ncpu, 16)
    if nproc > 1:
        try:
            import multiprocessing as mp

            keys = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")
            saved = {k: os.environ.get(k) for k in keys}
            os.environ.update({k: "1" for k in keys})
            try:
                with mp.get_context("spawn").Pool(nproc) as pool:
                    res = pool.map(_cpu_dense_worker, [(kind, 100 + i, npairs, lmax, np.roll(ms, i), budget) for i in range(nproc)], chunksize=1)
            finally:
                for k, v_ in saved.items():
                    os.environ.pop(k, None) if v_ is None else os.environ.__setitem__(k, v_)
            rate = sum(n / t for n, t in res)
            arms[f"{nproc}_processes"] = {"cores": nproc, "ms_per_solve": 1e3 / rate, "solves": sum(n for n, _ in res)}
        except Exception as e:  # the baseline must never break the bench line
            print(f"cpu_baseline_dense: {nproc}-process arm failed: {e!r}", file=sys.stderr)
    for a in arms.values():
        a["job_seconds"] = a["ms_per_solve"] * 1e-3 * (lmax + 1) * nfreq
        a["m_modes_per_s"] = (lmax + 1) / a["job_seconds"]
    best_name = min(arms, key=lambda k: arms[k]["job_seconds"])
    best = arms[best_name]
    what = "scipy.linalg.svd pseudo-inverse (pinv_svd restated)" if kind == "ml" else "Hermitian-PD solve of the smaller normal system"
    return {
        "value": best["m_modes_per_s"], "unit": "m-modes/s", "cores": int(best["cores"]), "kind": "port",
        "sample": f"arm '{best_name}' (fastest of {list(arms)}): {best['solves']} {kind} solves ({what}) on {2*npairs} x {4*(lmax+1)} counter-hash tiles stratified over m = {ms.tolist()}, extrapolated linearly to {(lmax+1)*nfreq} solves; tile generation, the m-mode transform and alm2map not charged",
        "host_cores_visible": ncpu, "cpu_share": _cpu_share(), "blas": _blas_vendor(), "arms": arms,
    }


def dense_flops(cfg, npairs):
    """Useful FP64 flops of ONE frequency of a dense map-maker at the config's sizes: the Hermitian half of the smaller
    Gram matrix of every (m, f) tile, 8 k^2 K / 2, and the factorisation (8/3) k^3."""
    lmax, ntel = cfg["lmax"], 2 * npairs
    gram = chol = 0.0
    for m in range(lmax + 1):
        nsky = 4 * (lmax + 1 - m)
        k, K = min(ntel, nsky), max(ntel, nsky)
        gram += 4.0 * k * k * K
        chol += (8.0 / 3.0) * k**3
    return gram, chol


def main_dense(args, cpu):
    """`--maker ml|wiener`: one cfg-3 day through MModeTransform.process + {MaximumLikelihood,Wiener}MapMaker.process
    (single GPU), B tiles resident under the hbm-pool policy, physically structured by default."""
    out = dense_day(args, args.maker)
    out["cpu_baseline"] = cpu
    emit(out)


def dense_day(args, kind):
    """The record of `--maker ml|wiener` (also the `extra.ml_day` / `extra.wiener_day` entries of the headline line)."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd import workloads as wl
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import DirtyMapMaker, MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import BeamScreenProvider, PoolCycledProvider, SyntheticProvider, TransitTelescope
    from draco_amd.device import Context

    torch.cuda.set_device(0)
    ctx = Context.get()
    cfg = wl.CONFIGS[args.config]
    for opt in ("ml_reduce", "gram_stage", "wiener_overlap", "ml_null", "ml_rank_stop", "ml_chase_split"):  # (A/B switches of the dense solvers: see include/draco_amd.h)
        if os.environ.get("DMM_" + opt.upper()):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, opt.encode(), int(os.environ["DMM_" + opt.upper()])))
    if os.environ.get("DMM_ML_WS_CAP_MIB"):  # A/B: cap the solvers' workspace offer (chunks of fewer matrices)
        _cap = int(os.environ["DMM_ML_WS_CAP_MIB"])
        _offer = _solve.SolveEngine._offer_workspace
        _solve.SolveEngine._offer_workspace = lambda self, option, cap_mib: _offer(self, option, min(cap_mib, _cap))
    tiles = args.tiles or "screen"
    nfreq_cfg, nra, lmax, nside = cfg["nfreq"], cfg["nra"], cfg["lmax"], cfg["nside"]
    nfreq = args.freqs if args.freqs > 0 else nfreq_cfg
    pool_freqs = args.pool_freqs if args.pool_freqs > 0 else 16  # 102 GB of distinct tiles; leaves the solvers their workspace
    pool_freqs = min(pool_freqs, nfreq)
    while nfreq % pool_freqs:
        pool_freqs -= 1
    # The pool's distinct tiles are those of the telescope's first `pool_freqs` channels (the provider aliases f -> f % pool_freqs).
    # Structured tiles depend on the wavelength -- at 800 MHz the telescope reaches twice the m it reaches at 400 MHz and
    # its Gram matrices have twice the numerical rank -- so the pool's channels span the band ("spread"): every
    # (nfreq_cfg / pool_freqs)-th channel of the config, cycled over the day's frequencies.
    band = getattr(args, "band", "spread") if tiles == "screen" else "low"
    if band == "spread":
        tel_freqs = np.resize(wl.frequencies(nfreq_cfg)[:: nfreq_cfg // pool_freqs][:pool_freqs], nfreq)
    else:
        tel_freqs = wl.frequencies(nfreq_cfg)[:nfreq]
    tel = TransitTelescope(tel_freqs, lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    npairs = tel.npairs
    es = 16 if args.b_dtype == "complex128" else 8
    per_freq = sum(2 * npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * es
    base = BeamScreenProvider(tel, seed=3003) if tiles == "screen" else SyntheticProvider(tel, seed=3003)
    bt = PoolCycledProvider(base, pool_freqs)
    gen = torch.Generator(device=ctx.device).manual_seed(1000)
    vis = torch.randn((nfreq, npairs, nra), dtype=torch.complex64, device=ctx.device, generator=gen)
    weight = (torch.rand((nfreq, npairs, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5) * 20.0
    weight[torch.rand(weight.shape, dtype=torch.float32, device=ctx.device, generator=gen) < 0.01] = 0.0
    ss = containers.SiderealStream(freq=tel.frequencies, ra=nra, stack=npairs, allocate=False)
    ss.attach("vis", vis)
    ss.attach("vis_weight", weight)
    mt = MModeTransform()
    mt.setup(bt)
    cls = MaximumLikelihoodMapMaker if kind == "ml" else WienerMapMaker
    gram_resident = bool(getattr(args, "gram_resident", False))
    basis_resident = kind == "ml" and bool(getattr(args, "basis_resident", False))
    task = cls(nside=nside, b_dtype=args.b_dtype, pool_bytes=pool_freqs * per_freq + (1 << 20), **({"cache_beam_gram": True} if gram_resident else {}),
               **({"cache_beam_basis": True} if basis_resident else {}))
    task.setup(bt)

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    # B resident before the clock starts: a Dirty pass over the first slab fills the pool (the engines of equal providers share it)
    t_fill0 = time.perf_counter()
    warm = DirtyMapMaker(nside=nside, b_dtype=args.b_dtype, pool_bytes=pool_freqs * per_freq + (1 << 20))
    warm.setup(bt)
    mm0 = mt.process(ss)
    warm.make_alm(mm0)
    torch.cuda.synchronize()
    t_fill = time.perf_counter() - t_fill0
    del warm, mm0
    for _ in range(args.warmup):
        task.process(mt.process(ss))
    torch.cuda.synchronize()
    eng = task._get_engine()
    fills_before = eng.fills
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"profile", 1))
    c0 = {k: counter(k) for k in (b"ml_tiles_direct", b"ml_tiles_eigen", b"ml_tiles_ql_failed", b"ml_tiles_null", b"ml_gram_flops", b"ml_band_bytes", b"ml_tiles_stopped", b"ml_stop_cols", b"ml_gram_cached", b"ml_tiles_basis")}
    mem0 = torch.cuda.memory_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_map = task.process(mt.process(ss))
    out_map.map._dev  # (orders this stream behind the side-stream SHT)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mem1 = torch.cuda.memory_stats()
    prof = {k: {"ms": counter(f"prof_{k}_us".encode()) / 1e3 / args.steps, "spans": counter(f"prof_{k}_n".encode()) // max(args.steps, 1)}
            for k in ("gram", "chol", "tridiag", "band", "chase", "ql", "backproj", "solve", "null")}
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"profile", 0))
    c1 = {k: counter(k) for k in c0}
    assert eng.fills == fills_before or args.warmup == 0, "B was generated inside the timed region"

    day_s = elapsed / args.steps
    scale = nfreq_cfg / nfreq  # a frequency sample is scaled to the config's day (frequencies are independent)
    value = (lmax + 1) / (day_s * scale)
    gram_fl, chol_fl = dense_flops(cfg, npairs)
    if gram_resident and kind == "wiener":  # only the sky-side products are computed on a timed day (the telescope-side ones are resident)
        gram_fl = sum(4.0 * (4 * (lmax + 1 - m)) ** 2 * (2 * npairs) for m in range(lmax + 1) if 4 * (lmax + 1 - m) < 2 * npairs)
    gram_tf = gram_fl * nfreq / (prof["gram"]["ms"] * 1e-3) / 1e12 if prof["gram"]["ms"] > 0 else None
    if kind == "ml":  # the library's own count of what it formed (tiles answered by the null certificate never get a Gram matrix)
        gram_done = (c1[b"ml_gram_flops"] - c0[b"ml_gram_flops"]) / max(args.steps, 1)
        band_bytes = (c1[b"ml_band_bytes"] - c0[b"ml_band_bytes"]) / max(args.steps, 1)
        gram_tf = gram_done / (prof["gram"]["ms"] * 1e-3) / 1e12 if prof["gram"]["ms"] > 0 else None
    n_direct = (c1[b"ml_tiles_direct"] - c0[b"ml_tiles_direct"]) // max(args.steps, 1)
    n_eigen = (c1[b"ml_tiles_eigen"] - c0[b"ml_tiles_eigen"]) // max(args.steps, 1)
    secondary = []
    if kind == "ml" and (prof["tridiag"]["ms"] > 0 or prof["band"]["ms"] > 0):
        ntel = 2 * npairs
        frac_eig = n_eigen / max(n_direct + n_eigen, 1)
        if prof["band"]["ms"] > 0:
            # stage 1 of the two-stage reduction (k_sb_panel + k_sb_sweep_lo): per panel of 8 columns every 16 x 16 tile of
            # the trailing matrix's lower triangle is read and written once (8 KB) and leaves 0.5 KB of row partials
            by = band_bytes  # (counted by the library per reduced matrix: sum over panels of t (t + 1) / 2 tiles x 8.5 KB)
            ms = prof["band"]["ms"]
            secondary.append({"kernel": "two-stage reduction, stage 1: dense -> band (k_sb_sweep_lo + k_sb_panel)",
                              "bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_day": ms,
                              "note": "bytes = 8.5 KB per lower-triangle tile and panel (DESIGN 5.5); the time is the whole class, panel kernels (latency bound, about a fifth of it) included, measured inside the step with the chase and QL of the previous chunk beside it"})
            secondary.append({"kernel": "two-stage reduction, stage 2: band -> tridiagonal (k_sb_chase, side stream)", "bound": "latency",
                              "ms_per_day": prof["chase"]["ms"], "note": "a dependency chain of 2 (n - 1) block iterations per matrix, the band in LDS: no roofline applies"})
        if prof["tridiag"]["ms"] > 0:
            # one-stage Householder reduction (orders whose band does not fit the LDS, or ml_reduce = 1): n^3/6 * 20 B
            by = sum(min(ntel, 4 * (lmax + 1 - m)) ** 3 / 6.0 * 20.0 for m in range(lmax + 1)) * nfreq * frac_eig
            ms = prof["tridiag"]["ms"]
            secondary.append({"kernel": "one-stage Hermitian tridiagonal reduction (k_td_col + k_td_trail_tri)",
                              "bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_day": ms,
                              "note": "bytes = n^3/6 * 20 B per decomposed tile (a fraction is only meaningful when every tile went this way)"})
    gram_note = "useful flops (one Hermitian half-product of the smaller side per tile: 8 k^2 K / 2) / HIP-event time of every Gram launch of the timed day on its launch stream ('profile' option of the library); ML: the flops are the library's own count of the Gram matrices it formed (counter ml_gram_flops: a tile whose certificate is rejected counts twice, a tile answered by the null certificate not at all)"
    roofline = {"kernel": "k_nt<GRAM/GRAMX> (Hermitian products D B B^H D / B^H N B on v_mfma_f64_16x16x4_f64)",
                "bound": "mfma", "achieved": gram_tf, "peak": 78.6, "unit": "TFLOP/s", "frac": gram_tf / 78.6 if gram_tf else None, "traffic": None,
                "flops_per_day": (gram_done * scale if kind == "ml" else gram_fl * nfreq_cfg), "ms_per_day_timed": prof["gram"]["ms"], "note": gram_note}
    if gram_resident and kind == "ml" and secondary:
        # the day's telescope-side Gram matrices come from the resident products (k_gram_scale: one read of the slot, one
        # write of the matrix's lower blocks; no product computed); what is left of the Gram class are the sky-side
        # products.  Stage 1 of the reduction is the pass's dominant kernel: its roofline leads.
        n_c = (c1[b"ml_gram_cached"] - c0[b"ml_gram_cached"]) / max(args.steps, 1)
        gram_rec = dict(roofline, note=gram_note + f"; --gram-resident: {n_c:.0f} of the timed day's telescope-side Gram matrices were formed from the resident products B B^H (computed by the warm-up day); the flops counted here are the sky-side products', the class time includes k_gram_scale")
        roofline = dict(secondary[0], traffic=None, tiles_from_resident_products_per_day_timed=n_c)
        secondary = [gram_rec] + secondary[1:]
    if kind == "wiener" and prof["solve"]["ms"] > 0:
        # dmm_wiener_run keeps two batches in flight on two streams (one's factorisation beside the other's Gram products):
        # the class sums overlap in time, the span of the whole pass is the time the matrix cores were asked for
        tf = (gram_fl + chol_fl) * nfreq / (prof["solve"]["ms"] * 1e-3) / 1e12
        secondary.append(dict(roofline, note=gram_note + "; HERE the Gram launches of one stream share the chip with the other stream's factorisation, so this busy-time figure is not the kernel alone (alone: profiles/r03_gram_stage_ab.txt, 0.78)"))
        roofline = {"kernel": "dmm_wiener_run: Gram products + blocked Cholesky, all of it k_nt on v_mfma_f64_16x16x4_f64 (two batches in flight on two streams)",
                    "bound": "mfma", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6, "traffic": None,
                    "flops_per_day": (gram_fl + chol_fl) * nfreq_cfg, "ms_per_day_timed": prof["solve"]["ms"],
                    "note": "useful flops (Hermitian half of the smaller Gram matrix 8 k^2 K / 2 + factorisation (8/3) k^3 per tile) / HIP-event span of every dmm_wiener_run of the timed day on the caller's stream" + ("; --gram-resident: the telescope-side products B S B^H are resident (computed by the warm-up day) and NOT counted: the flops are the factorisations and the sky-side products" if gram_resident else "")}
    out = {
        "metric": f"m-modes/sec through MModeTransform+{cls.__name__} (128-feed, 256-freq)",
        "value": value, "unit": "m-modes/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": day_s * scale * 1e3, "day_scale": scale, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 (Gram / factorisations / eigen-solve); B stored " + args.b_dtype + "; FFT complex64 (as the reference)",
        "data": "synthetic",
        "config": {
            "workload": f"cfg{args.config}: {tel.nfeed}-feed ({npairs} stacked baselines), {nfreq} of {nfreq_cfg} freq timed" + (f" (scaled x{scale:g} to the day)" if scale != 1 else "") + f", {nra} RA, lmax=mmax={lmax}: MModeTransform.process + {cls.__name__}.process through the task classes ({(lmax+1)*nfreq} (m,f) solves + alm2map to nside={nside})",
            "tiles": ("physically structured (BeamScreenProvider: per-polarisation Jones screens, narrow east-west primary beam; ill-conditioned Gram matrices like real products)" if tiles == "screen" else "counter-hash (SyntheticProvider: best-conditioned tiles possible)"),
            "b_residency": f"hbm-pool: {pool_freqs} frequencies' B tiles resident ({pool_freqs*per_freq/1e9:.1f} GB distinct, {args.b_dtype}, l>=m packed), provider aliases f -> f % {pool_freqs}" + (f"; the pool's channels span the band ({tel_freqs[0]:.1f} ... {tel_freqs[pool_freqs - 1]:.1f} MHz, every {nfreq_cfg // pool_freqs}th channel of the config)" if band == "spread" else f"; the pool's channels are the config's lowest ({tel_freqs[0]:.1f} ... {tel_freqs[pool_freqs - 1]:.1f} MHz)") + f"; generated on the GPU in {t_fill:.1f} s before the clock starts" + ("; beam Gram products B B^H of the resident telescope-side tiles kept beside them (%.1f GB, filled by the warm-up day)" % (sum(1 for m in range(lmax + 1) if 4 * (lmax + 1 - m) >= 2 * npairs) * pool_freqs * ((2 * npairs + 63) // 64) * ((2 * npairs + 63) // 64 + 1) // 2 * 65536 / 1e9) if gram_resident else "")
                           + ("; singular bases (U, Sigma; up to 448 vectors per tile) of the resident telescope-side tiles kept beside them (%.1f GB, built by the warm-up day)" % (sum(1 for m in range(lmax + 1) if 4 * (lmax + 1 - m) >= 2 * npairs) * pool_freqs * 448 * 2 * npairs * 16 / 1e9) if basis_resident else ""),
            "frequencies_timed": nfreq,
            "solves_per_s": (lmax + 1) * nfreq / day_s,
            "ms_per_solve": day_s * 1e3 / ((lmax + 1) * nfreq),
            "ml_tiles": {"certified_direct": n_direct, "eigen_decomposed": n_eigen, "null_certificate": (c1[b"ml_tiles_null"] - c0[b"ml_tiles_null"]) // max(args.steps, 1),
                         "ql_failed": c1[b"ml_tiles_ql_failed"] - c0[b"ml_tiles_ql_failed"],
                         "basis_route": (c1[b"ml_tiles_basis"] - c0[b"ml_tiles_basis"]) // max(args.steps, 1),
                         "rank_stopped": (c1[b"ml_tiles_stopped"] - c0[b"ml_tiles_stopped"]) // max(args.steps, 1),
                         "rank_stop_mean_order": (c1[b"ml_stop_cols"] - c0[b"ml_stop_cols"]) / max(c1[b"ml_tiles_stopped"] - c0[b"ml_tiles_stopped"], 1)} if kind == "ml" else None,
        },
        "roofline": roofline,
        "roofline_secondary": secondary,
        "kernel_classes_ms_per_day_timed": prof,
        "allocator": {"num_alloc_retries": int(mem1.get("num_alloc_retries", 0) - mem0.get("num_alloc_retries", 0)),
                      "reserved_peak_GB": mem1.get("reserved_bytes.all.peak", 0) / 1e9},
    }
    del task, eng, mt, ss, vis, weight, out_map
    return out


class Job:
    """One rank's share of the job, held as the product's own containers and task objects."""

    def __init__(self, cfg, rank, world, scaling, b_dtype, pool_freqs, seed=3003, overlap_sht=None):
        import torch

        from draco_amd import parallel
        from draco_amd import workloads as wl
        from draco_amd.analysis.mapmaker import DirtyMapMaker
        from draco_amd.analysis.transform import MModeTransform
        from draco_amd.core import containers
        from draco_amd.core.products import PoolCycledProvider, SyntheticProvider, TransitTelescope
        from draco_amd.device import Context

        self.torch = torch
        self.ctx = ctx = Context.get()
        self.cfg = cfg
        nra, lmax = cfg["nra"], cfg["lmax"]
        self.nra, self.lmax, self.nside = nra, lmax, cfg["nside"]
        # the telescope knows every frequency of the job; the data containers hold this rank's slab
        if scaling == "weak":
            self.nfreq_job = cfg["nfreq"] * world
            all_freqs = np.concatenate([wl.frequencies(cfg["nfreq"]) + 400.0 * r for r in range(world)])
            count, start = cfg["nfreq"], cfg["nfreq"] * rank
        else:
            self.nfreq_job = cfg["nfreq"]
            all_freqs = wl.frequencies(cfg["nfreq"])
            count, start = parallel.split_local(cfg["nfreq"], rank, world)
        self.nfreq = nfreq = count
        self.tel = tel = TransitTelescope(all_freqs, lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
        self.npairs = npairs = tel.npairs
        es = 16 if b_dtype == "complex128" else 8
        self.per_freq = per_freq = sum(2 * npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * es

        gen = torch.Generator(device=ctx.device).manual_seed(1000 + rank)
        vis = torch.randn((nfreq, npairs, nra), dtype=torch.complex64, device=ctx.device, generator=gen)
        weight = torch.rand((nfreq, npairs, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
        weight[torch.rand(weight.shape, dtype=torch.float32, device=ctx.device, generator=gen) < 0.01] = 0.0  # 1 % exact zeros
        self.ss = containers.SiderealStream(freq=all_freqs[start : start + count], ra=nra, stack=npairs, allocate=False)
        self.ss.attach("vis", vis)
        self.ss.attach("vis_weight", weight)

        if pool_freqs <= 0:
            free, _ = torch.cuda.mem_get_info(ctx.device)
            # per step: m-modes (24 B per (m, sign, f, base)), a_lm, maps, SHT scratch, + slack
            reserve = (lmax + 1) * 2 * nfreq * npairs * 24 + nfreq * 4 * (lmax + 1) ** 2 * 16 + nfreq * 4 * 12 * self.nside**2 * 8 + (10 << 30)
            pool_freqs = 1
            while pool_freqs * 2 <= nfreq and pool_freqs * 2 * per_freq <= (free - reserve) * 0.95:
                pool_freqs *= 2
            pool_freqs = min(pool_freqs, 32)
        while nfreq % pool_freqs:
            pool_freqs -= 1
        self.pool_freqs = pool_freqs
        self.ncycle = nfreq // pool_freqs
        self.pool_bytes = pool_freqs * per_freq
        base = SyntheticProvider(tel, seed=seed)
        # frequencies alias with period pool_freqs (hbm-pool policy).  Under strong scaling at N = 8 a rank's 32
        # frequencies are one period: every tile it solves against is distinct and stays resident across days.
        self.bt = PoolCycledProvider(base, pool_freqs)
        self.mt = MModeTransform()
        self.mt.setup(self.bt)
        self.dm = DirtyMapMaker(nside=self.nside, b_dtype=b_dtype, pool_bytes=self.pool_bytes + (1 << 20))
        self.dm.overlap_sht = overlap_sht
        self.dm.setup(self.bt)
        ntel = 2 * npairs
        ms = np.tile(np.arange(lmax + 1), pool_freqs)
        # algorithmic bytes of ONE dirty launch: B (l>=m) + v, Ni + a per tile
        self.dirty_bytes = self.pool_bytes + len(ms) * ntel * (16 + 8) + int(sum(4 * (lmax + 1 - int(m)) * 16 for m in ms))
        ctx.sync()

    def step(self):
        """One sidereal day through the task classes: SiderealStream -> MModes -> Map."""
        mm = self.mt.process(self.ss)
        return self.dm.process(mm)

    def to_alm(self):
        mm = self.mt.process(self.ss)
        return self.dm.make_alm(mm)

    def timed_launches(self, fn):
        """Run ``fn`` with HIP events around every Dirty launch (on the launch stream); mean launch ms, count."""
        eng = self.dm._get_engine()
        eng.launch_events = []
        fn()
        self.torch.cuda.synchronize()
        ev = eng.launch_events
        eng.launch_events = None
        ms = [a.elapsed_time(b) for a, b, _, _ in ev]
        return float(np.mean(ms)), len(ms)

    def stage_alone_ms(self):
        """HIP-event times of the step's stages, each alone on the GPU (T_fft, T_solve, T_sht)."""
        from draco_amd import _lib
        from draco_amd.device import ptr

        from draco_amd.analysis.mapmaker import _alm2map_neighbourly

        ctx, torch = self.ctx, self.torch
        maps = ctx.empty((self.nfreq, 4, 12 * self.nside**2), np.float64)

        def once():
            ctx.timer_start()
            mm = self.mt.process(self.ss)
            t_fft = ctx.timer_stop()
            ctx.timer_start()
            alm = self.dm.make_alm(mm)
            t_solve = ctx.timer_stop()
            ctx.timer_start()  # (the Legendre form the timed day runs: what the map-makers select around their own alm2map)
            _alm2map_neighbourly(ctx, alm, self.nfreq, self.lmax, self.lmax, self.nside, maps)
            t_sht = ctx.timer_stop()
            return t_fft, t_solve, t_sht

        torch.cuda.synchronize()
        once()  # untimed: the first call of a stage in the process loads its code object and builds its tables
        runs = np.array([once() for _ in range(3)])
        med = np.median(runs, axis=0)
        return {"T_fft": float(med[0]), "T_solve": float(med[1]), "T_sht": float(med[2]), "calls": "1 warm + median of 3",
                "T_sht_spread": float((runs[:, 2].max() - runs[:, 2].min()) / med[2])}


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun around it: THIS process becomes the launcher.  It has not
    touched the GPU (the CPU baseline below spawns CPU workers only; `torch.cuda.device_count()` does not initialise
    HIP), starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process --
    one rank per GPU over RCCL, the reference's `mpirun -np N` (.github/workflows/main.yaml:83-92) --, relays rank 0's
    JSON line with its own `cpu_baseline` merged in, and exits with the child's code."""
    import socket
    import subprocess

    import torch

    n = args.gpus
    ndev = torch.cuda.device_count()
    if ndev < n and not args.same_device:
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible (one rank per GPU over RCCL; --backend gloo --same-device rehearses the N > 1 code path on one GPU)")
    cpu = None
    if not args.no_cpu_baseline and args.maker == "dirty":
        from draco_amd import workloads as _wl

        cpu = cpu_baseline(_wl.CONFIGS[args.config], args.cpu_seconds)
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), DMM_BENCH_LAUNCHER="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln
        else:
            sys.stdout.write(ln)
    rc = proc.wait()
    if line is None:
        raise SystemExit(rc or f"bench.py: the {n}-rank child printed no result line")
    out = json.loads(line)  # rank 0's compact line; its full record is in bench_extra.json
    launcher = {"command": " ".join(cmd[1:6]) + " ... bench.py", "ranks_started": n, "devices_visible": ndev}
    full = None
    try:
        with open(os.path.join(ROOT, EXTRA_FILE)) as fh:
            full = json.load(fh)
        if full.get("n_gpus") != out.get("n_gpus") or abs(full.get("value", 0.0) - out["value"]) > 1e-4 * abs(out["value"]):
            full = None  # (a stale file of another run)
    except Exception:
        full = None
    if full is not None:
        full["cpu_baseline"], full["launcher"] = cpu, launcher
        emit(full)
    else:
        out["cpu_baseline"] = compact_record({"cpu_baseline": cpu})["cpu_baseline"]
        out["launcher"] = {"ranks_started": n, "devices_visible": ndev}
        print(json.dumps(out), flush=True)
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: the launcher's world size is what runs", file=sys.stderr)
    if args.scaling is None:
        # N > 1: the metric's own job -- (128-feed, 256-freq) -- split over the ranks; at N = 8 a rank's 32 frequencies
        # are exactly the resident pool.  N = 1: both forms are the same job.
        args.scaling = "strong" if world > 1 else "weak"
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from draco_amd import workloads as _wl  # (the oracle is imported inside cpu_baseline only)

        if args.maker == "dirty":
            cpu = cpu_baseline(_wl.CONFIGS[args.config], args.cpu_seconds)  # before any GPU work: it spawns processes
        else:
            cpu = cpu_baseline_dense(_wl.CONFIGS[args.config], args.maker, max(args.cpu_seconds, 24.0))
    if args.maker != "dirty":
        if world != 1:
            raise SystemExit("--maker ml / wiener is a single-GPU measurement (frequencies are independent: shard them as the Dirty job does)")
        return main_dense(args, cpu)

    import torch
    import torch.distributed as dist

    if world > 1:
        if args.same_device:
            local = 0
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)

    def allreduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ranks_rec = None
    if world > 1:
        # who is really there: every rank contributes (rank, local device index) through the SAME kind of collective the
        # map gather uses (`all_gather_into_tensor`: RCCL under nccl), so the record says how many ranks RCCL saw
        mine = torch.tensor([rank, torch.cuda.current_device()], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        seen = torch.empty(world * 2, dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(seen, mine)
        seen = seen.view(world, 2).cpu().tolist()
        ranks_rec = {"world_size": world, "ranks_seen_by_all_gather_into_tensor": len({r for r, _ in seen}), "devices": [d for _, d in seen],
                     "distinct_devices": len({d for _, d in seen}), "backend": args.backend + (" (RCCL)" if args.backend == "nccl" else " (rehearsal)")}

    from draco_amd import parallel
    from draco_amd import workloads as wl
    from draco_amd.analysis import _solve

    cfg = wl.CONFIGS[args.config]
    if os.environ.get("DMM_OPTS"):  # A/B switches of the library ("name=value,..."; include/draco_amd.h), e.g. DMM_OPTS=dirty_prio=1
        from draco_amd import _lib
        from draco_amd.device import Context

        for kv in os.environ["DMM_OPTS"].split(","):
            k_, v_ = kv.split("=")
            for c_ in (Context.get(), Context.side(Context.get().device_index)):
                _lib.check(_lib.lib.dmm_ctx_set_option(c_.handle, k_.strip().encode(), int(v_)))
                if k_.strip() == "sht_variant":
                    c_.sht_variant_pin = int(v_)  # (the map-makers then leave the Legendre form alone)
    job = Job(cfg, rank, world, args.scaling, args.b_dtype, args.pool_freqs, overlap_sht=False if args.no_overlap else None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1) if args.steps else args.warmup):
        job.step()  # (the first pass fills the pool: at least one untimed pass so that B is resident when the clock starts)
    # reference points outside the timed region: the stages each alone on the GPU
    alone_ms, _ = job.timed_launches(job.to_alm)
    stage_ms = job.stage_alone_ms()
    fills_before = job.dm._get_engine().fills
    eng = job.dm._get_engine()
    eng.launch_events = []
    barrier()
    mem0 = torch.cuda.memory_stats()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    day_issue_ms = []
    for _ in range(args.steps):
        out_map = job.step()
        day_issue_ms.append((time.perf_counter() - t0) * 1e3)  # host clock when the day's launches were all queued
    barrier()
    elapsed = time.perf_counter() - t0
    mem1 = torch.cuda.memory_stats()
    free_now, total_hbm = torch.cuda.mem_get_info()
    allocator = {
        # what the caching allocator did INSIDE the timed region: a retry = it ran out, synchronised, freed its cache
        # and asked the driver again (the GPU idles meanwhile); device_alloc = hipMalloc calls (each one synchronises)
        "num_alloc_retries": int(mem1.get("num_alloc_retries", 0) - mem0.get("num_alloc_retries", 0)),
        "num_ooms": int(mem1.get("num_ooms", 0) - mem0.get("num_ooms", 0)),
        "num_device_alloc": int(mem1.get("num_device_alloc", 0) - mem0.get("num_device_alloc", 0)),
        "num_device_free": int(mem1.get("num_device_free", 0) - mem0.get("num_device_free", 0)),
        "reserved_peak_GB": mem1.get("reserved_bytes.all.peak", 0) / 1e9,
        "allocated_peak_GB": mem1.get("allocated_bytes.all.peak", 0) / 1e9,
        "hbm_total_GB": total_hbm / 1e9,
        "host_issue_ms_per_day": [round(b - a, 2) for a, b in zip([0.0] + day_issue_ms[:-1], day_issue_ms)],
    }
    ev = eng.launch_events
    eng.launch_events = None
    assert eng.fills == fills_before, "B was uploaded inside the timed region"
    # a retry of the caching allocator inside the timed region = the GPU idled while torch freed and re-allocated its
    # cache (what unbounded host run-ahead did to round 2's 20-step line): such a number is not the product's
    assert allocator["num_alloc_retries"] == 0 and allocator["num_ooms"] == 0, f"caching allocator retried inside the timed region: {allocator}"
    if world > 1:
        elapsed = allreduce_max(elapsed)
    launch_ms = [a.elapsed_time(b) for a, b, _, _ in ev]
    dirty_avg_ms = float(np.mean(launch_ms))
    nlaunch = len(launch_ms)

    lmax = cfg["lmax"]
    ms_per_step = elapsed / args.steps * 1e3
    scale = world if args.scaling == "weak" else 1
    value = scale * (lmax + 1) / (elapsed / args.steps)
    achieved = job.dirty_bytes / (dirty_avg_ms * 1e-3) / 1e9

    # the north star's one collective, after the timed region: all-gather of the rank-local Maps over RCCL.
    # It must never cost the headline: (1) rank 0 leaves a copy of the headline on stderr before the first collective
    # of the measurement, (2) every rank first checks ALONE that it can hold the gathered map (the one realistic
    # failure: N x 6.4 GB next to everything else) and the ranks AGREE on going ahead with one all-reduce, so that no
    # rank enters a collective the others skip, (3) what fails after that is recorded and agreed on the same way.
    gather = None
    if world > 1 and not args.no_allgather:
        if rank == 0:
            print("headline before the all-gather measurement: " + json.dumps({"value": value, "ms_per_step": ms_per_step, "n_gpus": world, "steps": args.steps}), file=sys.stderr, flush=True)

        def all_ok(ok):
            t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item() > 0.5)

        err = None
        try:
            _solve.release_pools()  # the gathered map of a weak-scaled job is N x 6.4 GB: make room first
            shard = out_map.map._dev
            shard_bytes = shard.numel() * 8
            probe = torch.empty((world * shard.shape[0], *shard.shape[1:]), dtype=shard.dtype, device=shard.device)
            del probe  # (stays in the caching allocator: the gather's own allocation of this size will be served from it)
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"[:400]
        if not all_ok(err is None):
            gather = {"skipped": err or "another rank could not hold the gathered map", "backend": args.backend}
        else:
            ts, nfull = [], None
            try:
                for _ in range(2):
                    barrier()
                    t0 = time.perf_counter()
                    full = parallel.allgather_map(out_map)
                    barrier()
                    ts.append(time.perf_counter() - t0)
                    nfull = len(full.index_map["freq"])
                    del full
            except Exception as exc:  # noqa: BLE001  (a failure INSIDE a collective cannot be recovered by the other ranks;
                err = f"{type(exc).__name__}: {exc}"[:400]  # the stderr copy above is what survives then)
            if not all_ok(err is None) or not ts:
                gather = {"error": err or "another rank failed", "backend": args.backend}
            else:
                tg = allreduce_max(min(ts))
                gather = {
                    "allgather_ms": tg * 1e3,
                    "shard_GB": shard_bytes / 1e9,
                    "gathered_GB": shard_bytes * world / 1e9,
                    "frequencies_gathered": nfull,
                    # every rank receives (N-1) shards; in a direct all-gather each arrives over its own xGMI link
                    "GBs_per_link": shard_bytes / tg / 1e9,
                    "GBs_per_gpu_in": shard_bytes * (world - 1) / tg / 1e9,
                    "backend": args.backend,
                    "note": "parallel.allgather_map (one all_gather_into_tensor; RCCL over xGMI under nccl) on the maps of the last timed day, outside the timed region; best of 2",
                }

    traffic = None
    try:  # HBM bytes per launch from the committed PMC profile of this same command, if it matches
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["k_dirty"]
        if rec["config"] == args.config and rec["b_dtype"] == args.b_dtype and rec["pool_freqs"] == job.pool_freqs:
            traffic = rec["hbm_bytes_per_launch"]
    except Exception:
        traffic = None

    nfreq_rank = job.nfreq
    out = {
        "metric": "m-modes/sec through MModeTransform+DirtyMapMaker (128-feed, 256-freq)",
        "value": value,
        "unit": "m-modes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64 accumulate; B stored " + args.b_dtype + "; FFT complex64 (as the reference)",
        "data": "synthetic",
        "config": {
            "workload": f"cfg{args.config}: {job.tel.nfeed}-feed ({job.npairs} stacked baselines), {nfreq_rank} freq per GPU ({job.nfreq_job} in the job), {cfg['nra']} RA, lmax=mmax={lmax}: MModeTransform.process + DirtyMapMaker.process through the task classes ({(lmax+1)*nfreq_rank} (m,f) solves + alm2map to nside={cfg['nside']} IQUV maps per GPU and day)",
            "b_residency": f"hbm-pool: {job.pool_freqs} frequencies' B tiles resident ({job.pool_bytes/1e9:.1f} GB distinct, {args.b_dtype}, l>=m packed), provider aliases f -> f % {job.pool_freqs}: {job.ncycle} slab(s) per day, filled once, resident across days",
            "solves_per_s": (lmax + 1) * job.nfreq_job / (elapsed / args.steps),
            "parallelism": f"freq-sharded x{world}, {args.scaling} scaling (no collective in the timed region)",
            "alm2map": ("caller's stream, per solved slab, between the slabs' solves" if (args.no_overlap or args.b_dtype == "complex64") else "side stream, per solved slab, beside the next slab's solves") + " (BaseMapMaker.process; default by B storage type, DirtyMapMaker.overlap_sht)",
        },
        # SURVEY 8d asks for the stage times next to the metric; each measured alone on the GPU after warmup (in the
        # step the SHT runs beside the solves).  value_to_alm = (mmax+1) / (T_fft + T_solve): the metric with T ending
        # at "a_lm of all (m,f) resident", i.e. without DirtyMapMaker's final alm2map
        "stages_alone_ms": stage_ms,
        "value_to_alm": scale * (lmax + 1) / ((stage_ms["T_fft"] + stage_ms["T_solve"]) * 1e-3),
        "roofline": {
            "kernel": "k_dirty (a = B^H N^-1 v, batched over (m,f))",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "bytes_per_launch": job.dirty_bytes,
            "avg_launch_ms": dirty_avg_ms,
            "launches": nlaunch,
            "alone": {
                "avg_launch_ms": alone_ms,
                "achieved": job.dirty_bytes / (alone_ms * 1e-3) / 1e9,
                "frac": job.dirty_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "same kernel, same launches through DirtyMapMaker.make_alm, no concurrent alm2map (untimed reference pass after warmup)",
            },
        },
    }
    out["allocator"] = allocator
    if ranks_rec is not None:
        out["ranks"] = ranks_rec
    if gather is not None:
        out["allgather"] = gather

    if rank == 0 and world == 1 and not args.no_extra:
        out["extra"] = extras(args, cfg, job)

    if rank == 0:
        out["cpu_baseline"] = cpu
        emit(out)
    if world > 1:
        dist.destroy_process_group()


def extras(args, cfg, job):
    """Secondary measurements (single GPU, after the headline): never allowed to break the headline line."""
    import torch

    from draco_amd import _lib
    from draco_amd import workloads as wl
    from draco_amd.analysis import _solve
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.analysis.transform import mmode_forward
    from draco_amd.core import containers
    from draco_amd.core.hoststage import HostStager
    from draco_amd.core.products import PackedStoreProvider, PoolCycledProvider, SyntheticProvider, TransitTelescope
    from draco_amd.device import Context, ptr

    extra = {}
    ctx = Context.get()
    nfreq, lmax, nside, nra = cfg["nfreq"], cfg["lmax"], cfg["nside"], cfg["nra"]
    try:
        # (1) the same day driven through the raw C ABI (no task objects): what the Python layer costs
        eng = job.dm._get_engine()
        mv, mw = mmode_forward(ctx, job.ss.vis.device(ctx), job.ss.weight.device(ctx), lmax)
        alm = torch.empty((nfreq, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mv, mw = mmode_forward(ctx, job.ss.vis.device(ctx), job.ss.weight.device(ctx), lmax)
            for s_ in eng.slabs(list(range(nfreq)), lmax, nfreq, lmax + 1):
                _lib.check(_lib.lib.dmm_dirty_run(s_.plan, ptr(s_.pool), ptr(mv), ptr(mw), ptr(alm)))
            torch.cuda.synchronize()
            t_raw = time.perf_counter() - t0
        extra["raw_abi_to_alm"] = {"value": (lmax + 1) / t_raw, "unit": "m-modes/s", "ms": t_raw * 1e3, "note": "dmm_mfft_pack + dmm_mmode_weight + one dmm_dirty_run per slab, no alm2map"}
        del alm, mv, mw, s_
        # (2) complex64 storage of B (half the bytes, float64 accumulation), same task classes
        if args.b_dtype == "complex128":
            _solve.release_pools()
            pf = job.pool_freqs
            dm64 = DirtyMapMaker(nside=nside, b_dtype="complex64", pool_bytes=job.pool_bytes // 2 + (1 << 20))
            dm64.setup(PoolCycledProvider(SyntheticProvider(job.tel, seed=3003), pf))
            mm = job.mt.process(job.ss)
            dm64.process(mm)
            e64 = dm64._get_engine()
            e64.launch_events = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dm64.process(job.mt.process(job.ss))
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            lm = [a.elapsed_time(b) for a, b, _, _ in e64.launch_events]
            e64.launch_events = None
            b64 = job.dirty_bytes - job.pool_bytes // 2
            extra["b_complex64"] = {"value": (lmax + 1) / (el / args.steps), "unit": "m-modes/s", "roofline_GBs": b64 / (np.mean(lm) * 1e-3) / 1e9, "frac": b64 / (np.mean(lm) * 1e-3) / 1e9 / HBM_PEAK_GBS, "pool_freqs": pf}
            del dm64, e64, mm
            _solve.release_pools()
        # (3) Wiener / ML samples at cfg-3 tile sizes
        tel = TransitTelescope(wl.frequencies(nfreq), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
        gen = torch.Generator(device=ctx.device).manual_seed(7)
        eng2 = SolveEngine(SyntheticProvider(tel, seed=5), ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, cache=True)
        nf_w = min(16, nfreq)  # (a 4-frequency sample sits on the launch and QL latency floors of the dense solvers: 0.062 / 0.17 ms)
        vis1 = torch.randn((nf_w, tel.npairs, nra), dtype=torch.complex64, device=ctx.device, generator=gen)
        w1 = torch.rand((nf_w, tel.npairs, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
        mv1, mw1 = mmode_forward(ctx, vis1, w1, lmax)
        for kind in ("wiener", "ml"):
            for _ in range(2):
                ctx.sync()
                t0 = time.perf_counter()
                eng2.solve(kind, mv1, mw1, list(range(nf_w)), lmax, prior_amp=1.0, prior_tilt=0.5)
                ctx.sync()
                t_w = time.perf_counter() - t0
            extra[f"{kind}_ms_per_solve"] = t_w * 1e3 / (nf_w * (lmax + 1))
            if kind == "wiener":
                # FP64 flops of the sample: Hermitian Gram of the smaller side 8 k^2 K / 2 + Cholesky (8/3) k^3
                ntel_ = 2 * tel.npairs
                fl = 0.0
                for m_ in range(lmax + 1):
                    nsky_ = 4 * (lmax + 1 - m_)
                    k_, K_ = min(ntel_, nsky_), max(ntel_, nsky_)
                    fl += 4.0 * k_ * k_ * K_ + (8.0 / 3.0) * k_**3
                extra["wiener_mfma"] = {"achieved": fl * nf_w / t_w / 1e12, "peak": 78.6, "unit": "TFLOP/s (FP64 MFMA, whole solve incl. factorisation and triangular solves)",
                                        "frac": fl * nf_w / t_w / 1e12 / 78.6}
        # ML with every tile eigen-decomposed ("ml_shortcut" = 2): what ill-conditioned beam transfers cost, where the
        # full-rank certificate of the sample above does not pass
        try:
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
            ctx.sync()
            t0 = time.perf_counter()
            eng2.solve("ml", mv1, mw1, list(range(nf_w)), lmax)
            ctx.sync()
            extra["ml_eigen_ms_per_solve"] = (time.perf_counter() - t0) * 1e3 / (nf_w * (lmax + 1))
        finally:
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
        extra["dense_sample"] = f"all {lmax + 1} m of {nf_w} frequencies, B resident"
        del eng2, mv1, mw1, vis1, w1
        _solve.release_pools()
        # (3b) the same two makers on PHYSICALLY STRUCTURED tiles through the task classes, as `bench.py --maker wiener|ml
        # --freqs 32` reports them: a 32-frequency sample of the cfg-3 day (scaled to the day, frequencies are
        # independent) with the SAME roofline objects -- Wiener: Gram + Cholesky over the span of the pass against the
        # FP64 MFMA peak; ML: the Gram kernel's MFMA fraction, stage 1 of the reduction against the HBM peak, the tile
        # counters (none of these tiles passes the full-rank certificate: every one is eigen-decomposed)
        import copy

        # ("ml_day_gram_resident": the ML day again with the beam Gram products B B^H of the resident tiles kept beside the B
        # block -- multi-day processing, `MaximumLikelihoodMapMaker.cache_beam_gram`; a labelled mode: the warm-up day fills them)
        # ("ml_day_basis_resident": with the singular bases of the resident beam transfers kept instead -- `cache_beam_basis`)
        modes = [("wiener", "wiener_day", 0), ("ml", "ml_day", 0)]
        if getattr(args, "extra_resident", False):
            modes += [("ml", "ml_day_gram_resident", 1), ("wiener", "wiener_day_gram_resident", 1), ("ml", "ml_day_basis_resident", 2)]
        for kind, key, resident in modes:
            try:
                a2 = copy.copy(args)
                a2.maker, a2.tiles, a2.freqs, a2.pool_freqs, a2.steps, a2.warmup, a2.b_dtype = kind, "screen", min(32, nfreq), 16, 1, 1, "complex128"
                a2.gram_resident = resident == 1
                a2.basis_resident = resident == 2
                rec = dense_day(a2, kind)
                extra[key] = {k: rec[k] for k in ("metric", "value", "unit", "ms_per_step", "day_scale", "config", "roofline", "roofline_secondary", "kernel_classes_ms_per_day_timed", "allocator")}
            except Exception as e:  # noqa: BLE001
                extra[key] = {"error": repr(e)[:300]}
            _solve.release_pools()
        # (4) B = host-stream (SURVEY 8d's second residency policy) THROUGH DirtyMapMaker.process: the tiles of a few
        # frequencies live in pinned host memory in the pool's wire format, nothing is resident on the GPU beforehand;
        # uploads of slab k+1 run under the solves of slab k (two buffers)
        nf_h = min(4, nfreq)
        tel_h = TransitTelescope(wl.frequencies(nf_h), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
        shape = (lmax + 1, 2, nf_h, tel_h.npairs)
        mm_h = containers.MModes(mmax=lmax, freq=tel_h.frequencies, stack=tel_h.npairs, allocate=False)
        mm_h.attach("vis", torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen))
        mm_h.attach("vis_weight", torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5)
        mm_more = []
        for _ in range(15):
            mm_d = containers.MModes(mmax=lmax, freq=tel_h.frequencies, stack=tel_h.npairs, allocate=False)
            mm_d.attach("vis", torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen))
            mm_d.attach("vis_weight", torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5)
            mm_more.append(mm_d)
        hs = {}
        for b_dtype, npdt in (("complex128", np.complex128), ("complex64", np.complex64)):
            store = PackedStoreProvider.from_provider(SyntheticProvider(tel_h, seed=9), ctx, npdt, pin=True)
            per_f = store.per_freq * np.dtype(npdt).itemsize
            t_ = DirtyMapMaker(nside=64, b_dtype=b_dtype, pool_bytes=int(2 * 1.05 * per_f))  # two buffers of one frequency
            t_.setup(store)
            best = None
            for _ in range(2):
                _solve.release_pools()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                t_.process(mm_h)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            nb = t_._get_engine().last_b_bytes
            hs[b_dtype] = {"value": (lmax + 1) / (best * nfreq / nf_h), "unit": "m-modes/s", "h2d_GBs": nb / best / 1e9, "seconds": best, "b_GB": nb / 1e9}
            # D days per PCIe crossing of B (BaseMapMaker.process_many): per day-equivalent
            for D in (4, 16):
                days = [mm_h] + [mm_more[d] for d in range(D - 1)]
                _solve.release_pools()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                mps = t_.process_many(days)
                mps[-1].map._dev
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                hs[b_dtype][f"process_many_D{D}"] = {"value": (lmax + 1) * D / (dt * nfreq / nf_h), "unit": "m-modes/s per day-equivalent", "seconds": dt,
                                                      "h2d_GBs": t_._get_engine().last_b_bytes / dt / 1e9}
                del mps, days
            del store, t_
        hs["note"] = f"DirtyMapMaker.process with a PackedStoreProvider over pinned host memory ({nf_h} frequencies' cfg-3 tiles, every byte crosses PCIe, double-buffered under the solves), scaled to the {nfreq}-frequency day; process_many_D*: D days share one crossing (each day's a_lm bit-identical to its own pass, tests/test_gpu_process_many.py)"
        extra["b_host_stream"] = hs
        del mm_more, mm_h
        HostStager.release()
        _solve.release_pools()
        # (5) B resident, D days per READ of B: MModeTransform.process per day + DirtyMapMaker.process_many (the Dirty
        # kernel keeps 8 accumulators per column: 8 days per pass over a slab), 16 frequencies' tiles resident
        torch.cuda.empty_cache()
        md = {}
        pf = 16
        dmm_ = DirtyMapMaker(nside=nside, pool_bytes=pf * job.per_freq + (1 << 20))
        bt16 = PoolCycledProvider(SyntheticProvider(job.tel, seed=3003), pf)
        dmm_.setup(bt16)
        streams = [job.ss]
        for d in range(7):
            s_ = containers.SiderealStream(freq=job.tel.frequencies, ra=nra, stack=job.npairs, allocate=False)
            s_.attach("vis", torch.randn((nfreq, job.npairs, nra), dtype=torch.complex64, device=ctx.device, generator=gen))
            s_.attach("vis_weight", torch.rand((nfreq, job.npairs, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5)
            streams.append(s_)
        for D in (1, 8):
            dmm_.process_many([job.mt.process(s_) for s_ in streams[:D]])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                mps = dmm_.process_many([job.mt.process(s_) for s_ in streams[:D]])
            mps[-1].map._dev
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            md[f"D{D}"] = {"value": (lmax + 1) * D / dt, "unit": "m-modes/s per day-equivalent", "seconds_per_group": dt}
            del mps
        md["note"] = f"B resident ({pf} frequencies' distinct tiles), {nfreq}-frequency days: MModeTransform.process per day + DirtyMapMaker.process_many (dmm_dirty_run_multi: D days per read of every tile; alm2map of every day included)"
        extra["many_days"] = md
        del streams, dmm_
        _solve.release_pools()
    except Exception as e:  # secondary numbers must never break the headline line
        import traceback

        extra["error"] = repr(e)
        extra["traceback"] = traceback.format_exc()[-1500:]
    return extra


if __name__ == "__main__":
    main()
