#!/usr/bin/env python
"""B = host-stream measured through DirtyMapMaker.process (SURVEY 8d's second residency policy).

cfg-3 tiles (758 x 4(513-m), packed) of `nf` frequencies live in HOST memory in the pool's wire format
(PackedStoreProvider); DirtyMapMaker.process uploads them slab by slab over PCIe under the previous slab's solves.
Arms: pinned store (copy engine reads it directly), pageable store (worker threads memcpy it through the pinned
staging ring), per-tile provider (only `beam_m`, packs tile by tile like a driftscan BeamTransfer would), complex64
wire format.  Every arm's a_lm must equal the device-generated pool's bit for bit.

    python tools/host_stream.py [nf=8] [config=3]
"""

import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class PerTileFromStore:
    """Picklable factory for `PackedStoreProvider.pack`'s spawned workers: a provider that only has per-tile `beam_m`
    (what a driftscan BeamTransfer offers), served from a memory-mapped store."""

    def __init__(self, tel, path):
        self.tel, self.path = tel, path

    def __call__(self):
        from draco_amd.core.products import ArrayProvider, PackedStoreProvider

        st = PackedStoreProvider.open(self.tel, self.path)
        return ArrayProvider(self.tel, lambda m, f: st.beam_m(m, fi=f))


def main():
    import torch

    from draco_amd import workloads as wl
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.hoststage import HostStager
    from draco_amd.core.products import ArrayProvider, PackedStoreProvider, SyntheticProvider, TransitTelescope
    from draco_amd.device import Context

    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cfg = wl.CONFIGS[int(sys.argv[2]) if len(sys.argv) > 2 else 3]
    lmax = cfg["lmax"]
    ctx = Context.get()
    tel = TransitTelescope(wl.frequencies(nf), lmax=lmax, ncyl=cfg["ncyl"], nfeed_cyl=cfg["nfeed_cyl"])
    syn = SyntheticProvider(tel, seed=3003)
    gen = torch.Generator(device=ctx.device).manual_seed(5)
    shape = (lmax + 1, 2, nf, tel.npairs)
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
    mm.attach("vis", torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen))
    mm.attach("vis_weight", torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5)
    out = {"config": f"cfg-3 tiles, {nf} frequencies, {tel.npairs} baselines, lmax {lmax}", "arms": {}}
    nside = 64  # the SHT is not what is measured here

    def run(provider, b_dtype, pool_bytes, label, reps=2):
        t = DirtyMapMaker(nside=nside, b_dtype=b_dtype, pool_bytes=pool_bytes)
        t.setup(provider)
        best = None
        for _ in range(reps):
            _solve.release_pools()  # nothing resident: every byte crosses PCIe
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            alm = t.make_alm(mm)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        nbytes = t._engine.last_b_bytes
        rec = {"seconds": best, "b_GB": nbytes / 1e9, "GBs": nbytes / best / 1e9, "slabs": t._engine.fills // reps,
               "m_modes_per_s_cfg3_day": (lmax + 1) / (best * cfg["nfreq"] / nf)}
        out["arms"][label] = rec
        print(label, json.dumps(rec), flush=True)
        return alm.cpu().numpy()

    for b_dtype, npdt in (("complex128", np.complex128), ("complex64", np.complex64)):
        es = np.dtype(npdt).itemsize
        per_freq = PackedStoreProvider.elements(tel) // nf * es
        ref = run(syn, b_dtype, None, f"device-generated {b_dtype} (reference, no PCIe)", reps=1)
        store = PackedStoreProvider.from_provider(syn, ctx, npdt, pin=True)
        two_freq = int(2 * 2.05 * per_freq)  # two buffers of two frequencies each
        a = run(store, b_dtype, two_freq, f"pinned store {b_dtype}")
        assert np.array_equal(a, ref), "pinned store: a_lm differs"
        if b_dtype == "complex128":
            pageable = PackedStoreProvider(tel, np.array(store.store), pinned=False)  # an ordinary (pageable) copy
            a = run(pageable, b_dtype, two_freq, f"pageable store {b_dtype} (staged, {HostStager.get(ctx.device).workers} threads)")
            assert np.array_equal(a, ref), "pageable store: a_lm differs"
            st = pageable.store
            per_tile = ArrayProvider(tel, lambda m, f, p=pageable: p.beam_m(m, fi=f))
            nf_small = ArrayProvider  # noqa: F841
            a = run(per_tile, b_dtype, two_freq, "per-tile beam_m provider complex128 (generic pack)", reps=1)
            assert np.array_equal(a, ref), "per-tile provider: a_lm differs"
            # the same per-tile provider packed ONCE by worker processes into a memory-mapped .npy store: what a real
            # driftscan product pays on the first day, and what every later day streams at
            import tempfile

            tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
            with tempfile.TemporaryDirectory(dir=tmpdir) as d:
                path = os.path.join(d, "b_packed.npy")
                # this process has run GPU passes: pack() spawns its workers (never forks a process that holds a HIP
                # context), so they get a picklable factory -- the per-tile provider over a memory-mapped copy
                src = os.path.join(d, "b_source.npy")
                np.save(src, pageable.store)
                t0 = time.perf_counter()
                packed = PackedStoreProvider.pack(per_tile, path, factory=PerTileFromStore(tel, src))
                t_pack = time.perf_counter() - t0
                nb = packed.store.size * packed.store.itemsize
                out["arms"]["per-tile provider: one-time pack"] = {"seconds": t_pack, "b_GB": nb / 1e9, "GBs": nb / t_pack / 1e9, "where": d}
                print("per-tile provider: one-time pack", json.dumps(out["arms"]["per-tile provider: one-time pack"]), flush=True)
                a = run(packed, b_dtype, two_freq, "per-tile provider, second day: packed memory-mapped store (staged)")
                assert np.array_equal(a, ref), "packed store: a_lm differs"
                del packed
            del pageable, per_tile, st
        del store
        HostStager.release()
    # raw pinned copy rate for comparison
    hb = torch.empty(1 << 32, dtype=torch.uint8).pin_memory()
    dv = torch.empty(1 << 32, dtype=torch.uint8, device=ctx.device)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dv.copy_(hb, non_blocking=True)
        torch.cuda.synchronize()
        raw = (1 << 32) / (time.perf_counter() - t0) / 1e9
    out["raw_pinned_h2d_GBs"] = raw
    print(json.dumps(out))


if __name__ == "__main__":
    main()
