"""N>1 path on CPU: world_size-2 gloo run of the frequency sharding + map all-gather.

No GPU here, so each rank computes its slab with the ORACLE (test infrastructure); what is
under test is draco_amd.parallel (slab arithmetic, container sharding, the collective).
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from draco_amd import parallel
from draco_amd.core import containers


def test_split_local_matches_mpi_rule():
    for n in (1, 5, 8, 256, 257):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                c, s = parallel.split_local(n, r, world)
                cover += list(range(s, s + c))
                assert c in (n // world, n // world + 1)
            assert cover == list(range(n))


def test_shard_freq_containers():
    ss = containers.SiderealStream(freq=np.arange(5) + 400.0, ra=8, stack=3)
    ss.vis[:] = np.arange(5 * 3 * 8).reshape(5, 3, 8)
    a, b = parallel.shard_freq(ss, 0, 2), parallel.shard_freq(ss, 1, 2)
    assert a.vis.shape == (3, 3, 8) and b.vis.shape == (2, 3, 8)
    assert np.array_equal(np.concatenate([a.vis[:], b.vis[:]]), ss.vis[:])
    assert np.array_equal(b.index_map["freq"]["centre"], [403.0, 404.0])
    mm = containers.MModes(mmax=4, oddra=True, freq=np.arange(5) + 400.0, stack=3)
    mm.vis[:] = np.random.default_rng(0).standard_normal((5, 2, 5, 3))
    b = parallel.shard_freq(mm, 1, 2)
    assert b.vis.shape == (5, 2, 2, 3) and b.oddra and np.array_equal(b.vis[:], mm.vis[:][:, :, 3:])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nfreq, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import mapmaker as omm
        from oracle import sht as osht
        from oracle import synth as osyn

        lmax, nside, npairs = 6, 4, 3
        freqs = osyn.frequencies(nfreq)
        rng = np.random.default_rng(7)  # same full data set on every rank, then sharded
        mv = rng.standard_normal((lmax + 1, 2, nfreq, npairs)) + 1j * rng.standard_normal((lmax + 1, 2, nfreq, npairs))
        mw = rng.uniform(0.5, 1.5, mv.shape)
        full = containers.MModes(mmax=lmax, freq=freqs, stack=npairs)
        full.vis[:] = mv
        full.weight[:] = mw
        local = parallel.shard_freq(full)  # rank / world from the process group
        count, start = parallel.split_local(nfreq, rank, world)
        assert local.vis.shape[2] == count
        # the per-frequency pipeline on the local slab (oracle stands in for the kernels)
        find = omm.find_keys(freqs, local.index_map["freq"]["centre"], require_match=True)
        assert find == list(range(start, start + count))
        alm = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(5, m, f, npairs, 4, lmax), local.vis[:], local.weight[:], lmax, lmax, find)
        lm = containers.Map(nside=nside, axes_from=local)
        lm.map[:] = osht.sphtrans_inv_sky(alm, nside)
        g = parallel.allgather_map(lm, nfreq_total=nfreq)
        if rank == 0:
            alm_all = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(5, m, f, npairs, 4, lmax), mv, mw, lmax, lmax, list(range(nfreq)))
            ref = osht.sphtrans_inv_sky(alm_all, nside)
            ok = g.map.shape == ref.shape and np.array_equal(g.map[:], ref) and np.array_equal(g.index_map["freq"]["centre"], freqs)
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nfreq", [4, 5])
def test_two_rank_gloo_allgather(nfreq):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nfreq, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _xchg_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_m, nfreq, nbase = 5, 7, 3
        rng = np.random.default_rng(1)
        vis = rng.standard_normal((n_m, 2, nfreq, nbase)) + 1j * rng.standard_normal((n_m, 2, nfreq, nbase))
        wgt = rng.uniform(size=vis.shape)
        c, s0 = parallel.split_local(nfreq, rank, world)
        mv = torch.from_numpy(np.ascontiguousarray(vis[:, :, s0 : s0 + c]))
        mw = torch.from_numpy(np.ascontiguousarray(wgt[:, :, s0 : s0 + c]))
        a, b, lay = parallel.freq_to_m(mv, mw)
        mc, ms = parallel.split_local(n_m, rank, world)
        ok = np.array_equal(a.numpy(), vis[ms : ms + mc]) and np.array_equal(b.numpy(), wgt[ms : ms + mc])
        back = parallel.m_to_freq(a * 2.0, lay)
        ok = ok and np.array_equal(back.numpy(), 2.0 * vis[:, :, s0 : s0 + c])
        spec = parallel.gather_m(torch.from_numpy(np.arange(ms, ms + mc, dtype=np.float64)[:, None] * np.ones((1, 4))), lay)
        ok = ok and np.array_equal(spec.numpy(), np.arange(n_m, dtype=np.float64)[:, None] * np.ones((1, 4)))
        ok = ok and parallel.allreduce_max(float(rank) + 0.5) == world - 0.5
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_freq_m_exchange_world2():
    """The SVD filter's frequency <-> m exchange (uneven slabs on both axes), gather and MAX-reduce."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xchg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
