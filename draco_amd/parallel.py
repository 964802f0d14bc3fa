"""Frequency sharding across GPUs: one process per GPU; RCCL for the final map all-gather and, for the
SVD filter alone, the frequency <-> m exchange.

Frequency is the path's natural parallel axis (the reference distributes it with MPI,
``containers.py:505-506``): the m-FFT, every (m, f) solve and the SHT are independent per
frequency.  The reference additionally transposes frequency <-> m around the solves
(``mapmaker.py:62-67,99``; ``stream.py:96,119``) only because driftscan stores B per m on
disk; with the provider's bulk hand-over none of those all-to-alls exist here.  The single
collective is the optional gather of the output ``Map`` (the reference never gathers it,
``mapmaker.py:113-116``): :func:`allgather_map` -- one ``all_gather`` over
``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in CPU tests).
"""

from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .core import containers


def split_local(n: int, rank: int, world: int) -> tuple[int, int]:
    """``(count, start)`` of this rank's contiguous share of ``n`` items.

    Same rule as caput ``mpitools.split_local`` [3P] used at ``stream.py:73``: the first
    ``n % world`` ranks get one extra item.
    """
    base, rem = divmod(int(n), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return count, start


def _rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_freq(cont, rank=None, world=None):
    """Return a container of the same type holding only this rank's frequency slab.

    Works for :class:`SiderealStream`, :class:`MModes` and :class:`Map`; datasets stay
    where they are (host slices / device views).
    """
    r, w = _rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    nfreq = len(cont.index_map["freq"])
    count, start = split_local(nfreq, rank, world)
    sl = slice(start, start + count)
    kwargs = {"freq": cont.index_map["freq"][sl], "axes_from": cont, "attrs_from": cont, "allocate": False}
    if isinstance(cont, containers.MContainer):
        out = type(cont)(mmax=cont.mmax, oddra=cont.oddra, **kwargs)
    elif isinstance(cont, containers.Map):
        out = type(cont)(pol=cont.index_map["pol"], **kwargs)
    else:
        out = type(cont)(**kwargs)
    for name, ds in cont.datasets.items():
        fax = cont._dataset_spec[name]["axes"].index("freq")
        idx = [slice(None)] * len(ds.shape)
        idx[fax] = sl
        if ds.on_device:
            out.datasets[name] = containers.Dataset(dev=ds._dev[tuple(idx)].contiguous(), attrs=ds.attrs)
        else:
            out.datasets[name] = containers.Dataset(host=np.ascontiguousarray(ds.host()[tuple(idx)]), attrs=ds.attrs)
    return out


def allgather_map(local_map, nfreq_total=None, group=None):
    """Gather every rank's ``Map`` frequency slab into the full ``Map`` on every rank.

    One ``all_gather`` of ``[nfreq_local, npol, npix]`` float64 shards (uneven slabs are padded
    to the largest and trimmed after).  Over xGMI this is the only data-path collective of a
    sim -> map run; shards of a frequency-sharded map are contiguous so no reordering is needed.
    """
    rank, world = _rank_world(group)
    if world == 1:
        return local_map
    backend = dist.get_backend(group)
    ds = local_map.map
    if backend == "nccl":
        if not ds.on_device:
            raise RuntimeError("allgather_map over RCCL needs a device-resident map")
        local = ds._dev
    else:
        local = torch.from_numpy(np.ascontiguousarray(ds.host()))
    nloc = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(nloc) for _ in range(world)]
    dist.all_gather(counts, nloc, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if nfreq_total is not None and sum(counts) != nfreq_total:
        raise ValueError(f"shards hold {sum(counts)} frequencies, expected {nfreq_total}")
    pad = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax, *local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    full = torch.empty((world * nmax, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, pad.contiguous(), group=group)
    parts = [full[r * nmax : r * nmax + counts[r]] for r in range(world)]
    gathered = torch.cat(parts, dim=0) if any(c != nmax for c in counts) else full

    # frequency axis: gather the (tiny) structured index maps as objects
    fm = local_map.index_map["freq"]
    fms = [None] * world
    dist.all_gather_object(fms, fm, group=group)
    freq = np.concatenate(fms)
    out = containers.Map(freq=freq, pol=local_map.index_map["pol"], pixel=local_map.index_map["pixel"], attrs_from=local_map, allocate=False)
    if backend == "nccl":
        out.attach("map", gathered)
    else:
        out.datasets["map"] = containers.Dataset(host=gathered.numpy(), attrs=ds.attrs)
    return out


# ---------------------------------------------------------------- frequency <-> m exchange
# The SVD filter (analysis/svdfilter.py) couples all frequencies of one m: the one task on this
# path with a real exchange step (the reference's ``mmodes.redistribute("m")``, svdfilter.py:34,93).
# It is an all-to-all of ``[m-slab, msign, freq-slab, base]`` blocks, issued as one batch of
# point-to-point sends/receives (RCCL groups them into a single all-to-all over xGMI; gloo, which
# has no all-to-all, runs the same code in the CPU tests).


def _wire(t):
    """Complex tensors travel as their (re, im) float view (gloo has no complex types)."""
    return torch.view_as_real(t) if t.is_complex() else t


def _exchange(send_blocks, recv_blocks, group=None):
    rank, world = _rank_world(group)
    ops = []
    for q in range(world):
        if q == rank:
            recv_blocks[q].copy_(send_blocks[q])
            continue
        ops.append(dist.P2POp(dist.isend, _wire(send_blocks[q]), q, group))
        ops.append(dist.P2POp(dist.irecv, _wire(recv_blocks[q]), q, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def freq_to_m(mvis, mweight, group=None):
    """Frequency-sharded ``[n_m, 2, nfreq_local, nbase]`` -> m-sharded ``[n_m_local, 2, nfreq_total, nbase]``.

    Returns ``(mvis_m, mweight_m, layout)``; ``layout`` is what :func:`m_to_freq` / :func:`gather_m`
    need to undo it (``None`` on a single rank, where this is the identity).
    """
    rank, world = _rank_world(group)
    if world == 1:
        return mvis, mweight, None
    n_m, _, nf_loc, nbase = mvis.shape
    cnt = torch.tensor([nf_loc], dtype=torch.int64, device=mvis.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    nf = [int(c.item()) for c in cnts]
    f0 = [sum(nf[:q]) for q in range(world)]
    mc = [split_local(n_m, q, world) for q in range(world)]  # (count, start) of every rank's m slab
    lay = {"world": world, "rank": rank, "nf": nf, "f0": f0, "mc": mc, "n_m": n_m, "group": group}
    out = []
    for src in (mvis, mweight):
        send = [src[mc[q][1] : mc[q][1] + mc[q][0]].contiguous() for q in range(world)]
        recv = [torch.empty((mc[rank][0], 2, nf[q], nbase), dtype=src.dtype, device=src.device) for q in range(world)]
        _exchange(send, recv, group)
        out.append(torch.cat(recv, dim=2).contiguous())
    return out[0], out[1], lay


def m_to_freq(mvis_m, lay):
    """Inverse of :func:`freq_to_m` for one array: m-sharded -> this rank's frequency slab, all m."""
    if lay is None:
        return mvis_m
    world, rank, nf, f0, mc = lay["world"], lay["rank"], lay["nf"], lay["f0"], lay["mc"]
    nbase = mvis_m.shape[3]
    send = [mvis_m[:, :, f0[q] : f0[q] + nf[q]].contiguous() for q in range(world)]
    recv = [torch.empty((mc[q][0], 2, nf[rank], nbase), dtype=mvis_m.dtype, device=mvis_m.device) for q in range(world)]
    _exchange(send, recv, lay["group"])
    return torch.cat(recv, dim=0).contiguous()


def gather_m(x_m, lay):
    """All ranks' ``[n_m_local, ...]`` pieces -> the full ``[n_m, ...]`` array on every rank."""
    if lay is None:
        return x_m
    world, mc = lay["world"], lay["mc"]
    nmax = max(c for c, _ in mc)
    pad = torch.zeros((nmax, *x_m.shape[1:]), dtype=x_m.dtype, device=x_m.device)
    pad[: x_m.shape[0]] = x_m
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=lay["group"])
    return torch.cat([parts[q][: mc[q][0]] for q in range(world)], dim=0).contiguous()


def allreduce_max(x, device=None, group=None):
    """Maximum of a Python float over all ranks (``svdfilter.py:113``)."""
    rank, world = _rank_world(group)
    if world == 1:
        return float(x)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
