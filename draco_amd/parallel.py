"""Frequency sharding across GPUs: one process per GPU, RCCL for the final map all-gather only.

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
