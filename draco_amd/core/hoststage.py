"""Host -> HBM upload of beam-transfer tiles through a ring of pinned staging slots.

The reference reads one ``beam_m`` tile from HDF5 inside every ``_solve_m`` call
(``mapmaker.py:160-162``).  Here a slab's tiles are produced in bulk
(:meth:`~draco_amd.core.products.BeamTransferProvider.beam_block`) by a pool of worker threads,
each packing one chunk into a pinned slot, while the calling thread issues one asynchronous copy per
finished chunk on the fill stream: packing, PCIe transfer and the previous slab's solves overlap.
Chunks that already live in pinned host memory skip the staging and are copied from where they are.

PyTorch supplies the pinned allocations, the stream and the events; nothing numerical happens here.
"""

from __future__ import annotations

import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def default_workers():
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # pragma: no cover
        n = os.cpu_count() or 1
    return max(1, min(16, n - 1))  # a GPU box gives one GPU's job a 16-core share of the host


class HostStager:
    """A ring of pinned slots + worker threads; one per fill stream."""

    _cache: dict = {}
    _lock = threading.Lock()

    def __init__(self, device, slot_bytes=128 << 20, workers=None):
        self.device = device
        self.workers = int(workers or default_workers())
        self.slot_bytes = int(slot_bytes)
        self.nslot = self.workers + 3
        self._ring = torch.empty(self.nslot * self.slot_bytes, dtype=torch.uint8).pin_memory()
        self._ring_np = self._ring.numpy()
        self._free_ev = [None] * self.nslot  # event of the last copy that read the slot
        self._pool = ThreadPoolExecutor(self.workers, thread_name_prefix="dmm-stage")
        self.bytes_staged = 0
        self.bytes_direct = 0

    @classmethod
    def get(cls, device, **kw):
        key = (torch.device(device).index, tuple(sorted(kw.items())))
        with cls._lock:
            if key not in cls._cache:
                cls._cache[key] = cls(device, **kw)
            return cls._cache[key]

    @classmethod
    def release(cls):
        with cls._lock:
            for s in cls._cache.values():
                s._pool.shutdown(wait=True)
            cls._cache.clear()

    def slot(self, i):
        return self._ring_np[i * self.slot_bytes : (i + 1) * self.slot_bytes]

    def upload(self, jobs, pool_u8, stream):
        """Run ``jobs`` in order; every job is ``(dst_byte_offset, nbytes, source)`` with ``nbytes <= slot_bytes``.

        ``source`` is either a pinned uint8 tensor of ``nbytes`` (copied from where it is) or a callable
        ``produce(out_u8)`` that fills a ``nbytes`` NumPy uint8 view (run on a worker thread; NumPy copies and
        conversions release the GIL).  Copies are enqueued on ``stream`` in job order; the call returns when the last
        copy has been ENQUEUED (not finished): order later work behind an event recorded on ``stream``.
        """
        jobs = list(jobs)
        for _, nbytes, src in jobs:  # before any worker is started: a producer must never see a truncated view
            if nbytes > self.slot_bytes and callable(src):
                raise ValueError(f"staged chunk of {nbytes} bytes exceeds the slot size {self.slot_bytes}")
        staged = [j for j, (_, _, src) in enumerate(jobs) if callable(src)]
        slot_of = {j: k % self.nslot for k, j in enumerate(staged)}

        def run(j):
            s = slot_of[j]
            ev = self._free_ev[s]
            if ev is not None:
                ev.synchronize()  # the copy that last read this slot (host wait on the fill stream only)
            nbytes, produce = jobs[j][1], jobs[j][2]
            produce(self.slot(s)[:nbytes])
            return s

        futs = {}
        ahead = iter(staged)
        inflight = 0

        def feed(done=0):
            nonlocal inflight
            inflight -= done
            while inflight < self.nslot:
                j = next(ahead, None)
                if j is None:
                    return
                futs[j] = self._pool.submit(run, j)
                inflight += 1

        feed()
        try:
            self._copy_loop(jobs, futs, feed, pool_u8, stream)
        except BaseException:
            # a producer (or a copy) failed: nothing may keep writing into ring slots after the call has returned
            for f in futs.values():
                f.cancel()
            for f in futs.values():
                try:
                    f.result()
                except BaseException:  # noqa: BLE001  (the first failure is the one reported)
                    pass
            raise

    def _copy_loop(self, jobs, futs, feed, pool_u8, stream):
        with torch.cuda.stream(stream):
            for j, (dst, nbytes, src) in enumerate(jobs):
                if callable(src):
                    s = futs.pop(j).result()
                    pool_u8[dst : dst + nbytes].copy_(self._ring[s * self.slot_bytes : s * self.slot_bytes + nbytes], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    self._free_ev[s] = ev
                    self.bytes_staged += nbytes
                    feed(done=1)
                else:
                    pool_u8[dst : dst + nbytes].copy_(src, non_blocking=True)
                    self.bytes_direct += nbytes
