"""Device plumbing: one :class:`Context` per GPU.

PyTorch is used for what it is good at here -- device memory, streams and (in
``draco_amd.parallel``) ``torch.distributed`` -- and nothing else: every number the path
produces comes out of a kernel in ``libdraco_amd.so``.  Raises if there is no GPU.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_NP2TORCH = {
    np.dtype(np.complex64): torch.complex64,
    np.dtype(np.complex128): torch.complex128,
    np.dtype(np.float32): torch.float32,
    np.dtype(np.float64): torch.float64,
    np.dtype(np.int32): torch.int32,
    np.dtype(np.uint8): torch.uint8,
}


class Context:
    """Owns a ``dmm_ctx`` bound to ``cuda:<device>`` and torch's current stream on it."""

    _cache: dict[int, "Context"] = {}
    _side_cache: dict[int, "Context"] = {}
    _fill_cache: dict[int, "Context"] = {}

    def __init__(self, device: int = 0, stream: "torch.cuda.Stream | None" = None):
        if not torch.cuda.is_available():
            raise RuntimeError(
                "draco_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback."
            )
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p()
        _lib.check(_lib.lib.dmm_ctx_create(self.device_index, C.byref(h)))
        self.handle = h
        self.stream = stream  # None: follow torch's current stream; else pinned to this one
        self._held: list = []  # tensors in use by this context's stream, released at sync()
        self.bind_stream()

    @classmethod
    def get(cls, device: int | None = None) -> "Context":
        if device is None:
            device = torch.cuda.current_device()
        if device not in cls._cache:
            cls._cache[device] = cls(device)
        ctx = cls._cache[device]
        ctx.bind_stream()
        return ctx

    @classmethod
    def side(cls, device: int | None = None) -> "Context":
        """A second ``dmm_ctx`` of the same GPU pinned to its own stream (own scratch, own tables).

        For stages that can run beside the main stream's work, e.g. the compute-bound alm2map of
        finished frequencies under the HBM-bound solves of the next slab.  Order the two with
        events: ``side.wait_for(main_stream)`` before launching, ``side.join(main_stream)`` after.
        """
        if device is None:
            device = torch.cuda.current_device()
        if device not in cls._side_cache:
            dev = torch.device("cuda", device)
            try:  # lowest priority the device offers: side work yields to the main stream's kernels
                least = int(torch.cuda.Stream.priority_range()[0])
            except Exception:
                least = 0
            try:
                st = torch.cuda.Stream(device=dev, priority=least)
            except Exception:
                st = torch.cuda.Stream(device=dev)
            cls._side_cache[device] = cls(device, st)
        return cls._side_cache[device]

    @classmethod
    def fill(cls, device: int | None = None) -> "Context":
        """A third ``dmm_ctx`` of the GPU pinned to a stream of its own, for filling B buffers (H2D copies of the next
        slab's tiles, or a provider's fill kernels) under the current slab's solves; ordered by events in
        ``analysis/_solve.py``."""
        if device is None:
            device = torch.cuda.current_device()
        if device not in cls._fill_cache:
            cls._fill_cache[device] = cls(device, torch.cuda.Stream(device=torch.device("cuda", device)))
        return cls._fill_cache[device]

    # ---- buffer ownership across streams
    # Rule: a tensor handed to work on a context whose stream is not the one torch allocated it on is (a) recorded on
    # that stream (the caching allocator then will not hand its memory to anyone else before the work has finished)
    # and (b) kept referenced by the context until `join()` / `sync()` -- so correctness never depends on which Python name
    # happens to stay alive at the call site.  `uses()` is called by every task that launches on `Context.side()`.
    def uses(self, *tensors):
        st = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        for t in tensors:
            if t is None:
                continue
            t.record_stream(st)
            self._held.append(t)
        return tensors[0] if len(tensors) == 1 else tensors

    def wait_for(self, other: "torch.cuda.Stream"):
        """Work launched on this (pinned) context after the call starts after ``other``'s work so far."""
        ev = torch.cuda.Event()
        ev.record(other)
        self.stream.wait_event(ev)

    def join(self, other: "torch.cuda.Stream"):
        """``other``'s later work waits for everything launched on this (pinned) context so far."""
        other.wait_stream(self.stream)
        # from here on `other` is ordered behind this context's work and `record_stream` guards the allocator:
        # the references taken by `uses()` have done their job
        self._held.clear()

    def release_held(self):
        """Drop the references taken by `uses()` without ordering anything: for a caller that has recorded its own
        event behind the work and relies on `record_stream` for the allocator."""
        self._held.clear()

    def bind_stream(self):
        """Launch on torch's current stream of this device (so torch allocations/copies order with kernels)."""
        s = (self.stream if self.stream is not None else torch.cuda.current_stream(self.device)).cuda_stream
        _lib.check(_lib.lib.dmm_ctx_set_stream(self.handle, C.c_void_p(s)))

    def sync(self):
        """Drain the context's stream (and the library's second one); buffers held by `uses()` are released."""
        _lib.check(_lib.lib.dmm_ctx_sync(self.handle))
        self._held.clear()

    def timer_start(self):
        _lib.check(_lib.lib.dmm_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        _lib.check(_lib.lib.dmm_timer_stop(self.handle, C.byref(ms)))
        return float(ms.value)

    # ---- memory helpers
    def empty(self, shape, dtype) -> torch.Tensor:
        return torch.empty(tuple(int(s) for s in shape), dtype=_NP2TORCH[np.dtype(dtype)], device=self.device)

    def zeros(self, shape, dtype) -> torch.Tensor:
        return torch.zeros(tuple(int(s) for s in shape), dtype=_NP2TORCH[np.dtype(dtype)], device=self.device)

    def to_device(self, arr, dtype=None) -> torch.Tensor:
        """ndarray / tensor -> contiguous device tensor of ``dtype`` (H2D copy if needed)."""
        if isinstance(arr, torch.Tensor):
            t = arr
            if dtype is not None and t.dtype != _NP2TORCH[np.dtype(dtype)]:
                t = t.to(_NP2TORCH[np.dtype(dtype)])
            return t.to(self.device).contiguous()
        a = np.ascontiguousarray(arr, dtype=dtype)
        return torch.from_numpy(a).to(self.device)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib.dmm_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class StreamDone:
    """An event recorded behind work on one stream, in the form `containers.Dataset(pending=...)` takes: ``done()``
    polls it, ``order()`` makes torch's current stream of the device wait for it (no host wait either way)."""

    def __init__(self, stream, device):
        self.device = device
        self.event = torch.cuda.Event()
        self.event.record(stream)

    def done(self) -> bool:
        return bool(self.event.query())

    def order(self):
        torch.cuda.current_stream(self.device).wait_event(self.event)

    def host_wait(self):
        self.event.synchronize()


def ptr(t) -> C.c_void_p:
    """Raw device pointer of a tensor (``None`` -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())
