"""Library-owned workspace arena (the reference's ``resizeFunctional`` hook, rasterize_points.cu:31-37,107-109, re-thought).

The reference obtains every per-forward buffer - geometry / binning / image state, the output maps, the 1.66 GB tracer list -
through torch's caching allocator, and its drivers call ``torch.cuda.empty_cache()`` every iteration
(train_semantic.py:208): everything is handed back to the driver and ``hipMalloc``-ed again on the next iteration (~1 GB per view
at 1080p with a 32-channel feature: tens of milliseconds).  This arena keeps those buffers: a pool of live ``uint8`` base
tensors per (device, stream, size class); ``empty()`` leases a view of a free block, and a block is free again when every tensor
that views its storage has died - read from the storage's reference count, so the lease ends exactly when the caller (or the
autograd graph holding the forward's state) drops the tensors; no hook, no explicit release.  ``empty_cache()`` frees nothing of it.

* Ownership contract unchanged: the caller gets tensors that nobody else writes while any of them is alive.
* Stream safety like the caching allocator's: a block is reused only by the stream that leased it; after
  ``used_on(tensor, stream)`` (the analogue of ``Tensor.record_stream``) the block is not handed out again until that stream has
  passed the point where the block was found idle (an event per such stream, polled - no stream ever WAITS for another because of
  the arena: a lease that would have to wait takes another block).
* Size classes (x 1.25) bound the number of distinct blocks under a drifting size; at most ``MAX_FREE`` idle blocks per class
  and ``ISR_ARENA_MAX_GB`` (default: half of the device's memory) in total are kept - beyond that idle blocks go back to the
  caching allocator.  Under memory pressure the arena lets go: an out-of-memory error while a block is created trims every idle
  block, empties torch's cache and retries once; the drop-in's pressure-gated ``empty_cache()`` trims it too (``dropin.py``).
* Leases are NOT autograd views of the block (``Tensor.set_`` on the block's storage): an in-place operation on a render output
  created inside the rasterizer's ``autograd.Function`` behaves like on the reference's fresh tensors.  (``torch.save`` of a
  lease serialises the whole block; clone first.)
* ``ISR_ARENA=0`` disables it (plain ``torch.empty``).  Requests below ``MIN_BYTES`` are not pooled.
"""
from __future__ import annotations

import bisect
import os
import threading

import torch

from . import _hot

# (the lease bookkeeping reads the storage's reference count through torch._C._storage_Use_Count - the call torch's own CUDA-graph
# trees use for the same purpose; a torch without it gets plain torch.empty)
ENABLED = os.environ.get("ISR_ARENA", "1") != "0" and hasattr(torch._C, "_storage_Use_Count")
MIN_BYTES = 1 << 20
MAX_FREE = 3
# cap on the bytes the arena keeps: ISR_ARENA_MAX_GB, else half of the device's memory (read at the first lease on a device)
_MAX_GB_ENV = os.environ.get("ISR_ARENA_MAX_GB")
MAX_BYTES = int(float(_MAX_GB_ENV) * (1 << 30)) if _MAX_GB_ENV else None
MAX_FRACTION = 0.5
_DEVICE_CAP = {}
_LOCK = threading.RLock()      # empty() runs on the main thread in forward and on autograd's device threads in backward

_CLASSES = [MIN_BYTES]
_POOLS = {}            # (device index, stream handle, class bytes) -> [_Block]
_BY_PTR = {}           # base data_ptr -> _Block
_TOTAL = [0]
STATS = {"leases": 0, "new_blocks": 0, "dropped_blocks": 0}


def _class_of(n: int) -> int:
    while _CLASSES[-1] < n:
        _CLASSES.append((int(_CLASSES[-1] * 1.25) + 4095) // 4096 * 4096)
    return _CLASSES[bisect.bisect_left(_CLASSES, n)]


_COUNT = getattr(torch._C, "_storage_Use_Count", None)


class _Block:
    __slots__ = ("base", "idle_count", "foreign", "events", "nbytes", "_st", "_cd")

    def __init__(self, nbytes, dev):
        try:
            base = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        except torch.OutOfMemoryError:
            # the arena's idle blocks are invisible to torch's own out-of-memory retry: let them go, then torch's cache, once
            trim()
            _real_empty_cache()
            base = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._adopt(base, nbytes)

    def _adopt(self, base, nbytes):
        self.base = base
        self._st = base.untyped_storage()       # held for good: the probe then makes no temporary handle of its own
        self._cd = self._st._cdata
        self.idle_count = _COUNT(self._cd)      # the base tensor + that handle
        self.foreign = set()
        self.events = None
        self.nbytes = nbytes

    def idle(self) -> bool:
        return _COUNT(self._cd) == self.idle_count

    def reusable(self, stream: int) -> bool:
        """Idle, and every other stream that used the last lease has passed the point where that was first seen
        (``stream``: the raw handle of the stream that asks)."""
        if not self.idle():
            return False
        if self.foreign:
            self.events = []
            for s in self.foreign:
                if s.cuda_stream != stream:
                    ev = torch.cuda.Event()
                    ev.record(s)
                    self.events.append(ev)
            self.foreign.clear()
        if self.events:
            self.events = [e for e in self.events if not e.query()]
            if self.events:
                return False
        return True


def _real_empty_cache():
    """torch's own ``empty_cache`` even when the drop-in has replaced the public name (dropin.py keeps the original here)."""
    fn = getattr(torch.cuda, "_isr_real_empty_cache", None) or torch.cuda.empty_cache
    fn()


def _cap(dev_index: int) -> int:
    if MAX_BYTES is not None:
        return MAX_BYTES
    c = _DEVICE_CAP.get(dev_index)
    if c is None:
        try:
            c = int(torch.cuda.mem_get_info(dev_index)[1] * MAX_FRACTION)
        except Exception:
            c = 64 << 30
        _DEVICE_CAP[dev_index] = c
    return c


def empty(shape, dtype, device) -> torch.Tensor:
    """``torch.empty(shape, dtype=dtype, device=device)`` out of the arena (uninitialised, 256-byte aligned)."""
    if isinstance(shape, int):
        shape = (shape,)
    n = 1
    for s in shape:
        n *= int(s)
    nbytes = n * dtype.itemsize
    dev = device if isinstance(device, torch.device) else torch.device(device)
    if not ENABLED or nbytes < MIN_BYTES or dev.type != "cuda":
        return torch.empty(shape, dtype=dtype, device=dev)
    with _LOCK:
        return _lease(shape, dtype, dev, nbytes)


def _lease(shape, dtype, dev, nbytes) -> torch.Tensor:
    cls = _class_of(nbytes)
    stream = _hot.raw_stream(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), stream, cls)
    pool = _POOLS.get(key)
    if pool is None:
        pool = _POOLS[key] = []
    blk = None
    idle = 0
    surplus = len(pool) > MAX_FREE            # (only then is the number of idle blocks of interest: stop at the first otherwise)
    for b in pool:
        if b.reusable(stream):
            idle += 1
            if blk is None:
                blk = b
                if not surplus:
                    break
    if idle > MAX_FREE:                                   # a burst is over: hand the surplus back
        keep = []
        for b in pool:
            if b is not blk and idle > MAX_FREE and b.reusable(stream):
                idle -= 1
                _drop(b)
            else:
                keep.append(b)
        pool[:] = keep
    if blk is None:
        cap = _cap(key[0])
        if _TOTAL[0] + cls > cap:
            trim(_TOTAL[0] + cls - cap)
        blk = _Block(cls, dev)
        pool.append(blk)
        _BY_PTR[blk.base.data_ptr()] = blk
        _TOTAL[0] += cls
        STATS["new_blocks"] += 1
    STATS["leases"] += 1
    # a tensor of its own on the block's storage, not a view of `base`: the storage's reference count still says when the lease
    # has ended, and autograd sees what the reference hands out - a fresh tensor (in-place operations on the render outputs)
    return torch.empty(0, dtype=dtype, device=dev).set_(blk._st, 0, tuple(int(v) for v in shape))


def used_on(t: torch.Tensor, stream) -> None:
    """``t`` (a lease, or any tensor) is used by kernels on ``stream``: the arena's analogue of ``Tensor.record_stream``."""
    if t is None or not t.is_cuda:
        return
    blk = _BY_PTR.get(t.untyped_storage().data_ptr())
    if blk is not None:
        with _LOCK:
            blk.foreign.add(stream)
    else:
        t.record_stream(stream)


def _drop(b: _Block) -> None:
    _BY_PTR.pop(b.base.data_ptr(), None)
    _TOTAL[0] -= b.nbytes
    STATS["dropped_blocks"] += 1


def trim(nbytes: int = None) -> int:
    """Give idle blocks back to the caching allocator (all of them, or at least ``nbytes`` worth).  Returns the bytes released."""
    freed = 0
    with _LOCK:
        for key in list(_POOLS):
            keep = []
            for b in _POOLS[key]:
                # key[1]: the stream that OWNS the pool - a block used on another stream (used_on) stays until that stream has
                # passed it, whichever stream happens to be current where trim() is called
                if (nbytes is None or freed < nbytes) and b.reusable(key[1]):
                    freed += b.nbytes
                    _drop(b)
                else:
                    keep.append(b)
            # in place: _lease() may hold this very list (it trims on the cap path and through _Block's out-of-memory retry) and
            # appends its new block to it afterwards - a rebound or deleted list would orphan that block for good
            _POOLS[key][:] = keep
    return freed


def reserved_bytes() -> int:
    return _TOTAL[0]
