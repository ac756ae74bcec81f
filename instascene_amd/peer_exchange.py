"""Direct gradient exchange over peer-mapped buffers - SURVEY section 5's alternative to RCCL's ring for the one collective of
the data-parallel step (sum all-reduce of dL/dparam, ``[P,F]`` fp32: 192 MB at C3/C4, 1.28 GB at C5).

An MI355X node is fully connected: every GPU has a point-to-point xGMI link to each of its 7 peers (~153 GB/s each way).
A ring all-reduce keeps ONE link per GPU busy per step (2 (W-1)/W x bytes over one link: ~2.2 ms for 192 MB at W = 8).  Here
every rank maps the gradient buffers of all peers into its address space (``hipIpcGetMemHandle`` / ``hipIpcOpenMemHandle``,
through torch's CUDA-IPC tensor sharing) and

  1. reduce-scatter: PULLS its own 1/W shard of the rows from all W buffers at once and sums them in rank order
     (``iso_peer_sum``: one kernel, W coalesced streams, all 7 links busy) - bytes per link: bytes / W;
  2. all-gather: pulls the W-1 reduced shards it does not own from their owners' buffers (plain device copies).

Per link that is 2 x bytes / W instead of 2 (W-1)/W x bytes: 0.31 ms for 192 MB at W = 8 by the link figure above.
Sums are taken in rank order on every rank, so the replicas are bit-identical to each other (and, for W = 2, to any other
all-reduce: a + b is commutative).

Synchronisation between the phases is a ``torch.distributed`` barrier on the control-plane group plus a stream
synchronisation - correct, and what a functional test needs; flags in device memory polled by the kernels (no host round
trip) are the obvious next step once an 8-GPU node is there to time it on.  No scaling number is claimed for this path:
it has only ever run with two ranks sharing one GPU (tests/test_gpu_dist.py)."""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch
import torch.distributed as dist

from ._lib import check, lib


class PeerExchange:
    """All-reduce (sum) of a fixed-size fp32 buffer across the ranks of ``group`` by direct peer access.

    ``buffer``: this rank's gradient buffer (CUDA, fp32, contiguous, 16-byte aligned, allocated once and reused every step -
    the mapping is set up here, not per step).  Every rank must construct its exchange collectively."""

    def __init__(self, buffer: torch.Tensor, group=None):
        if not (buffer.is_cuda and buffer.dtype == torch.float32 and buffer.is_contiguous()):
            raise ValueError("PeerExchange: a contiguous CUDA fp32 buffer is required")
        if buffer.data_ptr() % 16 != 0:
            raise ValueError("PeerExchange: the buffer must be 16-byte aligned")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 16:
            raise ValueError("PeerExchange: at most 16 ranks")
        self.buf = buffer
        self.flat = buffer.view(-1)
        n = self.flat.numel()
        # shard boundaries in elements, multiples of 4 (float4 accesses in iso_peer_sum)
        per = ((n + self.world - 1) // self.world + 3) // 4 * 4
        self.bounds = [(min(n, r * per), min(n, (r + 1) * per)) for r in range(self.world)]
        # torch's CUDA-IPC sharing: (rebuild function, arguments holding the hipIpcMemHandle of the allocation + offset)
        from torch.multiprocessing.reductions import reduce_tensor
        mine = reduce_tensor(self.flat)
        table: List[Optional[tuple]] = [None] * self.world
        dist.all_gather_object(table, mine, group=group)
        self.peers: List[torch.Tensor] = []
        for r, (fn, args) in enumerate(table):
            if r == self.rank:
                self.peers.append(self.flat)
            else:
                t = fn(*args)                         # opens the peer's handle: a tensor aliasing the peer's memory
                if t.numel() != n:
                    raise RuntimeError("PeerExchange: the ranks' buffers differ in size")
                self.peers.append(t)
        self._ptrs = (ctypes.c_void_p * self.world)(*[p.data_ptr() for p in self.peers])
        r0, r1 = self.bounds[self.rank]
        self.shard = torch.empty(max(1, r1 - r0), dtype=torch.float32, device=buffer.device)
        dist.barrier(group=group)

    def _sync(self):
        torch.cuda.synchronize(self.buf.device)
        dist.barrier(group=self.group)

    def all_reduce_(self) -> torch.Tensor:
        """Sum ``buffer`` across the ranks in place (every rank ends with the same bits)."""
        L = lib()
        dev = self.buf.device
        r0, r1 = self.bounds[self.rank]
        self._sync()                                   # every rank's gradient is complete in its buffer
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if r1 > r0:
                check(L.iso_peer_sum(self.world, self._ptrs, r0, r1 - r0, ctypes.c_void_p(self.shard.data_ptr()), st), "iso_peer_sum")
        self._sync()                                   # every rank has READ all buffers: they may be overwritten now
        if r1 > r0:
            self.flat[r0:r1].copy_(self.shard[:r1 - r0])
        self._sync()                                   # every owner's shard is in place in its own buffer
        for r in range(self.world):
            a, b = self.bounds[r]
            if r != self.rank and b > a:
                self.flat[a:b].copy_(self.peers[r][a:b])      # pull the reduced shard from its owner
        self._sync()
        return self.buf
