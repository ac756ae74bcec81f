"""Data-parallel plumbing for the per-view training loop (new functionality — the reference is
single-process).  One process per GPU; Gaussian parameters replicated; rank r of W renders
view ``(step * W + r) % n_views``; the trainable parameters' gradients are summed with ONE all-reduce
per tensor (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests), after which every rank
applies the identical optimiser step, so replicas stay bit-identical without a broadcast.

xGMI note: the [P,F] gradient is one contiguous 192 MB (C3/C4) or 1.28 GB (C5) message — far above
the size where RCCL switches to its bandwidth-optimal algorithm, so it is sent as a single collective
rather than bucketed; nothing else is exchanged on the data path."""
from __future__ import annotations

from typing import Iterable

import torch
import torch.distributed as dist


_EPOCHS = {}


def view_order(n_views: int, epoch: int, seed: int = 0):
    """The order in which epoch ``epoch`` visits the ``n_views`` training views: a seeded random permutation.  The reference
    pops a random view from a stack that it refills when empty (train_semantic.py:96-100, train.py:66-69), i.e. every view
    exactly once per epoch in random order; a permutation drawn per epoch is that process with the draws made up front, so
    the NEXT view is known one step early (its geometry pass is prefetched) and every rank of a data-parallel job can
    compute the whole schedule without communication."""
    key = (n_views, epoch, seed)
    order = _EPOCHS.get(key)
    if order is None:
        import numpy as np
        if len(_EPOCHS) > 64:
            _EPOCHS.clear()
        order = _EPOCHS[key] = np.random.RandomState((seed * 1000003 + 7919 * epoch + 12345) & 0x7FFFFFFF).permutation(n_views).tolist()
    return order


def view_for(step: int, rank: int, world: int, n_views: int, seed: int = 0, shuffle: bool = True) -> int:
    """View of rank ``rank`` at iteration ``step``: the ranks take consecutive entries of the epoch's random order
    (:func:`view_order`), so one step of a W-rank job covers W distinct views and an epoch covers every view once.
    ``shuffle=False``: plain round-robin."""
    g = step * world + rank
    if not shuffle:
        return g % n_views
    return view_order(n_views, g // n_views, seed)[g % n_views]


def allreduce_grads(params: Iterable[torch.nn.Parameter], world: int) -> None:
    """Sum gradients across ranks in place (parity definition: all-reduced gradient == sum of the
    single-GPU gradients of the same views)."""
    if world <= 1:
        return
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)


def allreduce_grads_async(params: Iterable[torch.nn.Parameter], world: int):
    """Start the gradient all-reduce and return the work handles (``[]`` for one rank): the caller overlaps it with work
    that does not read the gradient (the next view's geometry pass) and calls :func:`wait_all` before the optimiser
    step.  RCCL runs the collective on its own stream; ``wait()`` makes the current stream wait for it."""
    works = []
    if world <= 1:
        return works
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        works.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True))
    return works


def allreduce_bucket(tensors, world: int, like=None):
    """Sum a list of gradient tensors across ranks as ONE flat collective (what a train.py-style step exchanges: the six
    parameter groups' gradients, ~232 bytes per Gaussian, instead of six blocking all-reduces).  A ``None`` entry stands
    for zeros of the shape of ``like[i]`` (every rank must put the same sizes into the bucket).  Returns the summed tensors
    (views of the flat buffer); with one rank the input is returned as it is."""
    if world <= 1:
        return list(tensors)
    parts = []
    for i, t in enumerate(tensors):
        if t is None:
            if like is None or like[i] is None:
                raise ValueError("allreduce_bucket: a missing gradient needs like[i] for its shape")
            t = torch.zeros_like(like[i], dtype=torch.float32)
        parts.append(t)
    # every segment starts on a 16-byte boundary of the flat buffer (kernels read the returned views with float4 loads:
    # iso_gaussian_adam_step's rotation gradient): segments are padded to multiples of four floats
    sizes = [t.numel() for t in parts]
    starts, at = [], 0
    for n in sizes:
        starts.append(at)
        at += (n + 3) // 4 * 4
    flat = torch.zeros(at, dtype=torch.float32, device=parts[0].device)
    for t, a, n in zip(parts, starts, sizes):
        flat[a:a + n].copy_(t.reshape(-1))
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return [flat[a:a + n].view(t.shape) for t, a, n in zip(parts, starts, sizes)]


def row_ranges(n_rows: int, chunks: int):
    """``chunks`` contiguous row ranges ``(r0, r1)`` covering ``[0, n_rows)``."""
    c = max(1, min(int(chunks), max(1, n_rows)))
    return [(n_rows * k // c, n_rows * (k + 1) // c) for k in range(c)]


def allreduce_rows_async(t: torch.Tensor, r0: int, r1: int, world: int):
    """Start the sum all-reduce of rows ``[r0, r1)`` of a contiguous ``[P, ...]`` tensor; returns the work handle (None for
    one rank or an empty range).  A trainer that produces the gradient range by range overlaps the collective of one range
    with the kernels of the others (SegTrainer._tail_with_allreduce)."""
    if world <= 1 or r1 <= r0:
        return None
    return dist.all_reduce(t[r0:r1], op=dist.ReduceOp.SUM, async_op=True)


def shard_rows(n_rows: int, rank: int, world: int):
    """Rows ``[r0, r1)`` owned by ``rank`` when ``n_rows`` (a multiple of ``world``) are dealt out in contiguous shards."""
    per = n_rows // world
    return rank * per, (rank + 1) * per


def reduce_scatter_rows(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Sum the contiguous ``[P, ...]`` tensor ``t`` over the ranks and return THIS rank's shard of the sum (rows
    ``shard_rows(P, rank, world)``) - half the traffic of an all-reduce.  RCCL: one ``reduce_scatter_tensor``; gloo (the CPU /
    one-GPU tests) has no reduce-scatter: all-reduce and slice, the same values."""
    r0, r1 = shard_rows(t.shape[0], rank, world)
    if dist.get_backend() == "nccl":
        out = torch.empty_like(t[r0:r1])
        dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM)
        return out
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t[r0:r1]


def all_gather_rows(t: torch.Tensor, rank: int, world: int) -> None:
    """Every rank holds valid rows ``shard_rows(P, rank, world)`` of the contiguous ``t``; afterwards all of ``t`` is valid
    everywhere (the other half of an all-reduce's traffic)."""
    r0, r1 = shard_rows(t.shape[0], rank, world)
    mine = t[r0:r1].clone()
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(t, mine)
        return
    per = r1 - r0
    dist.all_gather([t[k * per:(k + 1) * per] for k in range(world)], mine)


def wait_all(works) -> None:
    for w in works:
        if w is not None:
            w.wait()


def replicas_in_sync(t: torch.Tensor, world: int) -> bool:
    """True iff every rank holds bit-identical ``t`` (checked with a max/min all-reduce)."""
    if world <= 1:
        return True
    hi, lo = t.detach().clone(), t.detach().clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return bool(torch.equal(hi, lo))
