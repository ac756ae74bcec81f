"""World-size-2 gloo test of the data-parallel path (runs on CPU): view sharding, gradient
all-reduce == sum of per-view gradients, replicas stay bit-identical after Adam."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from instascene_amd.dist_utils import (allreduce_grads, allreduce_grads_async, allreduce_rows_async, replicas_in_sync,
                                       row_ranges, view_for, wait_all)


def _toy_loss(param, view):
    # a deterministic, view-dependent differentiable stand-in for render+loss
    g = torch.Generator().manual_seed(100 + view)
    w = torch.randn(param.shape, generator=g)
    return ((param * w).sum()) ** 2 * 1e-3 + (param * w).sin().sum()


def _worker(rank, world, port, steps, n_views, out, overlapped=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.linspace(-1, 1, 64 * 8).reshape(64, 8).clone())
    opt = torch.optim.Adam([p], lr=0.025, eps=1e-15)
    seen = []
    for it in range(steps):
        v = view_for(it, rank, world, n_views)
        seen.append(v)
        _toy_loss(p, v).backward()
        if overlapped == "ranges":   # the trainer's pipelined form: the table in row ranges, one collective per range
            works = [allreduce_rows_async(p.grad, r0, r1, world) for r0, r1 in row_ranges(p.shape[0], 3)]
            for w in works:
                w.wait()
        elif overlapped:     # the trainer's overlapped form: start, do gradient-independent work, wait
            works = allreduce_grads_async([p], world)
            _ = torch.randn(16).sum()
            wait_all(works)
        else:
            allreduce_grads([p], world)
        opt.step()
        opt.zero_grad(set_to_none=True)
        assert replicas_in_sync(p.data, world)
    torch.save({"p": p.detach().clone(), "seen": seen}, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(120)
@pytest.mark.parametrize("overlapped", [False, True, "ranges"])
def test_two_rank_gradient_allreduce_equals_sum_of_view_gradients(tmp_path, overlapped):
    world, steps, n_views = 2, 4, 7
    mp.spawn(_worker, args=(world, _free_port(), steps, n_views, str(tmp_path), overlapped), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["p"], r1["p"])
    # the ranks take consecutive entries of each epoch's random order (dist_utils.view_order)
    from instascene_amd.dist_utils import view_order
    seq = [view_order(n_views, g // n_views)[g % n_views] for g in range(steps * world)]
    assert r0["seen"] == seq[0::2] and r1["seen"] == seq[1::2]
    assert sorted(view_order(n_views, 0)) == list(range(n_views)) and view_order(n_views, 0) != view_order(n_views, 1)
    assert r0["seen"] != [(i * 2) % n_views for i in range(steps)]          # not the round-robin of earlier rounds
    # single-process reference: accumulate the same views' gradients, same optimiser
    p = torch.nn.Parameter(torch.linspace(-1, 1, 64 * 8).reshape(64, 8).clone())
    opt = torch.optim.Adam([p], lr=0.025, eps=1e-15)
    for it in range(steps):
        for r in range(world):
            _toy_loss(p, view_for(it, r, world, n_views)).backward()   # .grad accumulates = sum
        opt.step()
        opt.zero_grad(set_to_none=True)
    torch.testing.assert_close(r0["p"], p.detach(), rtol=1e-6, atol=1e-7)


def test_view_sharding_is_a_partition():
    n_views, world = 16, 8
    for step in range(4):
        views = [view_for(step, r, world, n_views) for r in range(world)]
        assert len(set(views)) == world
    assert sorted(view_for(s, r, world, n_views) for s in range(2) for r in range(world)) == list(range(16))


def _bucket_worker(rank, world, port, out):
    from instascene_amd.dist_utils import allreduce_bucket
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7 + rank)
    shapes = [(51, 3), (51, 16, 3), (51, 1), (51, 2), (51, 4)]           # the five gradients of a train.py-style step, ODD row count
    ts = [torch.randn(s, generator=g) for s in shapes]
    if rank == 1:
        ts[2] = None                                                     # a rank without a gradient for one tensor
    like = [torch.empty(s) for s in shapes]
    summed = allreduce_bucket(ts, world, like=like)
    # every segment starts on a 16-byte boundary of the flat buffer (iso_gaussian_adam_step reads the views with float4 loads)
    assert all(t.storage_offset() % 4 == 0 for t in summed), [t.storage_offset() for t in summed]
    torch.save({"sum": [t.clone() for t in summed], "mine": [None if t is None else t.clone() for t in ts]},
               os.path.join(out, f"b{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_bucket_allreduce_equals_the_sum_of_the_tensors(tmp_path):
    """RgbTrainer's exchange: the six parameter groups' gradients as ONE flat collective; a missing gradient counts as
    zeros; every rank receives identical sums."""
    world = 2
    mp.spawn(_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "b0.pt"), torch.load(tmp_path / "b1.pt")
    for i in range(5):
        want = a["mine"][i] + (b["mine"][i] if b["mine"][i] is not None else 0)
        assert torch.equal(a["sum"][i], b["sum"][i])
        assert torch.equal(a["sum"][i], want)
    # one rank: the input comes back untouched
    from instascene_amd.dist_utils import allreduce_bucket
    t = [torch.ones(3), None]
    assert allreduce_bucket(t, 1) == t


def _shard_worker(rank, world, port, out):
    from instascene_amd.dist_utils import all_gather_rows, reduce_scatter_rows, shard_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(11 + rank)
    t = torch.randn(12, 8, generator=g)
    mine0 = t.clone()
    r0, r1 = shard_rows(12, rank, world)
    shard = reduce_scatter_rows(t, rank, world).clone()            # the sum over ranks of rows [r0, r1)
    table = torch.full((12, 8), float("nan"))
    table[r0:r1] = shard * 2.0                                     # "the owner's update of its rows"
    all_gather_rows(table, rank, world)
    torch.save({"mine": mine0, "shard": shard, "table": table, "rows": (r0, r1)}, os.path.join(out, f"s{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_reduce_scatter_and_all_gather_of_row_shards(tmp_path):
    """The two halves of the sharded tail's exchange (dist_utils.reduce_scatter_rows / all_gather_rows; gloo has no
    reduce-scatter, so the helper all-reduces and slices): a rank's shard is the sum of those rows over the ranks, and after the
    gather every rank holds every owner's rows."""
    world = 3
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"s{r}.pt") for r in range(world)]
    total = sum(r["mine"] for r in rs)
    owners = torch.cat([r["shard"] for r in rs])                   # every row as its owner received it
    torch.testing.assert_close(owners, total, rtol=0, atol=1e-6)   # (the collective's order of additions is its own)
    for r in rs:
        assert torch.equal(r["table"], owners * 2.0)               # ... and exactly those rows everywhere after the gather
    assert [r["rows"] for r in rs] == [(0, 4), (4, 8), (8, 12)]
