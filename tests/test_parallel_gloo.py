"""CPU, world_size 2, gloo: view sharding + the single all-reduce of the shared gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sdfest_amd.parallel import allreduce_shared_gradients, shard_views


def test_shard_views_partitions_exactly():
    for n in (0, 1, 7, 256, 2048, 2049):
        for w in (1, 2, 3, 8):
            spans = [shard_views(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_views(2048, 3, 8) == (768, 1024)
    with pytest.raises(ValueError):
        shard_views(4, 2, 2)


def test_allreduce_is_noop_when_not_distributed():
    g = torch.arange(8.0)
    assert allreduce_shared_gradients(g) is g and torch.equal(g, torch.arange(8.0))
    assert allreduce_shared_gradients(g, async_op=True).wait() and torch.equal(g, torch.arange(8.0))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # each rank owns a shard of 10 "views"; its local d/dSDF is the sum of its views' terms
        b, e = shard_views(10, rank, world)
        rng = np.random.default_rng(0)
        per_view = rng.normal(size=(10, 4, 4, 4)).astype(np.float32)
        pose = rng.normal(size=(10, 8)).astype(np.float32)
        g_sdf = torch.tensor(per_view[b:e].sum(0))
        g_pose = torch.tensor(pose[b:e].sum(0))
        allreduce_shared_gradients(g_sdf, extra=[g_pose])
        ok1 = np.allclose(g_sdf.numpy(), per_view.sum(0), atol=1e-5)
        ok2 = np.allclose(g_pose.numpy(), pose.sum(0), atol=1e-5)
        g2 = torch.tensor(per_view[b:e].sum(0))
        allreduce_shared_gradients(g2)
        ok3 = np.allclose(g2.numpy(), per_view.sum(0), atol=1e-5)
        # the asynchronous form bench.py uses to run the exchange beside the next forward
        g3 = torch.tensor(per_view[b:e].sum(0))
        handle = allreduce_shared_gradients(g3, async_op=True)
        other = torch.ones(3) * 2          # unrelated work between issue and wait
        handle.wait()
        ok4 = np.allclose(g3.numpy(), per_view.sum(0), atol=1e-5) and float(other.sum()) == 6.0
        out[rank] = int(ok1 and ok2 and ok3 and ok4)
    finally:
        dist.destroy_process_group()


def test_allreduce_shared_gradients_world2():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert list(out) == [1, 1]


def _bucket_worker(rank, world, port, out):
    """the sharded loop's exchange (sdfest_amd.parallel.allreduce_bucket) on CPU tensors over gloo"""
    from sdfest_amd.parallel import allreduce_bucket, resolve_group
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        group, r, w = resolve_group("world")
        ok = (r, w) == (rank, world) and resolve_group(None) == (None, 0, 1)
        V, REC = 5, 20
        b, e = shard_views(V, rank, world)
        rng = np.random.default_rng(3)
        full = rng.normal(size=(V, REC)).astype(np.float32)
        full[1, 16] = np.nan              # an empty overlap's loss
        full[3, 2] = -0.0                 # a sign only the integer sum keeps
        full[4, 7] = np.float32(1e-42)    # a denormal
        vol = rng.integers(-2 ** 40, 2 ** 40, size=64, dtype=np.int64)
        # integer bucket: [int64 volume | float records reinterpreted]: exact, whatever the records hold
        rec = np.zeros((V, REC), np.float32)
        rec[b:e] = full[b:e]
        bucket = torch.cat([torch.tensor(vol * (rank + 1)), torch.tensor(rec).view(-1).view(torch.int64)])
        allreduce_bucket(bucket, group, integer=True)
        got_rec = bucket[64:].view(torch.float32).view(V, REC).numpy()
        ok = ok and np.array_equal(bucket[:64].numpy(), vol * sum(range(1, world + 1)))
        ok = ok and np.array_equal(got_rec.view(np.uint32), full.view(np.uint32))      # bit for bit, NaN and -0 included
        # float bucket: sums of a record with zeros reproduce it (up to the sign of zero)
        fb = torch.cat([torch.full((8,), float(rank + 1)), torch.tensor(rec).view(-1)])
        allreduce_bucket(fb, group, integer=False)
        got = fb[8:].view(V, REC).numpy()
        ok = ok and np.all(fb[:8].numpy() == sum(range(1, world + 1)))
        ok = ok and np.array_equal(got, full, equal_nan=True)
        # no group: nothing happens
        x = torch.ones(4)
        ok = ok and allreduce_bucket(x, None) is x and float(x.sum()) == 4.0
        try:
            allreduce_bucket(torch.ones(3), group, integer=True)
            ok = False
        except RuntimeError:
            pass
        dist.barrier()
        out[rank] = int(ok)
    finally:
        dist.destroy_process_group()


def test_allreduce_bucket_world2_is_exact_in_integer_mode():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert list(out) == [1, 1]


def test_resolve_group_needs_an_initialised_process_group():
    from sdfest_amd.parallel import resolve_group
    assert resolve_group(None) == (None, 0, 1)
    with pytest.raises(RuntimeError):
        resolve_group("world")
