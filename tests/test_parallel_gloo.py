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
