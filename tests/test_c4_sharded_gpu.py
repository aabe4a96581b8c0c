"""GPU: BASELINE configs[3] (views sharded over ranks, one all-reduce of d/dSDF) in miniature: two ranks on the
test box's one GPU against a single-process run of all the views."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("n_views,W,H", [(10, 320, 240), (7, 160, 120)])
def test_two_ranks_equal_one_process(tmp_path, n_views, W, H):
    from sdfest_amd import BatchRenderPlan, Camera
    from sdfest_amd.parallel import spawn_ranks
    from sdfest_amd.synthetic import blobs_sdf, random_poses
    out = str(tmp_path / "c4.npz")
    rc = spawn_ranks([sys.executable, os.path.join(HERE, "_c4_worker.py"), out, str(n_views), str(W), str(H)], 2,
                     timeout=240)
    assert rc == 0
    r = np.load(out)
    assert r["spans"].tolist() == [[0, (n_views + 1) // 2], [(n_views + 1) // 2, n_views]]   # contiguous shards
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    pos, quat, isc = random_poses(n_views, seed=1, width=W, height=H, f=W / 2.0)
    sdf = t(blobs_sdf(0))
    g_all = (torch.rand((n_views, H, W), generator=torch.Generator().manual_seed(77)) * 2 - 1).to(dev)
    cam = Camera(W, H, W / 2.0, W / 2.0, W / 2.0, H / 2.0, pixel_center=0.5)
    plan = BatchRenderPlan(64, n_views, cam, device=dev)
    depth = plan.forward(sdf, t(pos), t(quat), t(isc), 0.005)
    assert np.array_equal(depth.cpu().numpy(), r["depth"]) and (r["depth"] > 0).sum() > 500 * n_views
    g_sdf, g_pos, g_quat, g_is = plan.backward(g_all.contiguous(), sdf, t(pos), t(quat), t(isc))
    # per-view gradients do not depend on how the views are sharded (fixed-order sums per view)
    assert np.array_equal(g_pos.cpu().numpy(), r["g_pos"]) and np.array_equal(g_quat.cpu().numpy(), r["g_quat"])
    assert np.array_equal(g_is.cpu().numpy(), r["g_is"])
    # the shared gradient: sum over ranks == the single launch, up to the order of float atomics
    ref = g_sdf.cpu().numpy()
    assert np.max(np.abs(ref - r["g_sdf"])) <= 1e-5 * np.max(np.abs(ref))


@pytest.mark.parametrize("n_views,W,H", [(9, 320, 240)])
def test_two_ranks_equal_one_process_bitwise_in_the_deterministic_mode(tmp_path, n_views, W, H):
    """SDFR_SDF_GRAD_DETERMINISTIC: the ranks add their int64 fixed-point volumes and convert afterwards -- the shared
    gradient is then BITWISE the single-process result, whatever the split."""
    from sdfest_amd import BatchRenderPlan, Camera
    from sdfest_amd.parallel import spawn_ranks
    from sdfest_amd.synthetic import blobs_sdf, random_poses
    out = str(tmp_path / "c4det.npz")
    rc = spawn_ranks([sys.executable, os.path.join(HERE, "_c4_worker.py"), out, str(n_views), str(W), str(H), "det"],
                     2, timeout=240)
    assert rc == 0
    r = np.load(out)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    pos, quat, isc = random_poses(n_views, seed=1, width=W, height=H, f=W / 2.0)
    sdf = t(blobs_sdf(0))
    g_all = (torch.rand((n_views, H, W), generator=torch.Generator().manual_seed(77)) * 2 - 1).to(dev)
    cam = Camera(W, H, W / 2.0, W / 2.0, W / 2.0, H / 2.0, pixel_center=0.5)
    plan = BatchRenderPlan(64, n_views, cam, device=dev, sdf_grad_mode=0x100)
    plan.forward(sdf, t(pos), t(quat), t(isc), 0.005)
    g_sdf = plan.backward(g_all.contiguous(), sdf, t(pos), t(quat), t(isc))[0]
    ref = g_sdf.cpu().numpy()
    assert np.abs(ref).max() > 0 and np.array_equal(ref, r["g_sdf"])


def test_half_of_c4_at_the_real_shard_size(tmp_path):
    """BASELINE configs[3] is 2048 views as 8 shards of 256 over RCCL; a one-GPU box can hold half of it: FOUR ranks
    (gloo, sharing the GPU) of 256 views of 640x480 each against ONE process rendering the 1024 views -- contiguous
    shards of the real size, the real image size, the batch kernels, the integer exchange: depth images equal
    (checksums of their bits and hit counts), per-view pose gradients bit for bit, and the summed d/dSDF BITWISE the
    single-process volume (deterministic mode)."""
    from sdfest_amd import BatchRenderPlan, Camera
    from sdfest_amd.parallel import spawn_ranks
    from sdfest_amd.synthetic import blobs_sdf, random_poses
    n_views, W, H = 1024, 640, 480
    out = str(tmp_path / "c4half.npz")
    rc = spawn_ranks([sys.executable, os.path.join(HERE, "_c4_worker.py"), out, str(n_views), str(W), str(H), "det_light"],
                     4, timeout=500)
    assert rc == 0
    r = np.load(out)
    assert r["spans"].tolist() == [[256 * k, 256 * (k + 1)] for k in range(4)]
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    pos, quat, isc = random_poses(n_views, seed=1, width=W, height=H, f=W / 2.0)
    sdf = t(blobs_sdf(0))
    g_all = torch.rand((n_views, H, W), generator=torch.Generator().manual_seed(77)) * 2 - 1
    cam = Camera(W, H, W / 2.0, W / 2.0, W / 2.0, H / 2.0, pixel_center=0.5)
    plan = BatchRenderPlan(64, n_views, cam, device=dev, sdf_grad_mode=0x100)
    depth = plan.forward(sdf, t(pos), t(quat), t(isc), 0.005)
    bits = depth.view(torch.int32).to(torch.int64).sum(dim=(1, 2)).cpu().numpy()
    hits = (depth > 0).sum(dim=(1, 2)).cpu().numpy()
    assert np.array_equal(bits, r["depth"][0]) and np.array_equal(hits, r["depth"][1]) and hits.sum() > 10_000_000
    g_sdf, g_pos, g_quat, g_is = plan.backward(g_all.to(dev).contiguous(), sdf, t(pos), t(quat), t(isc))
    assert np.array_equal(g_pos.cpu().numpy(), r["g_pos"]) and np.array_equal(g_quat.cpu().numpy(), r["g_quat"])
    assert np.array_equal(g_is.cpu().numpy(), r["g_is"])
    ref = g_sdf.cpu().numpy()
    assert np.abs(ref).max() > 0 and np.array_equal(ref, r["g_sdf"])


def test_c4_full_size_equals_the_sum_of_its_eight_shards():
    """All of BASELINE configs[3] on one GPU, one shard after the other: the 2048 seeded views of 640x480 as ONE batch
    against their 8 contiguous shards of 256 (what each of the 8 ranks renders) -- depth images and per-view pose
    gradients bit for bit; the shards' int64 d/dSDF volumes, added as the integer all-reduce adds them and converted
    once (sdfr_fixed_to_float), bitwise the full batch's volume.  What this cannot show is the exchange itself."""
    from sdfest_amd import BatchRenderPlan, Camera
    from sdfest_amd.parallel import allreduce_fixed_gradients, shard_views
    from sdfest_amd.synthetic import blobs_sdf, random_poses
    n_views, W, H, world = 2048, 640, 480, 8
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    pos, quat, isc = (t(a) for a in random_poses(n_views, seed=1, width=W, height=H, f=W / 2.0))
    sdf = t(blobs_sdf(0))
    cam = Camera(W, H, W / 2.0, W / 2.0, W / 2.0, H / 2.0, pixel_center=0.5)
    gen = torch.Generator(device=dev).manual_seed(5)
    g_all = torch.rand((n_views, H, W), device=dev, generator=gen) * 2 - 1
    full = BatchRenderPlan(64, n_views, cam, device=dev, sdf_grad_mode=0x100)
    depth = full.forward(sdf, pos, quat, isc, 0.005)
    bits_full = depth.view(torch.int32).to(torch.int64).sum(dim=(1, 2))
    g_sdf, g_pos, g_quat, g_is = (x.clone() for x in full.backward(g_all, sdf, pos, quat, isc))
    assert int((depth > 0).sum()) > 25_000_000
    del full, depth
    torch.cuda.empty_cache()
    shard = BatchRenderPlan(64, n_views // world, cam, device=dev, sdf_grad_mode=0x100)
    total = torch.zeros((64, 64, 64), dtype=torch.int64, device=dev)
    for r in range(world):
        b, e = shard_views(n_views, r, world)
        assert (b, e) == (256 * r, 256 * (r + 1))
        d = shard.forward(sdf, pos[b:e].contiguous(), quat[b:e].contiguous(), isc[b:e].contiguous(), 0.005)
        assert torch.equal(d.view(torch.int32).to(torch.int64).sum(dim=(1, 2)), bits_full[b:e])
        _, sp, sq, si = shard.backward(g_all[b:e], sdf, pos[b:e].contiguous(), quat[b:e].contiguous(), isc[b:e].contiguous())
        assert torch.equal(sp, g_pos[b:e]) and torch.equal(sq, g_quat[b:e]) and torch.equal(si, g_is[b:e])
        total += shard.g_sdf_fixed()
    out = torch.empty((64, 64, 64), device=dev)
    allreduce_fixed_gradients(total, out)        # (no process group: the conversion alone)
    assert out.abs().max() > 0 and torch.equal(out, g_sdf)


@pytest.mark.parametrize("exchange", ["ring", "sync"])
def test_bench_py_with_two_ranks_in_rehearsal_mode(tmp_path, exchange):
    """bench.py --gpus 2 itself, both ranks on the one GPU of the test box over gloo (marked REHEARSAL in its line):
    the rank spawn, the start-up watchdog, the pose shards, both exchanges of the gradient volumes -- batched over
    the ring, and finished after every step before the next forward -- and the single result line with both values."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, SDFR_BENCH_SHARE_GPU="1", SDFR_BENCH_BACKEND="gloo", SDFR_BENCH_GRAD_VOLUMES="4")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                          "--batch", "24", "--width", "320", "--height", "240", "--prewarm-ms", "0",
                          "--exchange", exchange],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "stage: init_process_group" in res.stderr and "WATCHDOG" not in res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    # the finished headline goes out BEFORE the multi-rank extras (a native abort inside one must not cost the
    # measurement), and again, augmented, at the end: readers take the last line
    assert len(lines) == 2, res.stdout
    first, line = json.loads(lines[0]), json.loads(lines[1])
    assert "loop_sharded" not in first and {k: v for k, v in line.items() if k != "loop_sharded"} == first
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["value"] > 0
    assert "REHEARSAL" in line["config"]["workload"]
    col = line["collective"]
    assert col["exchange_of_value"] == exchange
    assert line["value_ring_exchange"] > 0 and line["value_sync_exchange"] > 0
    assert line["value"] == line[f"value_{exchange}_exchange"]
    assert col["allreduce_us"]["n"] == 5 and col["allreduce_us"]["median"] > 0
    assert col["world_size_seen"] == 2 and col["backend"] == "gloo"
    assert sorted(r["rank"] for r in col["per_rank"]) == [0, 1]
    assert all(r["hit_pixels"] > 1000 and r["prologue_fallbacks"] == 0 for r in col["per_rank"])
    assert col["per_rank"][0]["hit_pixels"] != col["per_rank"][1]["hit_pixels"]      # different pose shards
    # N > 1 also measures the render-and-compare LOOP sharded over the ranks (its one all-reduce inside an iteration)
    rows = line["loop_sharded"]
    assert [r["views"] for r in rows] == [16, 128] and [r["views_per_rank"] for r in rows] == [8, 64]
    for r in rows:
        assert r["ms_per_iteration_sdf"] > 0 and r["ms_per_iteration_latent"] > 0
        assert r["single_rank_ms_same_views_per_rank"] > 0
        assert r["final_position_error_mm_sdf"] < 5.0 and r["final_position_error_mm_latent"] < 5.0
    # the sync exchange read against its own parts (bench.py: collective.sync_exchange_model)
    m = col["sync_exchange_model"]
    assert m["compute_ms_per_step_events"] > 0 and m["allreduce_us_measured_median"] > 0
    assert abs(m["sum_ms_per_step"] - (m["compute_ms_per_step_events"] + m["allreduce_us_measured_median"] * 1e-3)) < 1e-3
    assert m["allreduce_us_predicted_xgmi_ring"] > 0


def test_bench_py_with_five_ranks_in_rehearsal_mode():
    """bench.py --gpus 5 on the one GPU of the test box over gloo, tiny images: as many ranks as this pool's process
    guard lets share a card beside the test runner itself (six processes per GPU: an 8-rank rehearsal is not possible
    here; the 8-GPU run itself is the driver's).  The spawn, the rendezvous, five pose shards, the exchange, the extras:
    the final line carries `collective` with a per-rank list of 5 and `loop_sharded`."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, SDFR_BENCH_SHARE_GPU="1", SDFR_BENCH_BACKEND="gloo", SDFR_BENCH_GRAD_VOLUMES="4")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "5", "--steps", "3", "--warmup", "1",
                          "--batch", "8", "--width", "160", "--height", "120", "--prewarm-ms", "0"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2, res.stdout
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 5 and "REHEARSAL" in line["config"]["workload"] and "abandoned_sections" not in line
    col = line["collective"]
    assert col["world_size_seen"] == 5 and sorted(r["rank"] for r in col["per_rank"]) == list(range(5))
    assert len({r["hit_pixels"] for r in col["per_rank"]}) > 1
    rows = line["loop_sharded"]
    assert [r["views"] for r in rows] == [40, 320] and all(r["ms_per_iteration_sdf"] > 0 for r in rows)
    assert "loop_sharded_collective_in_graph" not in line     # (opt-in: SDFR_BENCH_GRAPH_COLLECTIVE=1, RCCL only)
