"""GPU: seeded random configurations of the loop's one-launch render step (sdfr_render_step_fused_l1_pc) -- image sizes
that are no multiple of a tile, odd grid resolutions up to 128, off-centre and anisotropic intrinsics, thresholds, objects
from far away to filling the screen or off it, 1 .. 8 views, both d/dSDF weightings, holes in the observed images --
against the TWO launches it replaces (sdfr_render_step_forward_l1 + sdfr_render_step_backward_l1_pc), which
tests/test_render_fuzz_gpu.py / test_render_l1_gpu.py / test_pc_loss_gpu.py hold against the float64 oracle.

Compared: the depth image (bit for bit), the views' overlap counts and loss sums, d/dSDF as  point-cloud volume + k x
depth volume  (1e-5 of the two-launch volume's largest entry: only where k multiplies differs), the tiles' pose sums times
k against the two-launch step's (the same yardstick), the sampler's per-block pose sums and losses (bit for bit: the same
blocks in the same order)."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def draw(seed):
    rng = np.random.default_rng(7000 + seed)
    R = int(rng.choice([8, 17, 32, 33, 64, 64, 100, 128]))
    B = int(rng.choice([1, 1, 2, 3, 5, 8]))
    W, H = int(rng.integers(17, 330)), int(rng.integers(9, 250))
    f = W * rng.uniform(0.45, 1.4)
    fx, fy = f, f * rng.uniform(0.8, 1.25)
    cx, cy = W / 2 + rng.uniform(-0.2, 0.2) * W, H / 2 + rng.uniform(-0.2, 0.2) * H
    thr = float(rng.choice([0.001, 0.005, 0.02]))
    pos, quat, isc = oracle.random_poses(B, seed=seed, width=W, height=H, f=f)
    pos = (pos * rng.uniform(0.5, 1.3, (B, 1))).astype(np.float32)
    isc = (isc * rng.uniform(0.6, 2.5, B)).astype(np.float32)
    if B >= 3:
        pos[1, 0] += 50.0            # one object off screen: an empty rectangle, count 0
    sdf = oracle.blobs_sdf(int(rng.integers(0, 3)), R=R).astype(np.float32)
    return dict(R=R, B=B, W=W, H=H, fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy), thr=thr, pos=pos, quat=quat,
                isc=isc, sdf=sdf, rng=rng)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SDFR_FUZZ_SEEDS", "16"))))
def test_random_configuration(seed):
    from sdfest_amd import BatchRenderPlan, Camera, _lib
    L = _lib.lib()
    c = draw(seed)
    B, W, H, R, rng = c["B"], c["W"], c["H"], c["R"], c["rng"]
    mode = seed % 2
    shape = seed % 3 != 2                # every third seed: nobody wants d/dSDF
    T = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    camera = Camera(W, H, c["fx"], c["fy"], c["cx"], c["cy"], pixel_center=0.5)
    sdf, pos, quat, isc = T(c["sdf"]), T(c["pos"]), T(c["quat"]), T(c["isc"])
    scale = (1.0 / isc).contiguous()
    # observed images: the render of perturbed poses with holes; their points
    plain = BatchRenderPlan(R, B, camera)
    tgt = plain.forward(sdf, pos + T(rng.normal(0, 0.01, c["pos"].shape)), quat, isc, c["thr"]).clone()
    tgt = torch.where(T(rng.uniform(size=tuple(tgt.shape))) < 0.1, torch.zeros_like(tgt), tgt).contiguous()
    pts, offs = [], [0]
    for v in range(B):
        p = oracle.depth_to_pointcloud(tgt[v].cpu().numpy(), c["fx"], c["fy"], c["cx"] - 0.5, c["cy"] - 0.5)
        if len(p) == 0:
            p = np.array([[0.0, 0.0, -1.0]], dtype=np.float32)          # (a view without a point: give it one)
        pts.append(p)
        offs.append(offs[-1] + len(p))
    points, offsets = T(np.concatenate(pts)), T(offs, torch.int32)
    max_pts = int(max(np.diff(offs)))
    w_depth, w_pc = float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 4.0))
    nbytes = max(L.sdfr_pc_loss_backward_workspace_bytes(B, max_pts), 256)

    # the two launches
    two = BatchRenderPlan(R, B, camera, sdf_grad_mode=mode)
    ws_two = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    depth_two, _ = two.forward_l1(sdf, pos, quat, isc, c["thr"], tgt, prepare_backward=True, defer_loss=True)
    depth_two = depth_two.clone()
    g_two = two.backward_l1_pc(tgt, sdf, pos, quat, isc, scale, points, offsets, max_pts, ws_two, weight=w_depth,
                               pc_weight=w_pc).clone()
    torch.cuda.synchronize()
    stats = two.loss_stats.cpu().numpy().astype(np.float64)            # (sum, count) per view
    # the one launch
    one = BatchRenderPlan(R, B, camera, sdf_grad_mode=mode)
    ws_one = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    g_pc = torch.zeros((R, R, R), device="cuda") if shape else None
    depth_one = one.step_fused_l1_pc(sdf, pos, quat, isc, scale, c["thr"], tgt, points, offsets, max_pts, ws_one,
                                     pc_weight=w_pc, g_sdf=g_pc)
    torch.cuda.synchronize()
    assert torch.equal(depth_one, depth_two)
    cnt = one.view_count.cpu().numpy().astype(np.float64)
    np.testing.assert_array_equal(cnt, stats[:, 1])
    k = np.where(cnt > 0, w_depth / np.maximum(cnt, 1.0), 0.0)
    ntx, nty = (W + 31) // 32, (H + 7) // 8
    lo = L.sdfr_render_fused_tile_loss_offset(R, B, W, H)
    rec = one.workspace[lo:lo + B * ntx * nty * 32].view(torch.float32).view(B, nty * ntx, 8).cpu().numpy().astype(np.float64)
    np.testing.assert_array_equal(rec[:, :, 1].sum(axis=1), cnt)
    np.testing.assert_allclose(rec[:, :, 0].sum(axis=1), stats[:, 0], rtol=1e-5, atol=1e-7)
    # the sampler's blocks: the very same work in both launches
    assert torch.equal(ws_one, ws_two)
    # the tiles' pose sums
    po = one.partials_offset
    assert po == two.partials_offset
    part = lambda plan: plan.workspace[po:po + B * ntx * nty * 32].view(torch.float32).view(B, nty * ntx, 8).cpu().numpy().astype(np.float64)
    p_one, p_two = part(one).sum(axis=1) * k[:, None], part(two).sum(axis=1)
    mag = np.abs(part(two)).sum(axis=1) + 1e-30                        # the fp32-summation yardstick: sum of |tile sums|
    assert np.all(np.abs(p_one - p_two) <= 1e-5 * mag), (seed, p_one, p_two)
    if shape:
        g = g_pc.cpu().numpy().astype(np.float64) + np.tensordot(k, one.g_depth.cpu().numpy().astype(np.float64), 1)
        ref = g_two.cpu().numpy().astype(np.float64)
        top = np.abs(ref).max()
        assert np.abs(g - ref).max() <= 1e-5 * top + 1e-30, (seed, np.abs(g - ref).max() / max(top, 1e-30))
