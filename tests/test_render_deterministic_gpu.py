"""GPU: the deterministic d/dSDF mode (SDFR_SDF_GRAD_DETERMINISTIC, include/sdfr.h; SURVEY.md section 5 asks for a
deterministic reduction mode; the reference's float atomics are sdf_renderer_cuda.cu:373-388)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import check_sdf_grad, rel_err

pytestmark = pytest.mark.gpu
DET = 0x100


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def setup(B, W, H, f, seed):
    from sdfest_amd import Camera
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    pos, quat, isc = oracle.random_poses(B, seed=seed, width=W, height=H, f=f)
    g = np.random.default_rng(seed + 7).uniform(-1, 1, (B, H, W)).astype(np.float32)
    return cam, (pos, quat, isc), g


@pytest.mark.parametrize("B,W,H,f", [(256, 200, 136, 300.0),    # batch tiles (per-view shapes)
                                     (12, 320, 240, 160.0),     # small tiles
                                     (3, 640, 480, 320.0)])     # plain grid path of the forward
def test_bitwise_reproducible_and_independent_of_path_and_split(B, W, H, f):
    from sdfest_amd import BatchRenderPlan
    cam, (pos, quat, isc), g = setup(B, W, H, f, seed=21)
    sdf = dev(oracle.blobs_sdf(0))
    pose = (dev(pos), dev(quat), dev(isc))
    gd = dev(g)
    plan = BatchRenderPlan(64, B, cam, sdf_grad_mode=DET)
    ref_plan = BatchRenderPlan(64, B, cam)
    # stand-alone pair, twice
    runs = []
    for _ in range(2):
        plan.forward(sdf, *pose, 0.005)
        gs = plan.backward(gd, sdf, *pose)[0]
        runs.append((gs.clone(), plan.g_sdf_fixed().clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0].abs().max().item() > 0
    # the float volume is the int64 volume times 2^-40
    assert torch.equal(runs[0][0], (runs[0][1].to(torch.float32) * 2.0 ** -40))
    # the step path (other rectangles, other reduction order of the tiles): same bits
    plan.forward(sdf, *pose, 0.005, prepare_backward=True)
    gs_step = plan.backward(gd, sdf, *pose)[0]
    assert torch.equal(gs_step, runs[0][0]) and torch.equal(plan.g_sdf_fixed(), runs[0][1])
    # against the default mode: the same gradient up to the rounding of float atomics / the 2^-40 quantum
    ref_plan.forward(sdf, *pose, 0.005)
    gs_ref = ref_plan.backward(gd, sdf, *pose)[0]
    assert rel_err(runs[0][0].cpu().numpy(), gs_ref.cpu().numpy()) <= 1e-5
    # any split of the views: the int64 volumes of the parts add up to the whole, bit for bit
    if B >= 2:
        cut = B // 3 + 1
        total = torch.zeros_like(runs[0][1])
        for lo, hi in ((0, cut), (cut, B)):
            part = BatchRenderPlan(64, hi - lo, cam, sdf_grad_mode=DET)
            sub = tuple(p[lo:hi].contiguous() for p in pose)
            part.forward(sdf, *sub, 0.005)
            part.backward(gd[lo:hi].contiguous(), sdf, *sub)
            total += part.g_sdf_fixed()
        assert torch.equal(total, runs[0][1])


@pytest.mark.parametrize("mode", [0, 1])
def test_deterministic_mode_against_the_oracle_and_its_limits(mode):
    """mode: the d/dSDF weights underneath the flag -- exact (0) or the reference extension's (1,
    sdf_renderer_cuda.cu:373-388)"""
    from sdfest_amd import BatchRenderPlan, _lib
    B, W, H, f = 6, 160, 120, 80.0
    cam, (pos, quat, isc), g = setup(B, W, H, f, seed=22)
    sdf_np = oracle.blobs_sdf(0)
    sdf = dev(sdf_np)
    pose = (dev(pos), dev(quat), dev(isc))
    plan = BatchRenderPlan(64, B, cam, sdf_grad_mode=DET | mode)
    d = plan.forward(sdf, *pose, 0.005)
    gs = plan.backward(dev(g), sdf, *pose)[0].cpu().numpy()
    ref = oracle.render_backward(g, d.cpu().numpy(), sdf_np, pos, quat, isc, W / 2, H / 2, f, f, dtype=np.float64,
                                 sdf_grad_mode=mode)[0]
    check_sdf_grad(gs, ref, mode, int((d > 0).sum().item()), 1e-4)      # fp32 evaluation of the contributions
    # (bitwise repeatable in this mode as well, on the step path too)
    plan.forward(sdf, *pose, 0.005, prepare_backward=True)
    again = plan.backward(dev(g), sdf, *pose)[0].cpu().numpy()
    assert np.array_equal(gs, again)
    # per-view gradient volumes are not supported in this mode
    per_view = BatchRenderPlan(64, B, cam, per_view_sdf=True, sdf_grad_mode=DET | mode)
    sdfs = dev(np.stack([sdf_np] * B))
    per_view.forward(sdfs, *pose, 0.005)
    with pytest.raises(RuntimeError, match="DETERMINISTIC"):
        per_view.backward(dev(g), sdfs, *pose)
    assert _lib.lib().sdfr_render_fixed_volume_offset(64, B, W, H, 0) > 0
