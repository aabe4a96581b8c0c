"""GPU: the PLAIN relative error of the pose gradients where the pixel sum is well conditioned (round-2 verdict:
the suite's yardstick `1e-4 x sum |per-pixel term|` is an fp32-summation bound, not "1e-4 relative").  A component
is well conditioned when |sum| > 0.1 sum |terms| -- which a random-sign upstream gradient never produces (its sums
are residuals of cancellation), so the probe is the upstream gradient ONES, C1's other probe in SURVEY 8d; on
those components the HIP gradient must agree with the float64 oracle
(evaluated on the HIP depth image, as the reference's backward is evaluated on its own forward's output:
sdf_renderer_cuda.cu:334-467, simple_renderer.py:317-458) to 1e-4 of its own magnitude, at C1, C2 and C3."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def check(plan, views, d_hip, g_np, sdf_np, pos, quat, isc, W, H, f, name):
    assert np.all(g_np == 1.0)
    ref = oracle.render_backward(g_np[views], d_hip[views], sdf_np, pos[views], quat[views], isc[views], W / 2, H / 2,
                                 f, f, dtype=np.float64)
    dimg = oracle.render_derivative_images(d_hip[views], sdf_np, pos[views], quat[views], isc[views], W / 2, H / 2, f,
                                           f, dtype=np.float64)
    l1 = np.abs(dimg * g_np[views][:, :, :, None]).sum(axis=(1, 2))
    hip = np.concatenate([plan.g_pos.cpu().numpy()[views], plan.g_quat.cpu().numpy()[views],
                          plan.g_inv_scale.cpu().numpy()[views][:, None]], axis=1).astype(np.float64)
    rp = np.concatenate([ref[1], ref[2], ref[3][:, None]], axis=1)
    well = np.abs(rp) > 0.1 * l1
    assert well.sum() >= max(2, len(views)), f"{name}: too few well-conditioned components ({well.sum()})"
    rel = np.abs(hip - rp)[well] / np.abs(rp[well])
    assert rel.max() <= 1e-4, f"{name}: plain relative error {rel.max():.3e} on a well-conditioned component"
    # and the summation bound everywhere
    assert np.all(np.abs(hip - rp) <= 1e-4 * l1), name
    return rel.max(), int(well.sum())


@pytest.mark.parametrize("name,W,H", [("C1", 160, 120), ("C2", 640, 480)])
def test_single_view_configs(name, W, H):
    from sdfest_amd import BatchRenderPlan, Camera
    f = W / 2.0
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    sdf_np = oracle.blobs_sdf(0)
    pos, quat, isc = np.array([[0, 0, -1.5]], np.float32), np.array([[0, 0, 0, 1]], np.float32), np.array([2.0], np.float32)
    g_np = np.ones((1, H, W), np.float32)
    plan = BatchRenderPlan(64, 1, cam)
    sdf = dev(sdf_np)
    d = plan.forward(sdf, dev(pos), dev(quat), dev(isc), 0.005, prepare_backward=True)
    plan.backward(dev(g_np), sdf, dev(pos), dev(quat), dev(isc))
    check(plan, [0], d.cpu().numpy(), g_np, sdf_np, pos.astype(np.float64), quat.astype(np.float64),
          isc.astype(np.float64), W, H, f, name)


def test_c3_batch():
    from sdfest_amd import BatchRenderPlan, Camera
    B, W, H, f = 256, 640, 480, 320.0
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    sdf_np = oracle.blobs_sdf(0)
    pos, quat, isc = oracle.random_poses(B, seed=1, width=W, height=H, f=f)
    g = torch.ones((B, H, W), device="cuda")
    plan = BatchRenderPlan(64, B, cam)
    sdf = dev(sdf_np)
    pose = (dev(pos), dev(quat), dev(isc))
    d = plan.forward(sdf, *pose, 0.005, prepare_backward=True)
    plan.backward(g, sdf, *pose)
    views = list(range(8))
    d_hip = d[:8].cpu().numpy()
    g_np = g[:8].cpu().numpy()
    check(plan, views, d_hip, g_np, sdf_np, pos.astype(np.float64), quat.astype(np.float64), isc.astype(np.float64),
          W, H, f, "C3")
    # ALL 256 views, every component against the largest of its group (position / quaternion / scale) -- 1e-4 of that
    # where float arithmetic can reach it.  Where a group's largest component is itself a residual of cancellation it
    # cannot: the FLOOR is the oracle's float build (per-pixel terms in float, summed EXACTLY in double) -- the formulas
    # of sdf_renderer_cuda.cu:391-466 in the type the reference computes them in -- against their float64 values.  With
    # this seed its worst view is at 2.4e-4, the kernel's at 1.6e-4 (other views: the roundings are independent): the
    # kernel's terms round no worse than a plain float evaluation's, and its fixed-order float partial sums (wave_sum8
    # -> wave_part -> pose_reduce_kernel) add nothing visible.  Asserted: every view within 1e-4 or within 1.5 x the
    # floor's WORST view, and at most a handful of views above 1e-4 at all.
    d_all, g_all = d.cpu().numpy(), np.ones((B, H, W), np.float32)
    p64, q64, i64 = pos.astype(np.float64), quat.astype(np.float64), isc.astype(np.float64)
    r64 = oracle.render_backward(g_all, d_all, sdf_np, p64, q64, i64, W / 2, H / 2, f, f, dtype=np.float64)
    r32 = oracle.render_backward(g_all, d_all, sdf_np, pos.astype(np.float32), quat.astype(np.float32),
                                 isc.astype(np.float32), W / 2, H / 2, f, f, dtype=np.float32)
    cat = lambda r: np.concatenate([r[1], r[2], r[3][:, None]], axis=1).astype(np.float64)
    ref, flo = cat(r64), cat(r32)
    hip = np.concatenate([plan.g_pos.cpu().numpy(), plan.g_quat.cpu().numpy(), plan.g_inv_scale.cpu().numpy()[:, None]],
                         axis=1).astype(np.float64)
    e_hip, e_flo = np.zeros(B), np.zeros(B)
    for s_ in (slice(0, 3), slice(3, 7), slice(7, 8)):
        scale = np.max(np.abs(ref[:, s_]), axis=1, keepdims=True)
        e_hip = np.maximum(e_hip, np.max(np.abs(hip[:, s_] - ref[:, s_]) / scale, axis=1))
        e_flo = np.maximum(e_flo, np.max(np.abs(flo[:, s_] - ref[:, s_]) / scale, axis=1))
    assert e_flo.max() > 1e-5, "the floor is not vacuous: float terms do cost something in the worst view"
    assert e_hip.max() <= max(1e-4, 1.5 * e_flo.max()), \
        f"C3: worst view {e_hip.max():.3e}: above 1e-4 and above 1.5 x the float floor's worst view {e_flo.max():.3e}"
    assert (e_hip > 1e-4).sum() <= max(2, 2 * (e_flo > 1e-4).sum()), ((e_hip > 1e-4).sum(), (e_flo > 1e-4).sum())
    assert np.median(e_hip) <= 2e-6 and np.median(e_hip) <= 3 * np.median(e_flo) + 1e-7, (np.median(e_hip), np.median(e_flo))
