"""GPU: render + masked depth-L1 in one pass against the unfused HIP sequence
(bit for bit where the arithmetic is the same) and against the CPU oracle."""
import numpy as np
import pytest
import torch

import oracle
from helpers import check_sdf_grad, rel_err

pytestmark = pytest.mark.gpu

REL = 1e-4


@pytest.fixture(scope="module")
def R():
    import sdfest_amd.differentiable_renderer as r
    assert torch.cuda.is_available()
    return r


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def scene(B, W, H, f, seed=1, target_seed=7):
    """B random poses of blobs(0) and 'observed' images = the render of slightly different poses
    with holes (zeros) punched in, so that every case of the overlap mask occurs."""
    sdf = oracle.blobs_sdf(0)
    pos, quat, isc = oracle.random_poses(B, seed=seed, width=W, height=H, f=f)
    rng = np.random.default_rng(target_seed)
    pos_t = pos + rng.normal(0, 0.01, pos.shape).astype(np.float32)
    cam = (W, H, W / 2, H / 2, f, f)
    tgt = oracle.render_forward(sdf, pos_t, quat, isc, *cam, 0.005, dtype=np.float32)
    tgt = np.where(rng.uniform(size=tgt.shape) < 0.1, 0.0, tgt).astype(np.float32)
    return sdf, pos, quat, isc, cam, tgt


def fused(R, sdf, pos, quat, isc, cam, tgt, thr=0.005, weight=1.0, loss_grad=None, per_view=False, mode=0):
    from sdfest_amd.differentiable_renderer import BatchRenderPlan, Camera
    W, H, cx, cy, fx, fy = cam
    camera = Camera(W, H, fx, fy, cx, cy, pixel_center=0.5)
    plan = BatchRenderPlan(sdf.shape[-1], len(pos), camera, per_view_sdf=per_view, sdf_grad_mode=mode)
    a = [dev(sdf), dev(pos), dev(quat), dev(isc)]
    t = dev(tgt)
    depth, loss = plan.forward_l1(*a, thr, t)
    depth, loss, stats = depth.cpu().numpy().copy(), loss.cpu().numpy().copy(), plan.loss_stats.cpu().numpy().copy()
    g = plan.backward_l1(t, *a, weight=weight, loss_grad=None if loss_grad is None else dev(loss_grad))
    return depth, loss, stats, [x.cpu().numpy().copy() for x in g]


def unfused(R, sdf, pos, quat, isc, cam, tgt, thr=0.005, weight=1.0, mode=0):
    """sdfr_render_forward -> sdfr_depth_l1_loss -> sdfr_render_backward."""
    from sdfest_amd import _lib
    L = _lib.lib()
    W, H, cx, cy, fx, fy = cam
    B = len(pos)
    a = [dev(sdf), dev(pos), dev(quat), dev(isc)]
    depth = R.forward_raw(*a, W, H, cx, cy, fx, fy, thr)
    t = dev(tgt)
    loss = torch.empty(B, device="cuda")
    grad = torch.empty_like(depth)
    ws = torch.empty(max(L.sdfr_depth_l1_workspace_bytes(B, W, H), 256), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.sdfr_depth_l1_loss(depth.data_ptr(), t.data_ptr(), B, W, H, weight, loss.data_ptr(),
                                    grad.data_ptr(), ws.data_ptr(), ws.numel(), 0, st), "l1")
    g = R.backward_raw(grad, depth, *a, W, H, cx, cy, fx, fy, mode)
    return depth.cpu().numpy(), loss.cpu().numpy(), [x.cpu().numpy() for x in g]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("B,W,H,f", [(1, 160, 120, 80.0), (3, 640, 480, 320.0), (17, 320, 240, 160.0),
                                     (40, 640, 480, 320.0)])
def test_fused_equals_unfused_and_oracle(R, B, W, H, f, mode):
    """B=1,3: plain-grid small tiles; B=17: packed records; B=40 at 640x480: batch (macro) tiles.  mode 1: the d/dSDF
    weights of the reference's GPU extension (SDF_GRAD_CUDA_COMPAT, sdf_renderer_cuda.cu:373-388)."""
    sdf, pos, quat, isc, cam, tgt = scene(B, W, H, f)
    w = 0.7
    d_f, loss_f, stats, g_f = fused(R, sdf, pos, quat, isc, cam, tgt, weight=w, mode=mode)
    d_u, loss_u, g_u = unfused(R, sdf, pos, quat, isc, cam, tgt, weight=w, mode=mode)
    assert np.array_equal(d_f, d_u)                       # same march
    l_ref, grad_ref = oracle.depth_l1(d_f, tgt, weight=w)  # float64 on the HIP depth
    mask = (tgt > 0) & (d_f > 0)
    assert np.array_equal(stats[:, 1], mask.sum(axis=(1, 2)).astype(np.float32))  # exact counts
    assert mask.sum() > 100 * B
    np.testing.assert_allclose(loss_f, l_ref, rtol=2e-6)
    np.testing.assert_allclose(loss_u, l_ref, rtol=2e-6)
    np.testing.assert_allclose(stats[:, 0] / stats[:, 1], loss_f, rtol=1e-7)
    # the in-kernel gradient image equals the loss kernel's: identical pose sums (fixed order)
    for k in (1, 2, 3):
        assert np.array_equal(g_f[k], g_u[k]), k
    assert rel_err(g_f[0], g_u[0]) <= 1e-5                # float-atomic order only
    # and the oracle's backward on the float64 gradient image
    ob = oracle.render_backward(grad_ref.astype(np.float32), d_f, sdf, pos, quat, isc, *cam[2:], dtype=np.float32,
                                sdf_grad_mode=mode)
    check_sdf_grad(g_f[0], ob[0], mode, int((d_f > 0).sum()), REL)
    for b in range(B):
        dimg = oracle.render_derivative_images(d_f[b], sdf, pos[b], quat[b], isc[b:b + 1], *cam[2:],
                                               dtype=np.float64)[0]
        l1 = np.array([np.sum(np.abs(dimg[..., k] * grad_ref[b])) for k in range(8)])
        pose = np.concatenate([g_f[1][b], g_f[2][b], g_f[3][b:b + 1]])
        ref = np.concatenate([ob[1][b], ob[2][b], ob[3][b:b + 1]])
        assert np.all(np.abs(pose - ref) <= REL * l1), (b, pose, ref, l1)


@pytest.mark.parametrize("B,W,H,f", [(1, 160, 120, 80.0), (3, 640, 480, 320.0), (6, 320, 240, 160.0),
                                     (17, 320, 240, 160.0), (40, 640, 480, 320.0), (256, 640, 480, 320.0)])
def test_loss_fused_step_equals_the_unfused_step_bit_for_bit(R, B, W, H, f):
    """forward_l1(prepare_backward=True) -> backward_l1 as ONE step (sdfr_render_step_forward_l1 /
    sdfr_render_step_backward_l1), with the loss statistics reduced by the forward's own launch or DEFERRED into the
    backward's (no launch between the image kernels), against the unfused step on the same plan geometry: forward(
    prepare_backward=True) -> sdfr_depth_l1_loss -> backward.  Same depth, same loss and statistics, same pose
    gradients, bit for bit; d/dSDF up to the order of its float atomics.  B = 1: the inline set-up (no prologue
    launch at all); 3: the largest inline set-up; 6: plain grid behind the set-up launch; 17: packed records; 40, 256: batch tiles, 256 = the
    benchmark's C3."""
    from sdfest_amd import _lib
    from sdfest_amd.differentiable_renderer import BatchRenderPlan, Camera
    L = _lib.lib()
    if B == 256:     # (the oracle render of 256 observed images would take minutes: the observation is a HIP render)
        sdf = oracle.blobs_sdf(0)
        pos, quat, isc = oracle.random_poses(B, seed=1, width=W, height=H, f=f)
        cam = (W, H, W / 2, H / 2, f, f)
        rng = np.random.default_rng(7)
        pos_t = pos + rng.normal(0, 0.01, pos.shape).astype(np.float32)
        tgt = R.forward_raw(dev(sdf), dev(pos_t), dev(quat), dev(isc), W, H, W / 2, H / 2, f, f, 0.005)
        tgt = torch.where(torch.rand(tgt.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(3)) < 0.1,
                          torch.zeros_like(tgt), tgt).cpu().numpy()
    else:
        sdf, pos, quat, isc, cam, tgt = scene(B, W, H, f)
    camera = Camera(W, H, cam[4], cam[5], cam[2], cam[3], pixel_center=0.5)
    a = [dev(sdf), dev(pos), dev(quat), dev(isc)]
    t = dev(tgt)
    w = 0.7
    # the unfused step (close_views=False: the loss-fused forms ignore the half-grid hint)
    plan_u = BatchRenderPlan(64, B, camera, close_views=False)
    depth_u = plan_u.forward(*a, 0.005, prepare_backward=True)
    loss_u = torch.empty(B, device="cuda")
    grad = torch.empty_like(depth_u)
    ws = torch.empty(max(L.sdfr_depth_l1_workspace_bytes(B, W, H), 256), dtype=torch.uint8, device="cuda")
    _lib.check(L.sdfr_depth_l1_loss(depth_u.data_ptr(), t.data_ptr(), B, W, H, w, loss_u.data_ptr(), grad.data_ptr(),
                                    ws.data_ptr(), ws.numel(), 0, torch.cuda.current_stream().cuda_stream), "l1")
    g_u = [x.clone() for x in plan_u.backward(grad, *a)]
    assert (grad != 0).sum() > 50 * B
    runs = {}
    for deferred in (False, True):
        plan = BatchRenderPlan(64, B, camera, close_views=False)
        for rep in range(2):      # twice on the same plan: the ring of volumes, stale statistics
            plan.loss.fill_(-1.0); plan.loss_stats.fill_(-1.0)
            depth, _ = plan.forward_l1(*a, 0.005, t, prepare_backward=True, defer_loss=deferred)
            if deferred:          # nothing was reduced yet
                assert torch.all(plan.loss == -1.0)
            g = plan.backward_l1(t, *a, weight=w)
            runs[(deferred, rep)] = (depth.clone(), plan.loss.clone(), plan.loss_stats.clone(), [x.clone() for x in g])
        with pytest.raises(RuntimeError, match="same tensors") if deferred else _nullcontext():
            plan.forward_l1(*a, 0.005, t, prepare_backward=True, defer_loss=deferred)
            plan.backward_l1(t, a[0], a[1].clone(), a[2], a[3], weight=w)      # another tensor: not the step's backward
    ref = runs[(False, 0)]
    assert torch.equal(ref[0], depth_u)
    mask = (t > 0) & (depth_u > 0)
    assert torch.equal(ref[2][:, 1], mask.sum(dim=(1, 2)).float())
    np.testing.assert_allclose(ref[1].cpu().numpy(), loss_u.cpu().numpy(), rtol=2e-6)
    for k in (1, 2, 3):
        assert torch.equal(ref[3][k], g_u[k]), k
    assert rel_err(ref[3][0].cpu().numpy(), g_u[0].cpu().numpy()) <= 1e-5
    for key, r in runs.items():
        assert torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1]) and torch.equal(r[2], ref[2]), key
        for k in (1, 2, 3):
            assert torch.equal(r[3][k], ref[3][k]), (key, k)
        assert rel_err(r[3][0].cpu().numpy(), ref[3][0].cpu().numpy()) <= 1e-5, key


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def test_per_view_weights_and_reproducibility(R):
    sdf, pos, quat, isc, cam, tgt = scene(5, 320, 240, 160.0, seed=3)
    lg = np.array([1.0, -2.0, 0.0, 0.5, 3.0], np.float32)
    d1, loss1, s1, g1 = fused(R, sdf, pos, quat, isc, cam, tgt, weight=0.5, loss_grad=lg)
    d2, loss2, s2, g2 = fused(R, sdf, pos, quat, isc, cam, tgt, weight=0.5, loss_grad=lg)
    assert np.array_equal(loss1, loss2) and np.array_equal(s1, s2)      # fixed-order sums
    for k in (1, 2, 3):
        assert np.array_equal(g1[k], g2[k])
    # linear in the per-view weight: view b of the pose gradients scales with 0.5 * lg[b]
    _, _, _, g_unit = fused(R, sdf, pos, quat, isc, cam, tgt, weight=1.0)
    for b in range(5):
        np.testing.assert_allclose(g1[1][b], 0.5 * lg[b] * g_unit[1][b], rtol=2e-6, atol=1e-12)
        np.testing.assert_allclose(g1[3][b], 0.5 * lg[b] * g_unit[3][b], rtol=2e-6, atol=1e-12)
    assert np.all(g1[1][2] == 0) and np.all(g1[2][2] == 0)


def test_empty_overlap_and_per_view_sdf(R):
    sdfs = np.stack([oracle.blobs_sdf(0), oracle.sphere_sdf(0.5), oracle.blobs_sdf(3), oracle.blobs_sdf(0)])
    pos, quat, isc = oracle.random_poses(4, seed=5, width=160, height=120, f=80.0)
    cam = (160, 120, 80.0, 60.0, 80.0, 80.0)
    tgt = oracle.render_forward(sdfs[0], pos, quat, isc, *cam, 0.005, dtype=np.float32) * 1.01
    tgt[1] = 0.0                                          # view 1: nothing observed
    pos[3] = [0.0, 0.0, 5.0]                              # view 3: object behind the camera
    d, loss, stats, g = fused(R, sdfs, pos, quat, isc, cam, tgt, per_view=True)
    assert np.isnan(loss[1]) and np.isnan(loss[3]) and np.isfinite(loss[0]) and np.isfinite(loss[2])
    assert stats[1, 1] == 0 and stats[3, 1] == 0 and np.all(d[3] == 0)
    for k in (1, 2, 3):
        assert np.all(g[k][1] == 0) and np.all(g[k][3] == 0) and np.all(np.isfinite(g[k]))
    assert np.all(g[0][1] == 0) and np.all(g[0][3] == 0) and np.any(g[0][0] != 0)
    l_ref, _ = oracle.depth_l1(d, tgt)
    np.testing.assert_allclose(loss[[0, 2]], l_ref[[0, 2]], rtol=2e-6)


def test_autograd_interface_matches_torch_loss(R):
    """render_depth_l1_batch(...) == render_depth_batch + the reference's torch expression."""
    from sdfest_amd import Camera, render_depth_batch, render_depth_l1_batch
    sdf, pos, quat, isc, cam, tgt = scene(4, 320, 240, 160.0, seed=11)
    W, H, cx, cy, fx, fy = cam
    camera = Camera(W, H, fx, fy, cx, cy, pixel_center=0.5)
    t = dev(tgt)
    wv = torch.tensor([1.0, 0.5, 2.0, 0.25], device="cuda")

    def leaves():
        return [dev(x).requires_grad_(True) for x in (sdf, pos, quat, isc)]

    a = leaves()
    loss, depth = render_depth_l1_batch(*a, t, 0.005, camera)
    assert not depth.requires_grad
    (loss * wv).sum().backward()
    b = leaves()
    est = render_depth_batch(*b, 0.005, camera)
    assert torch.equal(est.detach(), depth)
    ref = torch.stack([torch.mean(torch.abs(est[v] - t[v])[(t[v] > 0) & (est[v] > 0)]) for v in range(4)])
    (ref * wv).sum().backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-6)
    for x, y in zip(a[1:], b[1:]):
        np.testing.assert_allclose(x.grad.cpu().numpy(), y.grad.cpu().numpy(), rtol=1e-5, atol=1e-9)
    assert rel_err(a[0].grad.cpu().numpy(), b[0].grad.cpu().numpy()) <= 1e-5


def test_l1_argument_errors(R):
    from sdfest_amd import _lib
    L = _lib.lib()
    z = torch.zeros(16, device="cuda")
    rc = L.sdfr_render_forward_l1(z.data_ptr(), 64, 0, z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 8, 8,
                                  4.0, 4.0, 8.0, 8.0, 0.0, None, z.data_ptr(), z.data_ptr(), z.data_ptr(),
                                  z.data_ptr(), 1 << 30, 0, None)
    assert rc == -2 and b"NULL" in L.sdfr_last_error()
    rc = L.sdfr_render_forward_l1(z.data_ptr(), 64, 0, z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 8, 8,
                                  4.0, 4.0, 8.0, 8.0, 0.0, z.data_ptr(), z.data_ptr(), z.data_ptr(),
                                  z.data_ptr(), z.data_ptr(), 16, 0, None)
    assert rc == -3
    rc = L.sdfr_render_backward_l1(None, 1.0, None, z.data_ptr(), z.data_ptr(), z.data_ptr(), 64, 0,
                                   z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 8, 8, 4.0, 4.0, 8.0, 8.0,
                                   0, z.data_ptr(), 0, z.data_ptr(), z.data_ptr(), z.data_ptr(),
                                   z.data_ptr(), 1 << 30, 0, None)
    assert rc == -2
    assert L.sdfr_render_forward_l1_workspace_bytes(64, 4, 640, 480) > L.sdfr_render_forward_workspace_bytes(64, 4, 640, 480)
