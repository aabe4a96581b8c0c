"""GPU: the HIP renderer (through the C ABI) against the CPU oracle and the golden vectors."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import check_sdf_grad, dense_from_sparse, load_render_case, rel_err, render_cases

pytestmark = pytest.mark.gpu

REL = 1e-4  # north-star tolerance: depth and gradients within 1e-4 relative (fp32)


@pytest.fixture(scope="module")
def R():
    import sdfest_amd.differentiable_renderer as r
    assert torch.cuda.is_available()
    return r


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def hip_forward(R, sdf, pos, quat, isc, W, H, cx, cy, fx, fy, thr):
    pos = np.asarray(pos, np.float32).reshape(-1, 3)
    quat = np.asarray(quat, np.float32).reshape(-1, 4)
    isc = np.asarray(isc, np.float32).reshape(-1)
    return R.forward_raw(dev(sdf), dev(pos), dev(quat), dev(isc), W, H, cx, cy, fx, fy, thr).cpu().numpy()


def hip_backward(R, g, depth, sdf, pos, quat, isc, W, H, cx, cy, fx, fy, mode=0):
    pos = np.asarray(pos, np.float32).reshape(-1, 3)
    quat = np.asarray(quat, np.float32).reshape(-1, 4)
    isc = np.asarray(isc, np.float32).reshape(-1)
    B = pos.shape[0]
    out = R.backward_raw(dev(np.asarray(g).reshape(B, H, W)), dev(np.asarray(depth).reshape(B, H, W)),
                         dev(sdf), dev(pos), dev(quat), dev(isc), W, H, cx, cy, fx, fy, mode)
    return [o.cpu().numpy() for o in out]


def check_depth(d_hip, d_ref, margin, name=""):
    robust = margin > 1e-5
    mism = (d_hip > 0) != (d_ref > 0)
    assert not np.any(mism & robust), f"{name}: hit mask differs on {int((mism & robust).sum())} robust pixels"
    assert mism.sum() <= max(2, 1e-4 * d_ref.size), f"{name}: {int(mism.sum())} fragile flips"
    both = (d_hip > 0) & (d_ref > 0)
    if both.any():
        # the oracle's margin is the smallest |lhs - rhs| of ANY branch of a pixel's march, the per-step hit tests
        # included: a pixel below it may also take its hit one sample earlier or later (a depth off by ~threshold)
        rel = np.abs(d_hip / np.where(both, d_ref, 1.0) - 1) * both
        assert rel[robust].max(initial=0.0) < REL, f"{name}: depth rel err {rel[robust].max()}"
        fragile = both & ~robust & (rel >= REL)
        assert fragile.sum() <= max(2, 1e-4 * d_ref.size) and rel[~robust].max(initial=0.0) < 0.05, \
            f"{name}: {int(fragile.sum())} fragile pixels, rel err {rel[~robust].max(initial=0.0)}"
    return mism


def pose_l1(dimg, g):
    return np.array([np.sum(np.abs(dimg[..., k] * g)) for k in range(8)])


@pytest.mark.parametrize("name", render_cases())
def test_forward_matches_oracle_and_golden(R, name):
    c = load_render_case(name)
    args = (c["sdf"], c["p"], c["q"], c["inv_scale"], c["W"], c["H"], c["cx"], c["cy"], c["fx"],
            c["fy"], c["thr"])
    d_hip = hip_forward(R, *args)[0]
    d_or, _, margin = oracle.render_forward(*args, dtype=np.float32, with_aux=True)
    check_depth(d_hip, d_or[0], margin[0], name)
    # and directly against the reference's numpy twin (float64 golden)
    _, _, m64 = oracle.render_forward(*args, dtype=np.float64, with_aux=True)
    check_depth(d_hip, c["depth"], m64[0], name + "/golden")


@pytest.mark.parametrize("name", render_cases())
@pytest.mark.parametrize("gi", [0, 1])
def test_backward_matches_oracle_and_golden(R, name, gi):
    c = load_render_case(name)
    g = c[f"g{gi}_image"].astype(np.float32)
    depth = c["depth"].astype(np.float32)   # the twin's depth: same hit points for everybody
    cam = (c["W"], c["H"], c["cx"], c["cy"], c["fx"], c["fy"])
    g_sdf, g_pos, g_quat, g_isc = hip_backward(R, g, depth, c["sdf"], c["p"], c["q"],
                                               c["inv_scale"], *cam)
    o_sdf, o_pos, o_quat, o_isc = oracle.render_backward(g, depth, c["sdf"], c["p"], c["q"],
                                                         c["inv_scale"], *cam[2:], dtype=np.float32)
    pose = np.concatenate([g_pos[0], g_quat[0], g_isc])
    l1 = pose_l1(c["dimg"], g)
    ref_o = np.concatenate([o_pos[0], o_quat[0], o_isc])
    assert np.all(np.abs(pose - ref_o) <= REL * np.maximum(l1, 1e-30) + 1e-30), (pose, ref_o)
    assert np.all(np.abs(pose - c[f"g{gi}_pose"]) <= REL * np.maximum(l1, 1e-30) + 1e-30)
    if len(c["gsdf_idx"]):
        assert rel_err(g_sdf, o_sdf) <= REL
        assert rel_err(g_sdf, dense_from_sparse(c["gsdf_idx"], c[f"g{gi}_gsdf"])) <= REL
    else:
        assert not g_sdf.any() and not pose.any()


def test_backward_cuda_compat_mode(R):
    c = load_render_case("g3_pose1_64x48")
    g = c["g1_image"].astype(np.float32)
    depth = c["depth"].astype(np.float32)
    cam = (c["W"], c["H"], c["cx"], c["cy"], c["fx"], c["fy"])
    g_sdf = hip_backward(R, g, depth, c["sdf"], c["p"], c["q"], c["inv_scale"], *cam, mode=1)[0]
    o_sdf = oracle.render_backward(g, depth, c["sdf"], c["p"], c["q"], c["inv_scale"], *cam[2:],
                                   dtype=np.float32, sdf_grad_mode=1)[0]
    assert rel_err(g_sdf, o_sdf) <= REL
    exact = oracle.render_backward(g, depth, c["sdf"], c["p"], c["q"], c["inv_scale"], *cam[2:],
                                   dtype=np.float32, sdf_grad_mode=0)[0]
    assert rel_err(g_sdf, exact) > 1e-2   # it really is a different assignment


def c2_scene(W=640, H=480):
    f = W / 2.0
    return dict(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, thr=0.005)


def test_c1_c2_single_view_full_size(R):
    """BASELINE configs[0] (160x120) and configs[1] (640x480): blobs(0), identity pose."""
    sdf = oracle.blobs_sdf(0)
    for W, H in ((160, 120), (640, 480)):
        s = c2_scene(W, H)
        args = (sdf, [0, 0, -1.5], [0, 0, 0, 1], [2.0], W, H, s["cx"], s["cy"], s["fx"], s["fy"], s["thr"])
        d_hip = hip_forward(R, *args)[0]
        d_or, steps, margin = oracle.render_forward(*args, dtype=np.float32, with_aux=True)
        check_depth(d_hip, d_or[0], margin[0], f"{W}x{H}")
        if (W, H) == (160, 120):
            assert int((d_or[0] > 0).sum()) == 948
        else:
            assert int((d_or[0] > 0).sum()) == 15138
        rng = np.random.default_rng(0)
        g = rng.uniform(-1, 1, (H, W)).astype(np.float32)
        cam = (W, H, s["cx"], s["cy"], s["fx"], s["fy"])
        hb = hip_backward(R, g, d_or[0], sdf, [0, 0, -1.5], [0, 0, 0, 1], [2.0], *cam)
        ob = oracle.render_backward(g, d_or[0], sdf, [0, 0, -1.5], [0, 0, 0, 1], [2.0], *cam[2:],
                                    dtype=np.float32)
        dimg = oracle.render_derivative_images(d_or[0], sdf, [0, 0, -1.5], [0, 0, 0, 1], [2.0],
                                               *cam[2:], dtype=np.float64)[0]
        l1 = pose_l1(dimg, g)
        pose = np.concatenate([hb[1][0], hb[2][0], hb[3]])
        ref = np.concatenate([ob[1][0], ob[2][0], ob[3]])
        assert np.all(np.abs(pose - ref) <= REL * l1)
        assert rel_err(hb[0], ob[0]) <= REL


@pytest.mark.parametrize("mode", [0, 1])
def test_batch_random_poses_equals_per_view(R, mode):
    """configs[2] shape (random poses of one SDF) at B=8: a batched launch equals B single
    launches bit for bit (forward, pose grads) and the oracle within tolerance -- with the exact d/dSDF weights
    (mode 0, simple_renderer.py:399-408) and with the ones the reference's GPU extension adds (mode 1,
    sdf_renderer_cuda.cu:373-388: the oracle's mode 1 is pinned by reading those lines)."""
    sdf = oracle.blobs_sdf(0)
    B, s = 8, c2_scene()
    pos, quat, isc = oracle.random_poses(B, seed=1)
    cam = (s["W"], s["H"], s["cx"], s["cy"], s["fx"], s["fy"])
    d_b = hip_forward(R, sdf, pos, quat, isc, *cam, s["thr"])
    d_or, _, margin = oracle.render_forward(sdf, pos, quat, isc, *cam, s["thr"], dtype=np.float32,
                                            with_aux=True)
    rng = np.random.default_rng(2)
    g = rng.uniform(-1, 1, d_b.shape).astype(np.float32)
    hb = hip_backward(R, g, d_b, sdf, pos, quat, isc, *cam, mode=mode)
    acc = np.zeros_like(hb[0], dtype=np.float64)
    for b in range(B):
        d1 = hip_forward(R, sdf, pos[b], quat[b], isc[b:b + 1], *cam, s["thr"])[0]
        assert np.array_equal(d1, d_b[b])
        check_depth(d_b[b], d_or[b], margin[b], f"view{b}")
        assert (d_b[b] > 0).sum() > 1000
        h1 = hip_backward(R, g[b], d_b[b], sdf, pos[b], quat[b], isc[b:b + 1], *cam, mode=mode)
        assert np.array_equal(h1[1][0], hb[1][b]) and np.array_equal(h1[2][0], hb[2][b])
        assert h1[3][0] == hb[3][b]
        acc += h1[0]
    assert rel_err(hb[0], acc) <= 1e-5          # float-atomic order only
    ob = oracle.render_backward(g, d_b, sdf, pos, quat, isc, *cam[2:], dtype=np.float32, sdf_grad_mode=mode)
    check_sdf_grad(hb[0], ob[0], mode, int((d_b > 0).sum()), REL)
    if mode == 1:     # (and the two weightings are not the same thing)
        o0 = oracle.render_backward(g, d_b, sdf, pos, quat, isc, *cam[2:], dtype=np.float32, sdf_grad_mode=0)
        assert rel_err(hb[0], o0[0]) > 0.05
    for b in range(B):
        dimg = oracle.render_derivative_images(d_b[b], sdf, pos[b], quat[b], isc[b:b + 1], *cam[2:],
                                               dtype=np.float64)[0]
        l1 = pose_l1(dimg, g[b])
        pose = np.concatenate([hb[1][b], hb[2][b], hb[3][b:b + 1]])
        ref = np.concatenate([ob[1][b], ob[2][b], ob[3][b:b + 1]])
        assert np.all(np.abs(pose - ref) <= REL * l1), (b, pose, ref, l1)


def test_batched_forward_with_a_4_byte_aligned_sdf(R):
    """The batch path packs face records and takes plane minima (four z-planes per block with 16-byte loads when
    the grid pointer allows it): a grid that starts 4 bytes into an allocation gives the same image."""
    sdf = oracle.blobs_sdf(0)
    B, W, H, f = 6, 160, 120, 80.0
    pos, quat, isc = oracle.random_poses(B, seed=3, width=W, height=H, f=f)
    cam = (W, H, W / 2, H / 2, f, f)
    aligned = R.forward_raw(dev(sdf), dev(pos), dev(quat), dev(isc), *cam, 0.005)
    big = torch.zeros(64 ** 3 + 8, device="cuda")
    for off in (1, 2, 3):
        view = big[off:off + 64 ** 3].view(64, 64, 64)
        view.copy_(dev(sdf))
        assert view.data_ptr() % 16 == 4 * off and view.is_contiguous()
        assert torch.equal(R.forward_raw(view, dev(pos), dev(quat), dev(isc), *cam, 0.005), aligned)
    assert (aligned > 0).sum().item() > 1000


def test_per_view_sdf_batches(R):
    sdfs = np.stack([oracle.blobs_sdf(0), oracle.sphere_sdf(0.5), oracle.blobs_sdf(3)])
    pos, quat, isc = oracle.random_poses(3, seed=5, width=160, height=120, f=80.0)
    cam = (160, 120, 80.0, 60.0, 80.0, 80.0)
    d = hip_forward(R, sdfs, pos, quat, isc, *cam, 0.005)
    g = np.random.default_rng(1).uniform(-1, 1, d.shape).astype(np.float32)
    hb = hip_backward(R, g, d, sdfs, pos, quat, isc, *cam)
    assert hb[0].shape == sdfs.shape
    for b in range(3):
        do, _, m = oracle.render_forward(sdfs[b], pos[b], quat[b], isc[b:b + 1], *cam, 0.005,
                                         dtype=np.float32, with_aux=True)
        check_depth(d[b], do[0], m[0], f"sdf{b}")
        ob = oracle.render_backward(g[b], d[b], sdfs[b], pos[b], quat[b], isc[b:b + 1], *cam[2:],
                                    dtype=np.float32)
        assert rel_err(hb[0][b], ob[0]) <= REL


def test_generic_resolution_and_ragged_image(R):
    """R != 64 takes the runtime-resolution kernel; odd image sizes exercise partial tiles."""
    for Rn, W, H in ((32, 37, 29), (48, 65, 9), (17, 1, 1)):
        sdf = oracle.sphere_sdf(0.6, R=Rn)
        f = 30.0
        args = (sdf, [0.1, -0.05, -2.0], np.array([0.1, 0.2, -0.3, 0.9]) / np.linalg.norm([0.1, 0.2, -0.3, 0.9]),
                [1.2], W, H, W / 2, H / 2, f, f, 0.01)
        d = hip_forward(R, *args)[0]
        do, _, m = oracle.render_forward(*args, dtype=np.float32, with_aux=True)
        check_depth(d, do[0], m[0], f"R{Rn}")
        g = np.random.default_rng(Rn).uniform(-1, 1, (H, W)).astype(np.float32)
        hb = hip_backward(R, g, do[0], *args[:4], W, H, W / 2, H / 2, f, f)
        ob = oracle.render_backward(g, do[0], *args[:4], W / 2, H / 2, f, f, dtype=np.float32)
        if ob[0].any():
            assert rel_err(hb[0], ob[0]) <= REL


def test_size_independent_properties_full_size(R):
    """640x480, B=4: linearity of the backward in the upstream gradient, zero gradient for a
    zero upstream image, idempotence of the forward, untouched voxels exactly zero."""
    sdf = oracle.blobs_sdf(0)
    s = c2_scene()
    pos, quat, isc = oracle.random_poses(4, seed=9)
    cam = (s["W"], s["H"], s["cx"], s["cy"], s["fx"], s["fy"])
    d = hip_forward(R, sdf, pos, quat, isc, *cam, s["thr"])
    assert np.array_equal(d, hip_forward(R, sdf, pos, quat, isc, *cam, s["thr"]))
    rng = np.random.default_rng(4)
    g1 = rng.uniform(-1, 1, d.shape).astype(np.float32)
    g2 = rng.uniform(-1, 1, d.shape).astype(np.float32)
    b1 = hip_backward(R, g1, d, sdf, pos, quat, isc, *cam)
    b2 = hip_backward(R, g2, d, sdf, pos, quat, isc, *cam)
    b12 = hip_backward(R, g1 + 2 * g2, d, sdf, pos, quat, isc, *cam)
    for a, b, c in zip(b1, b2, b12):
        scale = np.abs(a).max() + 2 * np.abs(b).max()
        assert np.max(np.abs(a + 2 * b - c)) <= 2e-5 * scale
    z = hip_backward(R, np.zeros_like(g1), d, sdf, pos, quat, isc, *cam)
    assert all(not t.any() for t in z)
    assert (b1[0] == 0).mean() > 0.8


def test_drop_in_autograd_interface(R):
    """render_depth_gpu with the reference's signature: shapes (3,), (4,) / (1,4), () / (1,)."""
    from sdfest_amd import Camera, render_depth_gpu
    sdf = dev(oracle.blobs_sdf(0)).requires_grad_()
    cam = Camera(160, 120, 80.0, 80.0, 80.0, 60.0, pixel_center=0.5)
    for qshape, sshape in (((4,), ()), ((1, 4), (1,))):
        p = torch.tensor([0.0, 0.0, -1.5], device="cuda", requires_grad=True)
        q = torch.tensor([0.0, 0.0, 0.0, 1.0], device="cuda").reshape(qshape).requires_grad_()
        i_s = torch.full(sshape, 2.0, device="cuda", requires_grad=True)
        img = render_depth_gpu(sdf, p, q, i_s, None, None, None, 0.005, cam)
        assert img.shape == (120, 160) and int((img > 0).sum()) == 948
        img2 = render_depth_gpu(sdf, p, q, i_s, 160, 120, 90.0, 0.005)
        assert torch.allclose(img, img2, atol=1e-6)
        sdf.grad = None
        img.sum().backward()
        assert p.grad.shape == (3,) and q.grad.shape == qshape and i_s.grad.shape == sshape
        assert sdf.grad.shape == (64, 64, 64) and sdf.grad.abs().sum() > 0
    with pytest.raises(ValueError):
        render_depth_gpu(sdf, p, q, i_s, 160, 120, 90.0, 0.005, cam)
    with pytest.raises(RuntimeError):
        render_depth_gpu(sdf.detach().cpu(), p, q, i_s, None, None, None, 0.005, cam)
    with pytest.raises(RuntimeError):
        render_depth_gpu(sdf.detach().transpose(0, 1), p, q, i_s, None, None, None, 0.005, cam)


def test_noise_field_long_marches(R):
    """SURVEY 8d stress input: a non-Lipschitz noise field with an embedded sphere -- long, irregular
    marches (tens of steps), B=17 so the packed-record path runs (from SDFR_PACKED_MIN_VIEWS = 17 views on); vs the
    oracle step for step."""
    rng = np.random.default_rng(11)
    sdf = rng.uniform(0.02, 0.3, (64, 64, 64)).astype(np.float32)
    sphere = oracle.sphere_sdf(0.4)
    sdf = np.where(sphere < 0.08, sphere, sdf).astype(np.float32)
    pos, quat, isc = oracle.random_poses(17, seed=21, width=320, height=240, f=160.0)
    cam = (320, 240, 160.0, 120.0, 160.0, 160.0)
    d = hip_forward(R, sdf, pos, quat, isc, *cam, 0.01)
    do, steps, m = oracle.render_forward(sdf, pos, quat, isc, *cam, 0.01, dtype=np.float32, with_aux=True)
    assert steps.max() >= 20
    flips = 0
    for b in range(17):
        # (per view a tenth, over all views a twentieth of the robust pixels may be outliers: small views of this field
        # scatter -- 6.6 % of one view's 1 551 pixels at these poses, where the oracle's own fp32 and fp64 builds differ)
        flips += check_depth_count(d[b], do[b], m[b], outlier_frac=0.10)
    assert flips <= 2e-3 * d.size
    ok = (d > 0) & (do > 0) & (m > 1e-5)
    assert np.mean(np.abs(d[ok] / do[ok] - 1) > 1e-4) < 0.05
    g = rng.uniform(-1, 1, d.shape).astype(np.float32)
    hb = hip_backward(R, g, do, sdf, pos, quat, isc, *cam)
    ob = oracle.render_backward(g, do, sdf, pos, quat, isc, *cam[2:], dtype=np.float32)
    assert rel_err(hb[0], ob[0]) <= REL


def check_depth_count(d_hip, d_ref, margin, outlier_frac=0.05):
    """like check_depth but returns the number of fragile flips (chaotic fields have more of them)"""
    robust = margin > 1e-5
    mism = (d_hip > 0) != (d_ref > 0)
    both = (d_hip > 0) & (d_ref > 0)
    if both.any():
        # on a non-Lipschitz field a last-bit difference early in the march can change the whole
        # trajectory; compare depth only where the oracle's decisions were robust
        # (slopes > 1 amplify rounding exponentially along the march, so a handful of outliers
        # is a property of the field; the bulk must agree to the usual tolerance)
        ok = both & robust
        err = np.abs(d_hip[ok] / d_ref[ok] - 1)
        # (the oracle's own fp32 and fp64 builds disagree by > 1e-4 on 0.9 % of these pixels)
        assert np.mean(err > 1e-4) < outlier_frac and np.median(err) < 5e-6
    return int((mism & robust).sum()) + int(mism.sum())


def test_packed_record_path_generic_and_odd_resolution(R):
    """B >= 17 views sharing a grid (SDFR_PACKED_MIN_VIEWS) take the face-record march; exercise it at R != 64 incl. an
    odd R (padded 2x2 blocks) and check it equals the per-view (plain-grid) launches bit for bit."""
    for Rn in (33, 48, 7):
        sdf = oracle.sphere_sdf(0.55, R=Rn)
        pos, quat, isc = oracle.random_poses(17, seed=Rn, width=96, height=72, f=60.0)
        cam = (96, 72, 48.0, 36.0, 60.0, 60.0)
        d = hip_forward(R, sdf, pos, quat, isc, *cam, 0.01)
        for b in range(17):
            d1 = hip_forward(R, sdf, pos[b], quat[b], isc[b:b + 1], *cam, 0.01)[0]
            assert np.array_equal(d1, d[b]), (Rn, b)
        do, _, m = oracle.render_forward(sdf, pos, quat, isc, *cam, 0.01, dtype=np.float32, with_aux=True)
        for b in range(17):
            check_depth(d[b], do[b], m[b], f"R{Rn}/view{b}")
        assert (d > 0).sum() > 200


def test_c3_full_bench_configuration_parity(R):
    """BASELINE configs[2] exactly as bench.py runs it (256 seeded random poses of blobs(0), 640x480):
    every view's depth against the oracle, and the batch-summed d/dSDF + all pose gradients."""
    sdf = oracle.blobs_sdf(0)
    B, s = 256, c2_scene()
    pos, quat, isc = oracle.random_poses(B, seed=1)
    cam = (s["W"], s["H"], s["cx"], s["cy"], s["fx"], s["fy"])
    oracle.set_threads(min(64, os.cpu_count() or 1))
    d = hip_forward(R, sdf, pos, quat, isc, *cam, s["thr"])
    assert int((d > 0).sum()) in range(4207800, 4207900)       # bench.py reports 4207852/3
    do, _, m = oracle.render_forward(sdf, pos, quat, isc, *cam, s["thr"], dtype=np.float32, with_aux=True)
    robust = m > 1e-5
    mism = (d > 0) != (do > 0)
    assert not np.any(mism & robust)
    assert mism.sum() <= 1e-5 * d.size
    both = (d > 0) & (do > 0)
    err = np.abs(d[both] / do[both] - 1)
    # a ray grazing a surface within rounding distance of the threshold can take the near or the
    # far hit: such pixels have a tiny decision margin in the oracle; everywhere else 1e-4 holds
    assert np.max(err[robust[both]]) < REL
    assert np.sum(err >= REL) <= 1e-5 * d.size
    g = np.random.default_rng(5).uniform(-1, 1, d.shape).astype(np.float32)
    hb = hip_backward(R, g, do, sdf, pos, quat, isc, *cam)
    ob = oracle.render_backward(g, do, sdf, pos, quat, isc, *cam[2:], dtype=np.float32)
    assert rel_err(hb[0], ob[0]) <= REL
    # pose gradients: sums of ~16k signed terms per view whose per-pixel derivative is discontinuous
    # across cell faces, so the yardstick is the sum of magnitudes (as in the small-scene tests)
    pose = np.concatenate([hb[1], hb[2], hb[3][:, None]], axis=1)
    ref = np.concatenate([ob[1], ob[2], ob[3][:, None]], axis=1)
    for b0 in range(0, B, 32):
        sl = slice(b0, b0 + 32)
        dimg = oracle.render_derivative_images(do[sl], sdf, pos[sl], quat[sl], isc[sl], *cam[2:],
                                               dtype=np.float32)
        l1 = np.abs(dimg * g[sl][..., None]).sum(axis=(1, 2), dtype=np.float64)
        assert np.all(np.abs(pose[sl] - ref[sl]) <= REL * l1), b0


@pytest.mark.parametrize("W,H,B,f", [(640, 480, 48, 320.0), (200, 136, 256, 300.0)])
def test_batch_backward_mixes_both_tile_shapes(R, W, H, B, f):
    """A batch backward tiles every view by how many pixels one voxel spans on the screen -- 32 x 32 pixel tiles
    from 2 pixels per voxel, 64 x 8 below (common.hpp, kBwdBigTile) -- in ONE launch: large and small objects
    interleaved in a batch (and an image whose size is no multiple of either tile) against the oracle, and against
    the same views one at a time (32 x 8 tiles, another summation order)."""
    sdf = oracle.blobs_sdf(0)
    cam = (W, H, W / 2.0, H / 2.0, f, f)
    assert B * ((W + 63) // 64) * ((H + 7) // 8) >= 16384          # a batch launch (common.hpp, backward_geom)
    pos, quat, isc = oracle.random_poses(B, seed=7, width=W, height=H, f=f)
    isc = isc.copy()
    isc[1::2] *= 2.6                               # every other object 2.6 x smaller
    ratio = f * (2.0 / isc / 63.0) / np.linalg.norm(pos, axis=1)
    assert (ratio >= 2.0).sum() >= 4 and (ratio < 2.0).sum() >= 16    # both tilings in one launch
    d = hip_forward(R, sdf, pos, quat, isc, *cam, 0.005)
    assert (d > 0).reshape(B, -1).sum(1).min() > 30
    g = np.random.default_rng(8).uniform(-1, 1, d.shape).astype(np.float32)
    hb = hip_backward(R, g, d, sdf, pos, quat, isc, *cam)
    ob = oracle.render_backward(g, d, sdf, pos, quat, isc, *cam[2:], dtype=np.float32)
    assert rel_err(hb[0], ob[0]) <= REL
    pose = np.concatenate([hb[1], hb[2], hb[3][:, None]], axis=1)
    ref = np.concatenate([ob[1], ob[2], ob[3][:, None]], axis=1)
    dimg = oracle.render_derivative_images(d, sdf, pos, quat, isc, *cam[2:], dtype=np.float32)
    l1 = np.abs(dimg * g[..., None]).sum(axis=(1, 2), dtype=np.float64)
    assert np.all(np.abs(pose - ref) <= REL * l1)
    for b in range(min(B, 64)):
        h1 = hip_backward(R, g[b], d[b], sdf, pos[b], quat[b], isc[b:b + 1], *cam)
        one = np.concatenate([h1[1][0], h1[2][0], h1[3]])
        assert np.all(np.abs(pose[b] - one) <= 1e-5 * l1[b]), b


def test_off_centre_non_square_intrinsics(R):
    """fx != fy and a principal point far from the image centre (single view and a packed batch):
    the set-up's screen rectangle and the ray generation must agree with the oracle."""
    sdf = oracle.blobs_sdf(0)
    W, H, fx, fy, cx, cy = 200, 136, 150.0, 95.0, 61.5, 103.25
    for B, seed in ((1, 21), (5, 22), (17, 23)):     # (one view; a plain-grid batch; a packed batch)
        pos, quat, isc = oracle.random_poses(B, seed=seed, width=W, height=H, f=120.0)
        pos[:, 0] -= 0.25 * np.abs(pos[:, 2])      # towards the shifted principal point
        pos[:, 1] -= 0.30 * np.abs(pos[:, 2])
        cam = (W, H, cx, cy, fx, fy)
        d = hip_forward(R, sdf, pos, quat, isc, *cam, 0.005)
        do, _, m = oracle.render_forward(sdf, pos, quat, isc, *cam, 0.005, dtype=np.float32, with_aux=True)
        assert (do > 0).sum() > 200 * B
        for b in range(B):
            check_depth(d[b], do[b], m[b], f"B{B} view{b}")
        g = np.random.default_rng(seed).uniform(-1, 1, d.shape).astype(np.float32)
        hb = hip_backward(R, g, d, sdf, pos, quat, isc, *cam)
        ob = oracle.render_backward(g, d, sdf, pos, quat, isc, *cam[2:], dtype=np.float32)
        assert rel_err(hb[0], ob[0]) <= REL
        for b in range(B):
            dimg = oracle.render_derivative_images(d[b], sdf, pos[b], quat[b], isc[b:b + 1], *cam[2:],
                                                   dtype=np.float64)[0]
            l1 = pose_l1(dimg, g[b])
            pose = np.concatenate([hb[1][b], hb[2][b], hb[3][b:b + 1]])
            ref = np.concatenate([ob[1][b], ob[2][b], ob[3][b:b + 1]])
            assert np.all(np.abs(pose - ref) <= REL * l1), (B, b, pose, ref, l1)


def test_large_grids_take_the_integer_index_path(R):
    """R = 129 (just above the packed-record limit) and R = 264 (> 256: the float-formed linear index
    would no longer be exact, the integer form is used) -- forward and backward against the oracle."""
    for Rn in (129, 264):
        sdf = oracle.sphere_sdf(0.55, R=Rn)
        W, H, f = 96, 72, 80.0
        B = 5 if Rn == 129 else 1
        pos, quat, isc = oracle.random_poses(B, seed=Rn, width=W, height=H, f=f)
        cam = (W, H, W / 2, H / 2, f, f)
        d = hip_forward(R, sdf, pos, quat, isc, *cam, 0.005)
        do, _, m = oracle.render_forward(sdf, pos, quat, isc, *cam, 0.005, dtype=np.float32, with_aux=True)
        assert (do > 0).sum() > 50 * B
        for b in range(B):
            check_depth(d[b], do[b], m[b], f"R{Rn} view{b}")
        g = np.random.default_rng(Rn).uniform(-1, 1, d.shape).astype(np.float32)
        hb = hip_backward(R, g, d, sdf, pos, quat, isc, *cam)
        ob = oracle.render_backward(g, d, sdf, pos, quat, isc, *cam[2:], dtype=np.float32)
        assert rel_err(hb[0], ob[0]) <= REL
        assert np.abs(hb[0]).max() > 0


def test_low_resolution_batch_takes_the_wide_table(R):
    """128 views of 320x240: batch tiles with the 4 x 1024 run table (render_backward_kernel<..., WIDE>)
    against the oracle, and the pose gradients against single-view calls (small tiles, other table)."""
    sdf = oracle.blobs_sdf(0)
    B, W, H, f = 128, 320, 240, 160.0
    pos, quat, isc = oracle.random_poses(B, seed=31, width=W, height=H, f=f)
    cam = (W, H, W / 2, H / 2, f, f)
    d = hip_forward(R, sdf, pos, quat, isc, *cam, 0.005)
    g = np.random.default_rng(31).uniform(-1, 1, d.shape).astype(np.float32)
    hb = hip_backward(R, g, d, sdf, pos, quat, isc, *cam)
    ob = oracle.render_backward(g, d, sdf, pos, quat, isc, *cam[2:], dtype=np.float32)
    assert (d > 0).sum() > 200 * B
    assert rel_err(hb[0], ob[0]) <= REL
    for b in (0, 17, 127):
        h1 = hip_backward(R, g[b], d[b], sdf, pos[b], quat[b], isc[b:b + 1], *cam)
        dimg = oracle.render_derivative_images(d[b], sdf, pos[b], quat[b], isc[b:b + 1], *cam[2:], dtype=np.float64)[0]
        l1 = pose_l1(dimg, g[b])
        pose_b = np.concatenate([hb[1][b], hb[2][b], hb[3][b:b + 1]])
        pose_1 = np.concatenate([h1[1][0], h1[2][0], h1[3]])
        ref = np.concatenate([ob[1][b], ob[2][b], ob[3][b:b + 1]])
        assert np.all(np.abs(pose_b - ref) <= REL * l1) and np.all(np.abs(pose_1 - ref) <= REL * l1)
