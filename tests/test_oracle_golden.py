"""CPU: the oracle (oracle/, test infrastructure) against golden vectors captured from
the reference's own importable code (tools/make_goldens.py).  This is what PINS the oracle."""
import os

import numpy as np
import pytest

import oracle
from helpers import (GOLDEN, POSE_NAMES, dense_from_sparse, load_render_case, rel_err,
                     render_cases)

CASES = render_cases()


def test_golden_cases_present():
    assert len(CASES) >= 12
    for f in ("pc_loss.npz", "decoder_mug.npz", "mug_decoder_weights.npz", "quaternion.npz"):
        assert os.path.exists(os.path.join(GOLDEN, f))


@pytest.mark.parametrize("name", CASES)
def test_render_forward_f64_matches_numpy_twin(name):
    c = load_render_case(name)
    depth, steps, margin = oracle.render_forward(
        c["sdf"], c["p"], c["q"], c["inv_scale"], c["W"], c["H"], c["cx"], c["cy"], c["fx"],
        c["fy"], c["thr"], dtype=np.float64, with_aux=True)
    depth, steps = depth[0], steps[0]
    # identical hit mask, identical march trajectory length, depth to ~1e-12
    assert np.array_equal(depth > 0, c["depth"] > 0)
    # the twin's "c" image records the step count of HIT rays only (simple_renderer.py:294-302)
    hit = c["count"] > 0
    assert np.array_equal(steps[hit], c["count"][hit])
    assert np.max(np.abs(depth - c["depth"])) <= 1e-10


@pytest.mark.parametrize("name", CASES)
def test_render_forward_f32_close_to_twin(name):
    c = load_render_case(name)
    d64, _, margin = oracle.render_forward(
        c["sdf"], c["p"], c["q"], c["inv_scale"], c["W"], c["H"], c["cx"], c["cy"], c["fx"],
        c["fy"], c["thr"], dtype=np.float64, with_aux=True)
    d32 = oracle.render_forward(
        c["sdf"], c["p"], c["q"], c["inv_scale"], c["W"], c["H"], c["cx"], c["cy"], c["fx"],
        c["fy"], c["thr"], dtype=np.float32)
    robust = margin[0] > 1e-5
    assert np.array_equal((d32[0] > 0)[robust], (c["depth"] > 0)[robust])
    both = (d32[0] > 0) & (c["depth"] > 0)
    if both.any():
        assert np.max(np.abs(d32[0][both] / c["depth"][both] - 1)) < 1e-5


@pytest.mark.parametrize("name", CASES)
def test_render_derivative_images_match_twin(name):
    c = load_render_case(name)
    # the twin also differentiates hits at t == 0 (depth 0); the CUDA kernel skips
    # depth == 0 pixels (sdf_renderer_cuda.cu:334) and so does the oracle.
    dimg = oracle.render_derivative_images(c["depth"], c["sdf"], c["p"], c["q"], c["inv_scale"],
                                           c["cx"], c["cy"], c["fx"], c["fy"], dtype=np.float64)[0]
    hit = c["depth"] > 0
    for k, nm in enumerate(POSE_NAMES):
        ref = c["dimg"][..., k]
        scale = max(np.max(np.abs(ref)), 1e-12)
        if hit.any():
            assert np.max(np.abs(dimg[..., k] - ref)[hit]) <= 1e-9 * scale + 1e-12, nm
        assert np.all(dimg[..., k][~hit] == 0)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("gi", [0, 1])
def test_render_backward_matches_twin(name, gi):
    c = load_render_case(name)
    g = c[f"g{gi}_image"]
    for dtype, tol in ((np.float64, 1e-9), (np.float32, 1e-4)):
        # feed the twin's own depth so both differentiate the same hit points
        g_sdf, g_pos, g_quat, g_isc = oracle.render_backward(
            g, c["depth"], c["sdf"], c["p"], c["q"], c["inv_scale"], c["cx"], c["cy"], c["fx"],
            c["fy"], dtype=dtype)
        pose = np.concatenate([g_pos[0], g_quat[0], g_isc])
        ref_pose = c[f"g{gi}_pose"]
        l1 = np.array([np.sum(np.abs(c["dimg"][..., k] * g)) for k in range(8)])
        assert np.all(np.abs(pose - ref_pose) <= tol * np.maximum(l1, 1e-30) + 1e-300)
        ref_sdf = dense_from_sparse(c["gsdf_idx"], c[f"g{gi}_gsdf"])
        if len(c["gsdf_idx"]):
            assert rel_err(g_sdf, ref_sdf) <= tol
        else:
            assert np.all(g_sdf == 0)


def test_cuda_compat_weights_are_the_documented_permutation():
    """mode 1 = sdf_renderer_cuda.cu:373-388.  Pinned by construction only:
    check it is the stated permutation of the exact weights on one pixel."""
    c = load_render_case("g1_sphere_32x24")
    g = np.zeros((c["H"], c["W"]))
    hit = np.argwhere(c["depth"] > 0)[7]
    g[tuple(hit)] = 1.0
    a = oracle.render_backward(g, c["depth"], c["sdf"], c["p"], c["q"], c["inv_scale"], c["cx"],
                               c["cy"], c["fx"], c["fy"], dtype=np.float64, sdf_grad_mode=0)[0]
    b = oracle.render_backward(g, c["depth"], c["sdf"], c["p"], c["q"], c["inv_scale"], c["cx"],
                               c["cy"], c["fx"], c["fy"], dtype=np.float64, sdf_grad_mode=1)[0]
    ia = np.argwhere(a != 0)
    base = ia.min(axis=0)
    wa = np.array([a[tuple(base + [i, j, k])] for i in (0, 1) for j in (0, 1) for k in (0, 1)])
    wb = np.array([b[tuple(base + [i, j, k])] for i in (0, 1) for j in (0, 1) for k in (0, 1)])
    # exact order: 000,001,010,011,100,101,110,111 ; compat: [001,010,011,100,101,101,110,111]
    assert np.allclose(wb, wa[[1, 2, 3, 4, 5, 5, 6, 7]], rtol=1e-12)
    assert abs(wa.sum() - 1.0 * (wa.sum())) == 0 and not np.allclose(wa, wb)


def test_pc_loss_matches_torch_reference():
    d = np.load(os.path.join(GOLDEN, "pc_loss.npz"))
    sdf = oracle.blobs_sdf(0)
    for i in range(int(d["n_cases"])):
        pts, go = d[f"c{i}_points"], d[f"c{i}_gout"]
        pos, quat, scale = d[f"c{i}_pos"], d[f"c{i}_quat"], d[f"c{i}_scale"]
        for tag, dtype, tol in (("f64", np.float64, 1e-10), ("f32", np.float32, 3e-5)):
            ref_val = d[f"c{i}_{tag}_value"]
            val = oracle.pc_loss_forward(pts, pos, quat, scale, sdf, dtype=dtype)
            if dtype == np.float64:
                assert np.array_equal(val != 0, ref_val != 0)
                assert np.max(np.abs(val - ref_val)) <= 1e-12
            else:
                # fp32 rounding may move a point across the volume face (mask flip)
                same = (val != 0) == (ref_val != 0)
                assert same.mean() > 0.995
                assert np.max(np.abs(val - ref_val)[same]) <= 1e-5
            g_sdf, g_pos, g_quat, g_scale = oracle.pc_loss_backward(go, pts, pos, quat, scale, sdf,
                                                                    dtype=dtype)
            ref_sdf = dense_from_sparse(d[f"c{i}_{tag}_gsdf_idx"], d[f"c{i}_{tag}_gsdf_val"])
            assert rel_err(g_sdf, ref_sdf) <= max(tol, 1e-9) or dtype == np.float32
            if dtype == np.float64:
                assert rel_err(g_pos, d[f"c{i}_{tag}_gpos"]) <= 1e-9
                assert rel_err(g_quat, d[f"c{i}_{tag}_gquat"]) <= 1e-9
                assert abs(g_scale - d[f"c{i}_{tag}_gscale"]) <= 1e-9 * abs(d[f"c{i}_{tag}_gscale"])


def _mug_config(d):
    return {"latent_size": int(d["latent_size"]), "tsdf": False, "decoder": {
        "fc_layers": [{"out": int(o)} for o in d["fc_out"]],
        "conv_layers": [{"in_size": int(a), "in_channels": int(b), "out_channels": int(c),
                         "kernel_size": int(k), "relu": bool(r)}
                        for a, b, c, k, r in zip(d["conv_in_size"], d["conv_cin"], d["conv_cout"],
                                                 d["conv_k"], d["conv_relu"])]}}


def test_decoder_matches_torch_reference():
    d = np.load(os.path.join(GOLDEN, "decoder_mug.npz"))
    w = np.load(os.path.join(GOLDEN, "mug_decoder_weights.npz"))
    cfg = _mug_config(d)
    params = oracle.pack_decoder_params(w, 3, 4)
    assert params.size == 430287
    out = oracle.decoder_forward(params, cfg, d["z"], dtype=np.float32)
    assert out.shape == (12, 1, 64, 64, 64)
    ref0 = d["z0_full"]
    assert np.max(np.abs(out[0, 0] - ref0)) <= 2e-5 * np.max(np.abs(ref0))
    for i in range(12):
        assert np.max(np.abs(out[i, 0, ::4, ::4, ::4] - d["sub16"][i])) <= 5e-5 * max(1.0, np.abs(d["sub16"][i]).max())
        s = d["stats"][i]
        assert abs(out[i].sum(dtype=np.float64) - s[0]) <= 1e-4 * s[1]
        assert abs(out[i].min() - s[2]) <= 1e-4 and abs(out[i].max() - s[3]) <= 1e-4 * max(1, abs(s[3]))
    out64 = oracle.decoder_forward(params.astype(np.float64), cfg, d["z"][:1], dtype=np.float64)
    assert np.max(np.abs(out64[0, 0] - ref0)) <= 2e-5 * np.max(np.abs(ref0))


def test_depth_to_pointcloud_convention():
    # pointset_utils.py:57-77: x=(col-cx0)*z/fx, y=-(row-cy0)*z/fy, z=-depth, row-major order
    depth = np.zeros((4, 6), dtype=np.float32)
    depth[1, 2] = 2.0
    depth[3, 5] = 0.5
    pts = oracle.depth_to_pointcloud(depth, fx=10.0, fy=20.0, cx0=2.5, cy0=1.5)
    assert pts.shape == (2, 3)
    assert np.allclose(pts[0], [(2 - 2.5) * 2 / 10, -(1 - 1.5) * 2 / 20, -2.0])
    assert np.allclose(pts[1], [(5 - 2.5) * 0.5 / 10, -(3 - 1.5) * 0.5 / 20, -0.5])
