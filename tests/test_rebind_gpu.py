"""GPU: the captured render-and-compare loop taking NEW observations (FusedRenderAndCompare.rebind), as the reference
is used -- one pipeline call per detected object with fresh depth images (simple_setup.py:213-225, :333-334,
:408-470).  A re-bound loop must give, bit for bit, what a fresh object gives for the same observation -- eager and
graph-replayed, tail and records form -- without allocating, synchronising or capturing again."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu

T = lambda a, dev="cuda": torch.tensor(np.asarray(a, dtype=np.float32), device=dev)


@pytest.fixture(scope="module")
def scenes():
    """two observations of the same shape (2 views, 160x120): G7 run C as captured from the reference's pieces, and a
    second one rendered at another pose, shape and camera pair (bitwise reproducible: the forward kernel)"""
    import _loop_scenes as S
    from sdfest_amd import Camera, render_depth_gpu
    dec, d = S.mug_decoder()
    g7 = np.load(os.path.join(GOLDEN, "loop_g7.npz"))
    W, H = int(g7["c_W"]), int(g7["c_H"])
    cam = Camera(W, H, float(g7["c_fx"]), float(g7["c_fy"]), float(g7["c_cx"]), float(g7["c_cy"]), pixel_center=0.5)
    init = g7["c_init"]
    obs_c = dict(depth=T(g7["c_depth_images"]), cam_pos=T(g7["c_cam_pos"]), cam_quat=T(g7["c_cam_quat"]),
                 init=(T(init[None, 0:3]), T(init[None, 3:7]), T(init[7:8]), T(init[None, 8:])))
    # the second observation: another latent, a closer object (more observed points), other extrinsics
    z_true = T(d["z"][10:11]) * 0.4
    p_true = np.array([-0.03, 0.02, -0.42]); s_true = 0.07
    q_true = np.array([-0.4, 0.3, 0.2, 0.85]); q_true /= np.linalg.norm(q_true)
    cam_pos = np.array([[0.0, 0.0, 0.0], [-0.2, 0.06, 0.03]])
    cq = np.array([[0, 0, 0, 1.0], [0.03, -0.22, 0.02, 1.0]]); cq /= np.linalg.norm(cq, axis=1, keepdims=True)
    with torch.no_grad():
        sdf = dec.decode(z_true)[0, 0]
        depth = []
        for v in range(2):
            qc = cq[v] * np.array([-1, -1, -1, 1.0])
            depth.append(render_depth_gpu(sdf, T(S._qrot(qc, p_true - cam_pos[v])), T(S._qmul(qc, q_true)),
                                          T(1.0 / s_true), None, None, None, float(g7["thr"]), cam))
    q0 = q_true + np.array([0.04, 0.03, -0.05, 0.0])
    obs_d = dict(depth=torch.stack(depth).contiguous(), cam_pos=T(cam_pos), cam_quat=T(cq),
                 init=(T((p_true + 0.006)[None]), T((q0 / np.linalg.norm(q0))[None]), T([0.075]),
                       torch.zeros(1, 8, device="cuda")))
    n_c, n_d = [(o["depth"] > 0).sum(dim=(1, 2)).tolist() for o in (obs_c, obs_d)]
    assert min(n_c) > 300 and min(n_d) > 300 and n_c != n_d          # different point counts: nothing exactly sized survives
    cfg = {"threshold": float(g7["thr"]), "max_iterations": 7, "depth_weight": 1.0, "pc_weight": 3.0,
           "result_selection_strategy": "best_inlier_ratio"}
    return dec, cam, cfg, {"c": obs_c, "d": obs_d}


def _run(loop, obs, use_graph, with_history):
    h = [] if with_history else None
    out = loop(*obs["init"], use_graph=use_graph, history=h)
    torch.cuda.synchronize()
    extra = (loop.inlier_history.clone(), loop.best_state.clone(), loop.best_params.clone(), int(loop.step.item()))
    return [t.clone() for t in out], h, extra


def _same(a, b, what):
    out_a, h_a, ex_a = a
    out_b, h_b, ex_b = b
    for x, y in zip(out_a, out_b):
        assert torch.equal(x, y), (what, x, y)
    for x, y in zip(ex_a[:3], ex_b[:3]):
        assert torch.equal(x, y), what
    assert ex_a[3] == ex_b[3]
    if h_a is not None:
        assert len(h_a) == len(h_b)
        for it, (ha, hb) in enumerate(zip(h_a, h_b)):
            for k in ("position", "orientation", "scale", "latent", "loss"):
                assert torch.equal(ha[k], hb[k]) or (torch.isnan(ha[k]).all() and torch.isnan(hb[k]).all()), (what, it, k)


MODES = {
    # name: (constructor kwargs, bitwise?)
    "tail_pose_only": (dict(shape_optimization=False), True),                       # no float atomics on the way to the pose
    "records_det": (dict(shape_optimization=True, form="records", sdf_grad_mode=0x100 | 0x400), True),
    "records_pose_only": (dict(shape_optimization=False, form="records"), True),
    "tail_shape": (dict(shape_optimization=True), False),                           # d/dSDF by float atomics: ~1e-7
}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("use_graph,with_history", [(False, True), (True, True), (True, False)])
def test_rebind_reproduces_fresh_objects(scenes, mode, use_graph, with_history):
    """observation C -> D -> C on ONE object against three fresh objects"""
    from sdfest_amd.pipeline import FusedRenderAndCompare
    dec, cam, cfg, obs = scenes
    kw, bitwise = MODES[mode]

    def fresh(o):
        return FusedRenderAndCompare(dec, cam, cfg, o["depth"], o["cam_pos"], o["cam_quat"], graph_iterations=3, **kw)

    ref = {k: _run(fresh(o), o, use_graph, with_history) for k, o in obs.items()}
    # (the two observations really differ)
    assert not torch.equal(ref["c"][0][0], ref["d"][0][0])
    loop = fresh(obs["c"])
    assert loop.records_form == mode.startswith("records")
    ptrs = (loop.points.data_ptr(), loop.offsets.data_ptr(), loop.target.data_ptr(), loop.params.data_ptr())
    graphs = None
    for k in ("c", "d", "c", "d"):
        o = obs[k]
        loop.rebind(o["depth"], o["cam_pos"], o["cam_quat"])
        got = _run(loop, o, use_graph, with_history)
        if bitwise:
            _same(got, ref[k], (mode, k))
        else:
            for x, y, tol in zip(got[0], ref[k][0], (2e-6, 2e-5, 2e-6, 5e-5)):
                assert (x - y).abs().max().item() <= tol, (mode, k, x, y)
        # no re-allocation, and -- once captured -- no re-capture
        assert ptrs == (loop.points.data_ptr(), loop.offsets.data_ptr(), loop.target.data_ptr(), loop.params.data_ptr())
        if use_graph:
            now = (id(loop.graph), id(loop.graph_many), id(loop.graph_tail))
            assert graphs in (None, now), "a rebind must not capture again"
            graphs = now
    counts = loop.view_point_counts().tolist()
    assert counts == (obs["d"]["depth"] > 0).sum(dim=(1, 2)).tolist()


def test_rebind_takes_host_images_and_default_cameras(scenes):
    """a new observation may arrive as a host tensor; cameras default to the identity, like simple_setup.py:327-331"""
    from sdfest_amd.pipeline import FusedRenderAndCompare
    dec, cam, cfg, obs = scenes
    o = obs["d"]
    a = FusedRenderAndCompare(dec, cam, cfg, o["depth"], shape_optimization=False)
    ref = _run(a, o, True, False)
    b = FusedRenderAndCompare(dec, cam, cfg, views=2, shape_optimization=False)
    with pytest.raises(RuntimeError, match="rebind"):
        b(*o["init"])
    b.rebind(obs["c"]["depth"], obs["c"]["cam_pos"], obs["c"]["cam_quat"])
    _run(b, obs["c"], True, False)
    b.rebind(o["depth"].cpu())                       # cameras back to their defaults
    _same(_run(b, o, True, False), ref, "host images")
    with pytest.raises(ValueError):
        b.rebind(o["depth"][:1])
    with pytest.raises(ValueError):
        FusedRenderAndCompare(dec, cam, cfg)


def test_rebind_with_point_constraints(scenes):
    """the constraint's points are re-bound in place; presence and weight are launch arguments, so graphs are kept per
    (present, weight) and a caller switching between them captures each ONCE"""
    from sdfest_amd.pipeline import FusedRenderAndCompare
    dec, cam, cfg, obs = scenes
    o = obs["c"]
    cons = {"none": None, "w2": (T([0.0, 0.1, 0.0]), T([0.02, 0.09, 0.03]), 2.0),
            "w2b": (T([0.1, 0.0, 0.0]), T([0.09, -0.02, 0.01]), 2.0), "w5": (T([0.0, 0.1, 0.0]), T([0.02, 0.09, 0.03]), 5.0)}
    kw = dict(shape_optimization=False)
    ref = {k: _run(FusedRenderAndCompare(dec, cam, cfg, o["depth"], o["cam_pos"], o["cam_quat"], point_constraint=c, **kw),
                   o, True, True) for k, c in cons.items()}
    assert not torch.equal(ref["none"][0][1], ref["w2"][0][1]) and not torch.equal(ref["w2"][0][1], ref["w5"][0][1])
    loop = FusedRenderAndCompare(dec, cam, cfg, o["depth"], o["cam_pos"], o["cam_quat"], **kw)
    seen = {}
    for k in ("none", "w2", "w2b", "none", "w5", "w2", "w2b"):
        loop.rebind(o["depth"], o["cam_pos"], o["cam_quat"], point_constraint=cons[k])
        _same(_run(loop, o, True, True), ref[k], k)
        key = None if cons[k] is None else cons[k][2]
        assert seen.setdefault(key, id(loop.graph)) == id(loop.graph)
    assert len(loop._graphs) == 3


def test_preprocess_depth_is_the_torch_expression_bit_for_bit():
    """simple_setup.py:689-693: depth[~mask] = 0; depth[depth > far] = 0 -- including NaN / inf pixels, an image size
    that is not a multiple of the kernel's 4-pixel vectors, an unaligned view, and the copy into a second buffer"""
    from sdfest_amd.pipeline import preprocess_depth
    g = torch.Generator().manual_seed(3)
    for shape, far in (((3, 37, 53), 2.0), ((1, 480, 640), 1.5), ((2, 48, 64), None), ((1, 5, 3), 0.7)):
        depth = (torch.rand(shape, generator=g) * 3.0).cuda()
        depth.view(-1)[::97] = float("nan")
        depth.view(-1)[5::131] = float("inf")
        depth.view(-1)[7::89] = 0.0
        mask = (torch.rand(shape, generator=g) < 0.6).cuda()
        ref = depth.clone()
        ref[~mask] = 0
        if far is not None:
            ref[ref > far] = 0
        for as_uint8 in (False, True):
            got = depth.clone()
            dst = torch.full(shape, -1.0, device="cuda")
            out = preprocess_depth(got, mask.to(torch.uint8) if as_uint8 else mask, far, copy_to=(dst, 0, shape[0]))
            assert out is got
            for t in (got, dst):
                assert torch.equal(t.view(torch.int32), ref.view(torch.int32)), (shape, far)
    # a shard of the batch copied elsewhere; no mask at all
    depth = (torch.rand((4, 24, 32), generator=g) * 3.0).cuda()
    dst = torch.zeros((2, 24, 32), device="cuda")
    ref = depth.clone(); ref[ref > 1.0] = 0
    preprocess_depth(depth, None, 1.0, copy_to=(dst, 1, 3))
    assert torch.equal(depth, ref) and torch.equal(dst, ref[1:3])
    with pytest.raises(RuntimeError):
        preprocess_depth(depth.cpu(), None, 1.0)
    with pytest.raises(RuntimeError):
        preprocess_depth(depth, torch.ones((4, 24, 31), dtype=torch.bool, device="cuda"))


@pytest.mark.parametrize("order", [0, 1])
def test_resident_point_sets_equal_the_two_call_form(order):
    """sdfr_depth_to_points_resident (counts and offsets on the device, points at capacity) against sdfr_depth_count_ordered
    + host prefix + sdfr_depth_to_points_ordered: the same points in the same order, bit for bit -- ragged image sizes,
    an empty view, a full view, both point orders"""
    from sdfest_amd import Camera, _lib
    from sdfest_amd.generated_views import depth_to_pointsets
    L = _lib.lib()
    rng = np.random.default_rng(31 + order)
    for W, H, V in ((37, 29, 4), (640, 480, 3), (1030, 3, 2), (64, 16, 70)):
        cam = Camera(W, H, 0.9 * W + 3.3, 1.1 * W + 1.7, 0.47 * W, 0.55 * H, pixel_center=0.5)
        dd = rng.uniform(0.3, 2.0, (V, H, W)).astype(np.float32)
        dd[rng.uniform(size=dd.shape) < 0.7] = 0
        dd[1] = 0
        if V > 2:
            dd[2] = 1.0
        depth = torch.tensor(dd, device="cuda")
        ref_pts, ref_counts = depth_to_pointsets(depth, cam, tiled=bool(order))
        fx, fy, cx0, cy0, _ = cam.get_pinhole_camera_parameters(0.0)
        counts = torch.full((V,), -1, dtype=torch.int32, device="cuda")
        offsets = torch.full((V + 1,), -1, dtype=torch.int32, device="cuda")
        ws = torch.empty(max(L.sdfr_depth_points_workspace_bytes(V, W, H), 256), dtype=torch.uint8, device="cuda")
        pts = torch.full((V * W * H, 3), float("nan"), device="cuda")
        _lib.check(L.sdfr_depth_to_points_resident(depth.data_ptr(), V, W, H, order, 1.0 / fx, 1.0 / fy, cx0, cy0,
                                                   counts.data_ptr(), offsets.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   pts.data_ptr(), 0, torch.cuda.current_stream().cuda_stream), "resident")
        assert counts.tolist() == ref_counts.tolist() == [(dd[v] != 0).sum() for v in range(V)]
        assert offsets.tolist() == [0] + np.cumsum(ref_counts.cpu().numpy()).tolist()
        n = int(offsets[-1])
        assert torch.equal(pts[:n], ref_pts) and torch.isnan(pts[n:]).all()       # nothing written beyond the count
