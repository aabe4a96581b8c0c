"""GPU: the render pair of a loop iteration as ONE launch (sdfr_render_step_fused_l1_pc -> sdfr_decoder_backward_latent_
deferred_scaled -> sdfr_loop_tail_fused; FusedRenderAndCompare(fused_render=True), the default of the tail form: up to 7
views, every view's depth term with its own weight / count and its own unscaled volume) against
the two launches it replaces (sdfr_render_step_forward_l1 + sdfr_render_step_backward_l1_pc), which the G7 goldens and
the oracle pin (tests/test_loop_g7_gpu.py, tests/test_render_l1_gpu.py).

What differs between the forms is WHERE the depth loss's  weight / count  is applied -- to every pixel's term before the
sums (two launches), to the sums (one launch: no tile knows the count before the launch ends) -- so:
  * depth images, overlap counts, loss values' inputs: bit for bit;
  * d loss / d (pose, scale, latent) before Adam: equal up to float rounding of the sums (1e-5 of each group's largest
    component is asserted; 1e-6 is typical);
  * trajectories: the same to a small multiple of that per iteration.
The reference's arithmetic for this path: simple_setup.py:408-462 (iteration), :129-135 (masked depth L1), :144 (point
L1), sdf_renderer_cuda.cu:300-468 (backward)."""
import ctypes

import numpy as np
import pytest
import torch

import _loop_scenes as S

pytestmark = pytest.mark.gpu


def _scene(V, iterations, W=None):
    s = S.build("seven", iterations=iterations)
    return dict(s, depth=s["depth"][:V].contiguous(), cam_pos=s["cam_pos"][:V].contiguous(),
                cam_quat=s["cam_quat"][:V].contiguous())


def _loop(s, fused, **kw):
    from sdfest_amd.pipeline import FusedRenderAndCompare
    return FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["depth"], s["cam_pos"], s["cam_quat"],
                                 fused_render=fused, form="tail", **kw)


def _group_scale(g):
    return np.array([np.abs(g[0:3]).max()] * 3 + [np.abs(g[3:7]).max()] * 4 + [abs(g[7])]
                    + ([np.abs(g[8:]).max()] * (len(g) - 8) if len(g) > 8 else []))


@pytest.mark.parametrize("views,shape", [(1, True), (1, False), (2, True), (3, False), (4, True), (7, True)])
@pytest.mark.parametrize("mode", [0, 1])
def test_first_gradient_equals_the_two_launch_form(views, shape, mode):
    """one eager iteration: the gradient vector Adam is given, the depth images, both loss values"""
    s = _scene(views, 1)
    got = {}
    for fused in (False, True):
        loop = _loop(s, fused, sdf_grad_mode=mode, shape_optimization=shape)
        assert loop.fused_render == fused
        loop(*s["init"], use_graph=False)
        torch.cuda.synchronize()
        ld, lp = loop.view_losses()
        got[fused] = (loop.grads.cpu().numpy().astype(np.float64), loop.plan.depth.clone(), ld.cpu().numpy(),
                      lp.cpu().numpy())
        if fused:
            # the consumers cleared what they read: the next step adds into zeros
            assert float(loop.plan.view_count.abs().max()) == 0.0
            if shape:
                assert float(loop.plan.g_depth.abs().max()) == 0.0
                assert float(loop.plan._g_sdf_ring[0].abs().max()) == 0.0
    assert torch.equal(got[True][1], got[False][1]), "depth images differ"
    g0, g1 = got[False][0], got[True][0]
    assert np.all(np.isfinite(g1)) and (np.abs(g0[8:]).max() > 0) == shape
    if not shape:
        g0, g1 = g0[:8], g1[:8]
    err = np.abs(g1 - g0) / _group_scale(g0)
    assert err.max() < 1e-5, err
    np.testing.assert_allclose(got[True][2], got[False][2], rtol=2e-6)    # depth loss: another order of the tile sums
    np.testing.assert_array_equal(got[True][3], got[False][3])           # point loss: the same blocks, the same order


@pytest.mark.parametrize("views,shape", [(1, True), (1, False), (2, True), (3, False), (5, True)])
@pytest.mark.parametrize("use_graph", [False, True])
def test_trajectory_follows_the_two_launch_form(views, shape, use_graph):
    s = _scene(views, 12)
    hist = {}
    for fused in (False, True):
        loop = _loop(s, fused, shape_optimization=shape, graph_iterations=5)
        h = []
        loop(*s["init"], use_graph=use_graph, history=h)
        torch.cuda.synchronize()
        hist[fused] = S.history_array(h)
        # ... and a second run on the same object starts from clean sums
        h2 = []
        loop(*s["init"], use_graph=use_graph, history=h2)
        torch.cuda.synchronize()
        np.testing.assert_allclose(S.history_array(h2), hist[fused], atol=2e-5 if shape else 0.0, rtol=0)
    d = np.abs(hist[True] - hist[False])
    # Adam's steps are +-lr at first whatever the gradient's size: a rounding-level difference of a gradient stays one
    assert d[:, :8].max() < 5e-6 and d[:, 8:].max() < 2e-4, (d[:, :8].max(), d[:, 8:].max())
    assert np.abs(hist[True][-1] - hist[True][0]).max() > 1e-3      # (something was optimised)


@pytest.mark.parametrize("use_graph", [False, True])
def test_the_next_linear_stack_in_the_tails_launch_changes_no_number(use_graph):
    """fc_in_tail: the tail's launch gets one workgroup per 256 outputs of the decoder's wide Linear layer; every one
    repeats the latent's share of the tail and forms its slice for the NEXT decode (sdfr_loop_tail_fused(decoder_tape),
    sdfr_decoder_forward_stage).  Same arithmetic on the same inputs: the trajectory must not move by a bit -- only the
    float atomics of d/dSDF (both forms have them) can, so the comparison is made where they cannot: the FIRST iteration
    bit for bit, the rest to the run-to-run spread of either form."""
    s = _scene(1, 12)
    hist = {}
    for fc in (False, True):
        loop = _loop(s, True, fc_in_tail=fc, graph_iterations=5)
        assert loop.fc_in_tail == fc
        runs = []
        for _ in range(2):
            h = []
            loop(*s["init"], use_graph=use_graph, history=h)
            torch.cuda.synchronize()
            runs.append(S.history_array(h))
        np.testing.assert_allclose(runs[1], runs[0], atol=2e-5, rtol=0)
        hist[fc] = runs[0]
    # iteration 1: Adam's first step is -lr sign(g) whatever the gradient's rounding; iteration 2 starts from the
    # Linear-stack output the first tail left (fc_in_tail) or from a decode of the same latent: the same SDF
    assert np.array_equal(hist[True][0], hist[False][0])
    np.testing.assert_allclose(hist[True], hist[False], atol=2e-5, rtol=0)
    assert np.abs(hist[True][-1, 8:] - hist[True][0, 8:]).max() > 1e-3     # the latent moved: later decodes differ


def test_the_tails_linear_stack_output_is_the_decoders_bit_for_bit():
    """after one iteration the tape's slot holds the wide layer's output for the UPDATED latent -- what a full decode of
    that latent writes there"""
    from sdfest_amd import _lib
    L = _lib.lib()
    s = _scene(1, 1)
    loop = _loop(s, True, fc_in_tail=True)
    loop(*s["init"], use_graph=False)
    torch.cuda.synchronize()
    n_fc = 8192
    got = loop.tape.view(torch.float32).clone()
    tape2 = torch.zeros_like(loop.tape)
    rc = L.sdfr_decoder_forward_stage(s["decoder"]._h, loop.latent.data_ptr(), 1, 0, None, tape2.data_ptr(),
                                      loop.ws_dec.data_ptr(), loop.ws_dec.numel(), None, 1)
    assert rc == 0
    torch.cuda.synchronize()
    ref = tape2.view(torch.float32)
    touched = ref != 0
    assert int(touched.sum()) > n_fc // 8             # (ReLU'd outputs: a good share is positive)
    assert torch.equal(got[touched], ref[touched])
    # ... and the convolutional part from there equals the whole decode
    out_a = torch.empty(64 ** 3, device="cuda"); out_b = torch.empty_like(out_a)
    assert L.sdfr_decoder_forward_stage(s["decoder"]._h, None, 1, 0, out_a.data_ptr(), tape2.data_ptr(),
                                        loop.ws_dec.data_ptr(), loop.ws_dec.numel(), None, 2) == 0
    assert L.sdfr_decoder_forward(s["decoder"]._h, loop.latent.data_ptr(), 1, 0, out_b.data_ptr(), tape2.data_ptr(),
                                  loop.ws_dec.data_ptr(), loop.ws_dec.numel(), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)
    assert L.sdfr_decoder_forward_stage(s["decoder"]._h, loop.latent.data_ptr(), 1, 0, out_a.data_ptr(), None,
                                        loop.ws_dec.data_ptr(), loop.ws_dec.numel(), None, 2) != 0      # no tape
    assert L.sdfr_decoder_forward_stage(s["decoder"]._h, loop.latent.data_ptr(), 1, 0, out_a.data_ptr(), tape2.data_ptr(),
                                        loop.ws_dec.data_ptr(), loop.ws_dec.numel(), None, 4) != 0      # no such stage


def test_decoders_without_a_narrow_linear_stack_keep_the_launch():
    s = _scene(1, 1)
    s["decoder"].set_option("fc_one_wave", 0)
    try:
        assert not s["decoder"].narrow_linear_stack()
        loop = _loop(s, True)
        assert loop.fused_render and not loop.fc_in_tail
        with pytest.raises(ValueError, match="fc_in_tail"):
            _loop(s, True, fc_in_tail=True)
    finally:
        s["decoder"].set_option("fc_one_wave", 1)
    assert s["decoder"].narrow_linear_stack()


def test_pose_only_run_is_reproducible_bit_for_bit():
    """no float atomic on the way to the pose: tile sums in a fixed order, the count a sum of integers"""
    s = _scene(2, 8)
    runs = []
    for _ in range(2):
        loop = _loop(s, True, shape_optimization=False)
        h = []
        loop(*s["init"], use_graph=True, history=h)
        torch.cuda.synchronize()
        runs.append(S.history_array(h))
    assert np.array_equal(runs[0], runs[1])


def test_decoder_whose_first_vjp_launch_is_not_the_fused_stage():
    """the scaled VJP on a decoder handle with the fused stages switched off: the two volumes are summed by a launch of
    their own (scaled_combine_kernel) -- same numbers as the stage that adds them on load, to rounding"""
    s = _scene(1, 1)
    got = {}
    for bits in (5, 0):
        s["decoder"].set_option("fused_single", bits)
        try:
            loop = _loop(s, True)
            loop(*s["init"], use_graph=False)
            torch.cuda.synchronize()
            got[bits] = loop.grads.cpu().numpy().astype(np.float64)
            assert float(loop.plan.g_depth.abs().max()) == 0.0 and float(loop.plan._g_sdf_ring[0].abs().max()) == 0.0
        finally:
            s["decoder"].set_option("fused_single", 5)
    err = np.abs(got[0] - got[5]) / _group_scale(got[5])
    assert err.max() < 1e-5, err


def test_an_object_out_of_sight_contributes_nothing():
    """no pixel overlaps: count 0 -> k = 0, the depth loss the reference's mean over nothing (NaN), no depth gradient"""
    s = _scene(1, 1)
    p0, q0, s0, z0 = s["init"]
    away = p0.clone()
    away[0, 0] += 5.0
    out = {}
    for fused in (False, True):
        loop = _loop(s, fused)
        loop(away, q0, s0, z0, use_graph=False)
        torch.cuda.synchronize()
        out[fused] = (loop.grads.cpu().numpy(), loop.view_losses()[0].cpu().numpy())
    assert np.isnan(out[True][1]).all() and np.isnan(out[False][1]).all()
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-5, atol=1e-9)


def test_which_loops_take_the_one_launch_form():
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    from sdfest_amd.pipeline import FusedRenderAndCompare
    assert _loop(_scene(1, 1), None).fused_render and _loop(_scene(4, 1), None).fused_render
    assert not _loop(_scene(7, 1), None).fused_render and _loop(_scene(7, 1), True).fused_render      # (slower there: opt-in)
    assert _loop(_scene(7, 1), None, shape_optimization=False).fused_render
    s9 = S.build("many", iterations=1)
    s9 = dict(s9, depth=s9["depth"][:9].contiguous(), cam_pos=s9["cam_pos"][:9].contiguous(),
              cam_quat=s9["cam_quat"][:9].contiguous())
    assert not _loop(s9, None).fused_render
    with pytest.raises(ValueError, match="fused_render"):
        _loop(s9, True)
    s8 = dict(s9, depth=s9["depth"][:8].contiguous(), cam_pos=s9["cam_pos"][:8].contiguous(),
              cam_quat=s9["cam_quat"][:8].contiguous())
    auto8 = FusedRenderAndCompare(s8["decoder"], s8["camera"], s8["config"], s8["depth"], s8["cam_pos"], s8["cam_quat"])
    assert auto8.records_form and not auto8.fused_render          # form="auto": the records form from 8 views on
    s = _scene(2, 1)
    det = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["depth"], s["cam_pos"], s["cam_quat"],
                                shape_optimization=False, sdf_grad_mode=SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES)
    assert not det.fused_render          # (the records form)
    with pytest.raises(ValueError, match="fused_render"):
        FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["depth"], s["cam_pos"], s["cam_quat"],
                              shape_optimization=False, fuse_depth_loss=False, fused_render=True)


def test_c_abi_argument_errors():
    from sdfest_amd import _lib
    L = _lib.lib()
    s = _scene(1, 1)
    loop = _loop(s, True)
    plan = loop.plan
    sdf = loop.sdf[0, 0]

    def call(B=1, mode=0, R=64, pts=loop.max_pts, ws_bytes=None):
        return L.sdfr_render_step_fused_l1_pc(
            sdf.data_ptr(), R, 0, loop.pos_c.data_ptr(), loop.quat_c.data_ptr(), loop.inv_scale.data_ptr(),
            loop.scale_v.data_ptr(), B, plan.W, plan.H, plan.cx, plan.cy, plan.fx, plan.fy, 0.005, loop.target.data_ptr(),
            plan.depth.data_ptr(), mode, None, None, plan.workspace.data_ptr(),
            plan.workspace.numel() if ws_bytes is None else ws_bytes, 3.0, loop.points.data_ptr(),
            loop.offsets.data_ptr(), pts, loop.ws_pc.data_ptr(), loop.ws_pc.numel(), 0, None)

    inv, wsp = -1, -3      # SDFR_E_INVALID, SDFR_E_WORKSPACE (include/sdfr.h)
    assert call(B=9) == inv and b"views" in L.sdfr_last_error()
    assert call(mode=0x100) == inv and b"sdf_grad_mode" in L.sdfr_last_error()
    assert call(R=256) == inv
    assert call(pts=0) == inv
    assert call(ws_bytes=1024) == wsp
    t_mid = ctypes.c_void_p()
    rc = L.sdfr_decoder_backward_latent_deferred_scaled(
        s["decoder"]._h, loop.latent.data_ptr(), loop.tape.data_ptr(), plan._g_sdf_ring[0].data_ptr(), None, 1,
        plan.view_count.data_ptr(), 1.0, loop.ws_dec.data_ptr(), loop.ws_dec.numel(), None, ctypes.byref(t_mid))
    assert rc != 0 and b"NULL" in L.sdfr_last_error()


# ---- the C ABI call itself against the float64 oracle (ragged images, other grid sizes, off-centre intrinsics) --------

def _oracle_step(sdf, pos, quat, isc, cam, tgt, points, offsets, thr, w_depth, w_pc, mode):
    """float64: depth, the depth loss and its gradient image, the renderer's backward of it, and the point-cloud L1's
    backward (losses.py:32-135 with the upstream gradient of simple_setup.py:144's mean of |value|) -- per view"""
    import oracle
    W, H, cx, cy, fx, fy = cam
    B = len(pos)
    depth = oracle.render_forward(sdf, pos, quat, isc, W, H, cx, cy, fx, fy, thr, dtype=np.float64)
    loss, grad = oracle.depth_l1(depth, tgt, w_depth)
    g_sdf = np.zeros_like(sdf, dtype=np.float64)
    g_pose = np.zeros((B, 8))
    for v in range(B):
        gs, gp, gq, gi = oracle.render_backward(grad[v], depth[v], sdf, pos[v], quat[v], isc[v], cx, cy, fx, fy,
                                                sdf_grad_mode=mode, dtype=np.float64)
        g_sdf += gs
        g_pose[v, 0:3], g_pose[v, 3:7], g_pose[v, 7] = gp.reshape(3), gq.reshape(4), float(np.ravel(gi)[0])
    g_pc = np.zeros_like(g_sdf)
    for v in range(B):
        P = points[offsets[v]:offsets[v + 1]]
        val = oracle.pc_loss_forward(P, pos[v], quat[v], 1.0 / isc[v], sdf, dtype=np.float64)
        go = np.sign(val) * (w_pc / max(len(P), 1))
        g_pc += oracle.pc_loss_backward(go, P, pos[v], quat[v], 1.0 / isc[v], sdf, dtype=np.float64)[0]
    return depth, loss, g_sdf, g_pc, g_pose


@pytest.mark.parametrize("R,W,H,B,mode,shape", [(64, 640, 480, 1, 0, True), (40, 150, 101, 1, 1, True),
                                                (100, 97, 64, 2, 0, True), (64, 160, 120, 3, 0, False),
                                                (33, 90, 50, 2, 1, False), (64, 160, 120, 5, 1, True)])
def test_one_launch_step_against_the_float64_oracle(R, W, H, B, mode, shape):
    import oracle
    from sdfest_amd import BatchRenderPlan, Camera, _lib
    L = _lib.lib()
    f = 0.55 * W
    cx, cy = W / 2 + 3.25, H / 2 - 2.5                      # off-centre principal point
    sdf = oracle.blobs_sdf(1, R=R)
    pos, quat, isc = oracle.random_poses(B, seed=3, width=W, height=H, f=f)
    rng = np.random.default_rng(11)
    cam = (W, H, cx, cy, f, f)
    tgt = oracle.render_forward(sdf, pos + rng.normal(0, 0.01, pos.shape).astype(np.float32), quat, isc, *cam, 0.005,
                                dtype=np.float32)
    tgt = np.where(rng.uniform(size=tgt.shape) < 0.1, 0.0, tgt).astype(np.float32)
    # observed points in the camera frame: the back-projection of the targets (pointset_utils.depth_to_pointcloud :57-77)
    pts, offs = [], [0]
    for v in range(B):
        p = oracle.depth_to_pointcloud(tgt[v], f, f, cx - 0.5, cy - 0.5, dtype=np.float32)
        pts.append(p)
        offs.append(offs[-1] + len(p))
    points = np.concatenate(pts).astype(np.float32)
    assert min(np.diff(offs)) > 50
    w_depth, w_pc, thr = 1.0, 3.0, 0.005
    ref_depth, ref_loss, ref_gsdf, ref_gpc, ref_pose = _oracle_step(sdf, pos, quat, isc, cam, tgt, points, offs, thr,
                                                                    w_depth, w_pc, mode)
    T = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    camera = Camera(W, H, f, f, cx, cy, pixel_center=0.5)
    plan = BatchRenderPlan(R, B, camera, sdf_grad_mode=mode)
    d_sdf, d_pos, d_quat, d_isc, d_tgt = T(sdf), T(pos), T(quat), T(isc), T(tgt)
    d_scale = (1.0 / d_isc).contiguous()
    d_pts, d_off = T(points), T(offs, torch.int32)
    max_pts = int(max(np.diff(offs)))
    ws_pc = torch.zeros(max(L.sdfr_pc_loss_backward_workspace_bytes(B, max_pts), 256), dtype=torch.uint8, device="cuda")
    g_pc = torch.zeros((R, R, R), device="cuda") if shape else None
    depth = plan.step_fused_l1_pc(d_sdf, d_pos, d_quat, d_isc, d_scale, thr, d_tgt, d_pts, d_off, max_pts, ws_pc,
                                  pc_weight=w_pc, g_sdf=g_pc)
    torch.cuda.synchronize()
    # depth: the forward kernel's, bit for bit -- and the oracle's to float rounding
    plain = BatchRenderPlan(R, B, camera).forward(d_sdf, d_pos, d_quat, d_isc, thr)
    assert torch.equal(depth, plain)
    hit = ref_depth > 0
    assert (depth.cpu().numpy() > 0).sum() == hit.sum() or abs(int((depth > 0).sum()) - int(hit.sum())) <= 2
    # the overlap counts: integers
    cnt = plan.view_count.cpu().numpy().astype(np.float64)
    ref_cnt = ((tgt > 0) & hit).sum(axis=(1, 2))
    assert np.all(np.abs(cnt - ref_cnt) <= 2), (cnt, ref_cnt)
    k = np.where(cnt > 0, w_depth / np.maximum(cnt, 1), 0.0)
    if shape:
        g = g_pc.cpu().numpy().astype(np.float64) + np.tensordot(k, plan.g_depth.cpu().numpy().astype(np.float64), 1)
        ref = ref_gsdf + ref_gpc
        from helpers import check_sdf_grad
        check_sdf_grad(g, ref, mode, int(hit.sum()), rel=1e-4, name=f"one launch R={R} {W}x{H}")
    # the tiles' pose sums, times k: the renderer's pose gradients
    off = plan.partials_offset
    ntx, nty = (W + 31) // 32, (H + 7) // 8
    part = plan.workspace[off:off + B * ntx * nty * 32].view(torch.float32).view(B, nty * ntx, 8).cpu().numpy().astype(np.float64)
    # (only the tiles of a view's rectangle are written; the plan's workspace starts zero-filled)
    got_pose = part.sum(axis=1) * k[:, None]
    scale = np.abs(ref_pose).max(axis=0) + 1e-12
    # the fp32 bound of the sums: 1e-4 of the sum of the per-pixel magnitudes is what tests/test_render_gpu.py asserts;
    # against each component's largest value over the views it stays well below 1e-3 on these scenes
    assert np.all(np.abs(got_pose - ref_pose) <= 1e-3 * scale), (got_pose, ref_pose)
    # the depth loss from the tiles' records
    lo = L.sdfr_render_fused_tile_loss_offset(R, B, W, H)
    rec = plan.workspace[lo:lo + B * ntx * nty * 32].view(torch.float32).view(B, nty * ntx, 8).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(rec[:, :, 0].sum(axis=1) / np.maximum(cnt, 1), np.nan_to_num(ref_loss), rtol=2e-4, atol=1e-7)
    np.testing.assert_array_equal(rec[:, :, 1].sum(axis=1), cnt)


# ---- an object that fills the image: the one-launch step pre-sums d/dSDF in the LDS tables (SDFR_FUSED_DIRECT_MAX_POINTS) --

def _close_scene(iterations):
    """the mug 0.17 m in front of a 320x240 camera: ~11 k observed pixels, above the bound of the straight atomics"""
    from sdfest_amd import Camera, render_depth_gpu
    dec, d = S.mug_decoder()
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda")
    cam = Camera(320, 240, 200.0, 200.0, 160.0, 120.0, pixel_center=0.5)
    p_true = np.array([0.005, -0.004, -0.17])
    q_true = np.array([0.2, 0.6, -0.15, 0.75]); q_true /= np.linalg.norm(q_true)
    s_true = 0.055
    with torch.no_grad():
        sdf = dec.decode(t(d["z"][9:10] * 0.5))[0, 0]
        depth = render_depth_gpu(sdf, t(p_true), t(q_true), t(1.0 / s_true), None, None, None, 0.005, cam)[None].contiguous()
    q0 = q_true + np.array([0.03, -0.02, 0.02, 0.0])
    cfg = {"threshold": 0.005, "max_iterations": iterations, "depth_weight": 1.0, "pc_weight": 3.0}
    return dict(decoder=dec, camera=cam, config=cfg, depth=depth, cam_pos=t([[0.0, 0.0, 0.0]]), cam_quat=t([[0.0, 0.0, 0.0, 1.0]]),
                init=(t(p_true[None] + 0.003), t((q0 / np.linalg.norm(q0))[None]), t([0.057]),
                      torch.zeros(1, int(d["latent_size"]), device="cuda")))


@pytest.mark.parametrize("mode", [0, 1])
def test_an_object_that_fills_the_image_takes_the_tables(mode):
    s = _close_scene(8)
    n_obs = int((s["depth"] > 0).sum())
    assert n_obs > 6144 + 2000, n_obs                   # (tuning.hpp: SDFR_FUSED_DIRECT_MAX_POINTS)
    got, hist = {}, {}
    for fused in (False, True):
        loop = _loop(s, fused, sdf_grad_mode=mode)
        h = []
        loop(*s["init"], use_graph=False, history=h)
        torch.cuda.synchronize()
        hist[fused] = S.history_array(h)
        one = _loop(s, fused, sdf_grad_mode=mode)
        one.cfg = dict(one.cfg, max_iterations=1)
        one(*s["init"], use_graph=False)
        torch.cuda.synchronize()
        got[fused] = (one.grads.cpu().numpy().astype(np.float64), one.plan.depth.clone())
    assert torch.equal(got[True][1], got[False][1])
    err = np.abs(got[True][0] - got[False][0]) / _group_scale(got[False][0])
    assert err.max() < 1e-5, err
    d = np.abs(hist[True] - hist[False])
    assert d[:, :8].max() < 5e-6 and d[:, 8:].max() < 2e-4, (d[:, :8].max(), d[:, 8:].max())


def test_rebinding_between_a_small_and_a_large_object_needs_no_capture():
    """straight atomics or tables is decided inside the launch, from the observed point set's size on the device: the
    captured graphs serve an object of any size (FusedRenderAndCompare.rebind)"""
    big = _close_scene(6)
    from sdfest_amd import render_depth_gpu
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda")
    with torch.no_grad():      # the same mug three times as far: a ninth of the pixels
        sdf = big["decoder"].decode(torch.tensor(S.mug_decoder()[1]["z"][9:10] * 0.5, device="cuda"))[0, 0]
        p_far = big["init"][0].clone()
        p_far[0, 2] = -0.51
        small_depth = render_depth_gpu(sdf, p_far[0], big["init"][1][0], 1.0 / big["init"][2][0], None, None, None, 0.005,
                                       big["camera"])[None].contiguous()
    assert int((small_depth > 0).sum()) < 6144 < int((big["depth"] > 0).sum())
    small = dict(big, depth=small_depth, init=(p_far + 0.003,) + tuple(big["init"][1:]))
    ref = {}
    for name, s in (("big", big), ("small", small)):
        loop = _loop(s, True)
        ref[name] = [x.clone() for x in loop(*s["init"], use_graph=True)]
    loop = _loop(big, True)
    graphs = None
    for name, s in (("big", big), ("small", small), ("big", big), ("small", small)):
        loop.rebind(s["depth"], s["cam_pos"], s["cam_quat"])
        out = loop(*s["init"], use_graph=True)
        torch.cuda.synchronize()
        for a, b, tol in zip(out, ref[name], (2e-6, 2e-5, 2e-6, 1e-4)):
            assert (a - b).abs().max().item() <= tol, (name, a, b)
        now = (id(loop.graph), id(loop.graph_many))
        assert graphs in (None, now)
        graphs = now
