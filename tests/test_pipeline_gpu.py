"""GPU: the render-and-compare iteration (SDFPipeline.__call__ hot loop) against a numpy restatement
built from the oracle's render / sampler forward+backward, and the C5 convergence run."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def qinv(q):
    return q * np.array([-1, -1, -1, 1.0])


def qrot(q, v):
    return qmul(qmul(q, np.append(v, 0.0)), qinv(q))[:3]


def left_mul_matrix(a):
    """M with qmul(a, b) = M @ b."""
    ax, ay, az, aw = a
    return np.array([[aw, -az, ay, ax], [az, aw, -ax, ay], [-ay, ax, aw, az], [-ax, -ay, -az, aw]])


class NumpyAdam:
    def __init__(self, lrs):
        self.lrs, self.m, self.v, self.t = lrs, [0.0] * len(lrs), [0.0] * len(lrs), 0

    def step(self, params, grads):
        self.t += 1
        out = []
        for i, (p, g) in enumerate(zip(params, grads)):
            self.m[i] = 0.9 * self.m[i] + 0.1 * g
            self.v[i] = 0.999 * self.v[i] + 0.001 * g * g
            mh = self.m[i] / (1 - 0.9 ** self.t)
            vh = self.v[i] / (1 - 0.999 ** self.t)
            out.append(p - self.lrs[i] * mh / (np.sqrt(vh) + 1e-8))
        return out


def oracle_loop(sdf, depth_images, cam, cam_pos, cam_quat, p, q, s, thr, iters, wd, wpc):
    """numpy float64 restatement of simple_setup.py:408-462 with shape_optimization=False."""
    W, H, fx, fy, cx, cy = cam
    V = depth_images.shape[0]
    clouds = [oracle.depth_to_pointcloud(d, fx, fy, cx - 0.5, cy - 0.5, dtype=np.float64) for d in depth_images]
    adam = NumpyAdam([1e-3, 1e-2, 1e-3])
    traj = []
    for _ in range(iters):
        nq = q / np.linalg.norm(q)
        gp, gnq, gs = np.zeros(3), np.zeros(4), 0.0
        for v in range(V):
            qw2c = qinv(cam_quat[v])
            Rw2c = np.stack([qrot(qw2c, e) for e in np.eye(3)], axis=1)
            pc = Rw2c @ (p - cam_pos[v])
            qc = qmul(qw2c, nq)
            est = oracle.render_forward(sdf, pc, qc, [1.0 / s], W, H, cx, cy, fx, fy, thr, dtype=np.float64)[0]
            mask = (depth_images[v] > 0) & (est > 0)
            gimg = wd * np.sign(est - depth_images[v]) * mask / mask.sum()
            _, g_pc, g_qc, g_is = oracle.render_backward(gimg, est, sdf, pc, qc, [1.0 / s], cx, cy, fx, fy,
                                                         dtype=np.float64)
            val = oracle.pc_loss_forward(clouds[v], pc, qc, s, sdf, dtype=np.float64)
            go = wpc * np.sign(val) / len(val)
            _, g_pc2, g_qc2, g_s2 = oracle.pc_loss_backward(go, clouds[v], pc, qc, s, sdf, dtype=np.float64)
            gp += Rw2c.T @ (g_pc[0] + g_pc2)
            gnq += left_mul_matrix(qw2c).T @ (g_qc[0] + g_qc2)
            gs += -g_is[0] / s ** 2 + g_s2
        n = np.linalg.norm(q)
        gq = (gnq - nq * (nq @ gnq)) / n
        p, q, s = adam.step([p, q, np.array(s)], [gp, gq, np.array(gs)])
        s = float(s)
        q = q / np.linalg.norm(q)
        traj.append((p.copy(), q.copy(), s))
    return traj


@pytest.fixture(scope="module")
def mug_decoder():
    from sdfest_amd import SDFDecoder
    from test_decoder_gpu import mug_config
    d = np.load(os.path.join(GOLDEN, "decoder_mug.npz"))
    w = np.load(os.path.join(GOLDEN, "mug_decoder_weights.npz"))
    return SDFDecoder.from_config(mug_config(d), {k: w[k] for k in w.files}), d


def test_iteration_matches_numpy_restatement(mug_decoder):
    """5 Adam iterations, 2 cameras with non-trivial extrinsics, 64x48, mug shape at z=0 (pose only,
    shape_optimization=False): parameter trajectory vs the oracle-based loop."""
    from sdfest_amd import Camera
    from sdfest_amd.pipeline import RenderAndCompare
    dec, d = mug_decoder
    sdf = d["z0_full"].astype(np.float64)
    W, H, f = 64, 48, 60.0
    cam = Camera(W, H, f, f, W / 2, H / 2, pixel_center=0.5)
    camt = (W, H, f, f, W / 2, H / 2)
    cam_pos = np.array([[0.0, 0.0, 0.0], [0.25, 0.05, 0.02]])
    cq = np.array([0.02, 0.27, 0.01, 1.0]); cq /= np.linalg.norm(cq)
    cam_quat = np.stack([np.array([0.0, 0, 0, 1.0]), cq])
    p_true = np.array([0.01, -0.015, -0.45]); s_true = 0.11
    q_true = np.array([0.3, 0.5, -0.1, 0.8]); q_true /= np.linalg.norm(q_true)
    obs = []
    for v in range(2):
        qw2c = qinv(cam_quat[v])
        obs.append(oracle.render_forward(sdf, qrot(qw2c, p_true - cam_pos[v]), qmul(qw2c, q_true), [1 / s_true],
                                         W, H, W / 2, H / 2, f, f, 0.005, dtype=np.float64)[0])
    obs = np.stack(obs)
    assert (obs > 0).sum(axis=(1, 2)).min() > 150
    p0 = p_true + np.array([0.008, -0.006, 0.01]); s0 = 0.12
    q0 = q_true + np.array([0.04, -0.03, 0.02, 0.01])
    cfg = {"threshold": 0.005, "max_iterations": 5, "depth_weight": 1.0, "pc_weight": 3.0}
    ref = oracle_loop(sdf, obs, camt, cam_pos, cam_quat, p0.copy(), q0.copy(), s0, 0.005, 5, 1.0, 3.0)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda")
    hist = []
    loop = RenderAndCompare(dec, cam, cfg)
    loop(t(obs), t(p0[None]), t(q0[None]), t([s0]), torch.zeros(1, 8, device="cuda"),
         camera_positions=t(cam_pos), camera_orientations=t(cam_quat), shape_optimization=False,
         history=hist)
    for it, (rp, rq, rs) in enumerate(ref):
        hp = hist[it]["position"].cpu().numpy()[0]
        hq = hist[it]["orientation"].cpu().numpy()[0]
        hs = hist[it]["scale"].item()
        # Adam normalises the gradient, so steps are ~lr; agree to a small fraction of a step
        assert np.max(np.abs(hp - rp)) < 2e-4 * (it + 1), (it, hp, rp)
        assert np.max(np.abs(hq - rq)) < 2e-3 * (it + 1), (it, hq, rq)
        assert abs(hs - rs) < 2e-4 * (it + 1), (it, hs, rs)
    # and the steps really moved the parameters
    assert np.linalg.norm(ref[-1][0] - p0) > 2e-3


def test_c5_full_loop_converges(mug_decoder):
    """BASELINE configs[4]: decoder(z) -> 64^3 SDF -> render, 50 Adam steps on a synthetic depth
    image, mug config, 640x480, shape optimisation on."""
    from sdfest_amd import Camera, render_depth_gpu
    from sdfest_amd.pipeline import RenderAndCompare
    dec, d = mug_decoder
    cam = Camera(640, 480, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
    dev = "cuda"
    z_true = torch.tensor(d["z"][9:10], device=dev) * 0.5
    p_true = torch.tensor([[0.02, -0.01, -0.5]], device=dev)
    q_true = torch.tensor([[0.2, 0.6, -0.15, 0.75]], device=dev)
    q_true = q_true / q_true.norm()
    s_true = torch.tensor([0.055], device=dev)
    with torch.no_grad():
        target = render_depth_gpu(dec.decode(z_true)[0, 0], p_true[0], q_true[0], 1 / s_true[0], None, None,
                                  None, 0.005, cam)
    assert (target > 0).sum() > 2000
    cfg = {"threshold": 0.005, "max_iterations": 50, "depth_weight": 1.0, "pc_weight": 3.0}
    hist = []
    p0 = p_true + torch.tensor([[0.01, 0.01, 0.01]], device=dev)
    q0 = q_true + torch.tensor([[0.06, -0.05, 0.04, 0.0]], device=dev)
    out = RenderAndCompare(dec, cam, cfg)(target[None], p0, q0 / q0.norm(), torch.tensor([0.06], device=dev),
                                          torch.zeros(1, 8, device=dev), history=hist)
    losses = [h["loss"].item() for h in hist]
    assert all(np.isfinite(losses))
    assert min(losses[-5:]) < 0.5 * losses[0], losses
    pos_err0 = (p0 - p_true).norm().item()
    pos_err = (out[0] - p_true).norm().item()
    assert pos_err < 0.6 * pos_err0, (pos_err0, pos_err)
    assert abs(out[2].item() - 0.055) < abs(0.06 - 0.055)
    assert hist[-1]["latent"].abs().max().item() > 1e-3      # the latent really was optimised


@pytest.mark.parametrize("shape_opt,fuse_l1", [(False, True), (True, True), (True, False)])
def test_fused_graph_loop_matches_autograd_loop(mug_decoder, shape_opt, fuse_l1):
    """The launch-sequence / hipGraph iteration against the autograd-driven one: same parameter
    trajectory (2 cameras, 160x120, 8 iterations), eager and graph-replayed."""
    from sdfest_amd import Camera
    from sdfest_amd.pipeline import FusedRenderAndCompare, RenderAndCompare
    dec, d = mug_decoder
    W, H, f = 160, 120, 150.0
    cam = Camera(W, H, f, f, W / 2, H / 2, pixel_center=0.5)
    dev = "cuda"
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    cam_pos = t([[0.0, 0.0, 0.0], [0.25, 0.05, 0.02]])
    cq = np.array([0.02, 0.27, 0.01, 1.0]); cq /= np.linalg.norm(cq)
    cam_quat = t([[0, 0, 0, 1.0], cq])
    p_true = t([[0.01, -0.015, -0.45]]); s_true = t([0.11])
    q_true = t([[0.3, 0.5, -0.1, 0.8]]); q_true = q_true / q_true.norm()
    z_true = t(d["z"][10:11]) * 0.3
    cfg = {"threshold": 0.005, "max_iterations": 8, "depth_weight": 1.0, "pc_weight": 3.0}
    ref_loop = RenderAndCompare(dec, cam, cfg)
    with torch.no_grad():
        sdf = dec.decode(z_true)[0, 0]
        _, _, obs = ref_loop.losses(torch.ones((2, H, W), device=dev), torch.zeros((0, 3), device=dev),
                                    None, [], cam_pos, cam_quat, p_true, q_true, s_true, sdf)
    obs = obs.contiguous()
    assert (obs > 0).sum(dim=(1, 2)).min() > 1000
    p0 = p_true + t([[0.008, -0.006, 0.01]]); s0 = t([0.12])
    q0 = q_true + t([[0.04, -0.03, 0.02, 0.01]]); z0 = torch.zeros(1, 8, device=dev)
    h_ref = []
    ref_loop(obs, p0, q0, s0, z0, camera_positions=cam_pos, camera_orientations=cam_quat,
             shape_optimization=shape_opt, history=h_ref)
    fused = FusedRenderAndCompare(dec, cam, cfg, obs, cam_pos, cam_quat, shape_optimization=shape_opt,
                                  fuse_depth_loss=fuse_l1)
    for use_graph in (False, True, True):
        h = []
        out = fused(p0, q0, s0, z0, use_graph=use_graph, history=h)
        for it in range(cfg["max_iterations"]):
            for key, tol in (("position", 2e-5), ("orientation", 2e-4), ("scale", 2e-5), ("latent", 5e-4)):
                a, b = h[it][key].reshape(-1), h_ref[it][key].reshape(-1)
                assert (a - b).abs().max().item() <= tol * (it + 1), (use_graph, it, key, a, b)
            assert abs(h[it]["loss"].item() - h_ref[it]["loss"].item()) <= 2e-3 * abs(h_ref[it]["loss"].item())
        assert torch.equal(out[0], h[-1]["position"])
    if shape_opt:
        assert h[-1]["latent"].abs().max().item() > 1e-3
    else:
        assert h[-1]["latent"].abs().max().item() == 0.0


@pytest.mark.parametrize("fuse_l1,many", [(True, False), (False, False), (True, True)])
def test_merged_reduction_launches_give_the_same_trajectory(mug_decoder, fuse_l1, many):
    """The merged launches of the loop (sdfr_views_to_pose_grad_deferred: the renderer's and the sampler's per-view
    reductions inside the gradient chain's launch; sdfr_render_backward_l1_pc: both backward passes side by side)
    against one launch each: the same sums in the same order, so a pose-only run (no atomics on the way to the pose
    gradients) agrees to the last bit or two (the compiler contracts a*b + c*d per kernel) and each form repeats
    bit for bit, eager and graph-replayed; 3 views of 160x120 -- and (`many`) 28 views of 640x480, which is a
    batch launch of the backward: 64 x 8 or 32 x 32 tiles per view, another layout of the deferred partials."""
    from sdfest_amd import Camera
    from sdfest_amd.pipeline import FusedRenderAndCompare, RenderAndCompare
    dec, d = mug_decoder
    W, H, f = (640, 480, 600.0) if many else (160, 120, 150.0)
    cam = Camera(W, H, f, f, W / 2, H / 2, pixel_center=0.5)
    dev = "cuda"
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    cq = np.array([0.02, 0.27, 0.01, 1.0]); cq /= np.linalg.norm(cq)
    cq2 = np.array([-0.05, -0.2, 0.03, 1.0]); cq2 /= np.linalg.norm(cq2)
    if many:   # 28 cameras: near ones see >= 2 pixels per voxel (32 x 32 tiles), far ones fewer (64 x 8)
        rng = np.random.default_rng(4)
        V = 28
        cp = rng.uniform(-0.05, 0.05, (V, 3)); cp[:, 2] = np.where(np.arange(V) % 2 == 0, 0.0, 1.2)
        cqs = np.concatenate([rng.uniform(-0.03, 0.03, (V, 3)), np.ones((V, 1))], axis=1)
        cam_pos, cam_quat = t(cp), t(cqs / np.linalg.norm(cqs, axis=1, keepdims=True))
    else:
        cam_pos = t([[0.0, 0.0, 0.0], [0.25, 0.05, 0.02], [-0.2, 0.1, 0.0]])
        cam_quat = t([[0, 0, 0, 1.0], cq, cq2])
    V = cam_pos.shape[0]
    p_true = t([[0.01, -0.015, -0.45]]); s_true = t([0.11])
    q_true = t([[0.3, 0.5, -0.1, 0.8]]); q_true = q_true / q_true.norm()
    cfg = {"threshold": 0.005, "max_iterations": 3 if many else 6, "depth_weight": 1.0, "pc_weight": 3.0}
    with torch.no_grad():
        sdf = dec.decode(t(d["z"][10:11]) * 0.3)[0, 0]
        _, _, obs = RenderAndCompare(dec, cam, cfg).losses(
            torch.ones((V, H, W), device=dev), torch.zeros((0, 3), device=dev), None, [], cam_pos, cam_quat,
            p_true, q_true, s_true, sdf)
    obs = obs.contiguous()
    assert (obs > 0).sum(dim=(1, 2)).min() > 500
    p0 = p_true + t([[0.008, -0.006, 0.01]]); s0 = t([0.12])
    q0 = q_true + t([[0.04, -0.03, 0.02, 0.01]]); z0 = t(d["z"][10:11]) * 0.3
    runs = {}
    for merged in (False, True):
        loop = FusedRenderAndCompare(dec, cam, cfg, obs, cam_pos, cam_quat, shape_optimization=False,
                                     fuse_depth_loss=fuse_l1, merge_launches=merged)
        assert loop.defer_pose == merged
        for use_graph in (False, True):
            h = []
            loop(p0, q0, s0, z0, use_graph=use_graph, history=h)
            runs[(merged, use_graph)] = h
    ref = runs[(False, False)]
    assert (ref[-1]["position"] - p0).abs().max().item() > (3e-4 if many else 1e-3)
    if many:   # both tilings took part
        ratio = f * (2 * 0.12 / 63) / np.linalg.norm(p0.cpu().numpy() - cam_pos.cpu().numpy(), axis=1)
        assert (ratio >= 2.0).any() and (ratio < 2.0).any(), ratio
    for key, h in runs.items():
        same_form = runs[(key[0], False)]
        for it in range(cfg["max_iterations"]):
            for name in ("position", "orientation", "scale", "loss"):
                assert torch.equal(h[it][name], same_form[it][name]), (key, it, name)
                err = (h[it][name] - ref[it][name]).abs().max().item()
                assert err <= 1e-6 * (it + 1), (key, it, name, h[it][name], ref[it][name])


def test_graph_replays_are_repeatable_without_host_work_between_them(mug_decoder):
    """Regression: 20 back-to-back replays per call, several calls on one captured graph, no host
    work between replays -- every call must reproduce the eager launch sequence.  (With
    hipMemsetAsync / hipMemcpyAsync nodes in the captured sequence the second and later calls
    went elsewhere; the entry points now launch kernels only.)"""
    from sdfest_amd import Camera, render_depth_gpu
    from sdfest_amd.pipeline import FusedRenderAndCompare
    dec, d = mug_decoder
    cam = Camera(320, 240, 160.0, 160.0, 160.0, 120.0, pixel_center=0.5)
    dev = "cuda"
    z_true = torch.tensor(d["z"][9:10], device=dev) * 0.5
    p_true = torch.tensor([[0.02, -0.01, -0.5]], device=dev)
    q_true = torch.tensor([[0.2, 0.6, -0.15, 0.75]], device=dev)
    q_true = q_true / q_true.norm()
    s_true = torch.tensor([0.055], device=dev)
    with torch.no_grad():
        target = render_depth_gpu(dec.decode(z_true)[0, 0], p_true[0], q_true[0], 1 / s_true[0], None, None,
                                  None, 0.005, cam)
    assert (target > 0).sum() > 500
    cfg = {"threshold": 0.005, "max_iterations": 20, "depth_weight": 1.0, "pc_weight": 3.0}
    q0 = q_true + torch.tensor([[0.06, -0.05, 0.04, 0.0]], device=dev)
    args = (p_true + 0.01, q0 / q0.norm(), torch.tensor([0.06], device=dev), torch.zeros(1, 8, device=dev))
    eager = FusedRenderAndCompare(dec, cam, cfg, target[None].contiguous())
    ref = [t.clone() for t in eager(*args, use_graph=False)]
    graph = FusedRenderAndCompare(dec, cam, cfg, target[None].contiguous())
    for call in range(4):
        out = graph(*args, use_graph=True)
        torch.cuda.synchronize()
        assert graph.step.item() == 20
        for a, b, tol in zip(out, ref, (2e-5, 2e-4, 2e-5, 5e-4)):
            assert torch.isfinite(a).all() and (a - b).abs().max().item() <= tol, (call, a, b)
    # several iterations per replayed graph (graph_iterations, default 5) with a remainder: 7 = 5 + 2 x 1, and one
    # iteration per graph, give the run of 7 eager iterations
    cfg7 = dict(cfg, max_iterations=7)
    ref7 = [t.clone() for t in FusedRenderAndCompare(dec, cam, cfg7, target[None].contiguous())(*args, use_graph=False)]
    for per_graph in (5, 3, 1):
        loop = FusedRenderAndCompare(dec, cam, cfg7, target[None].contiguous(), graph_iterations=per_graph)
        for call in range(2):
            out = loop(*args, use_graph=True)
            torch.cuda.synchronize()
            assert loop.step.item() == 7
            for a, b, tol in zip(out, ref7, (2e-5, 2e-4, 2e-5, 5e-4)):
                assert torch.isfinite(a).all() and (a - b).abs().max().item() <= tol, (per_graph, call, a, b)


def test_empty_overlap_view_gives_nan_loss_and_finite_steps(mug_decoder):
    """simple_setup.py:125-131: `torch.mean(depth_error[overlap_mask])` of an empty overlap is NaN, yet no pixel
    is selected, so that view contributes no gradient: the reference's loss is NaN while its Adam steps stay
    finite.  Both loops do the same, and agree with each other."""
    from sdfest_amd import Camera
    from sdfest_amd.pipeline import FusedRenderAndCompare, RenderAndCompare
    dec, d = mug_decoder
    sdf = d["z0_full"].astype(np.float64)
    W, H, f = 64, 48, 60.0
    cam = Camera(W, H, f, f, W / 2, H / 2, pixel_center=0.5)
    p_true, s_true = np.array([0.01, -0.015, -0.45]), 0.11
    q_true = np.array([0.3, 0.5, -0.1, 0.8]); q_true /= np.linalg.norm(q_true)
    obs0 = oracle.render_forward(sdf, p_true, q_true, [1 / s_true], W, H, W / 2, H / 2, f, f, 0.005, dtype=np.float64)[0]
    obs1 = np.zeros_like(obs0)
    obs1[2:6, 2:8] = 0.4                      # the second view saw something in a corner the estimate never covers
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda")
    obs = t(np.stack([obs0, obs1]))
    cfg = {"threshold": 0.005, "max_iterations": 3, "depth_weight": 1.0, "pc_weight": 3.0}
    p0, q0, s0, z0 = t((p_true + 0.005)[None]), t(q_true[None]), t([0.115]), torch.zeros(1, 8, device="cuda")
    h1, h2 = [], []
    RenderAndCompare(dec, cam, cfg)(obs, p0, q0, s0, z0, history=h1)
    FusedRenderAndCompare(dec, cam, cfg, obs)(p0, q0, s0, z0, use_graph=False, history=h2)
    for a, b in zip(h1, h2):
        assert torch.isnan(a["loss"]) and torch.isnan(b["loss"])
        for k in ("position", "orientation", "scale", "latent"):
            assert torch.isfinite(a[k]).all() and torch.isfinite(b[k]).all()
            assert torch.allclose(a[k].reshape(-1), b[k].reshape(-1), rtol=0, atol=2e-4), k
    assert (h1[-1]["position"] - p0).abs().max() > 1e-3       # and the first view did move the estimate


@pytest.mark.parametrize("views", [1, 9])
def test_one_wave_linear_backward_in_the_tail_is_bitwise_the_workgroup_form(mug_decoder, views):
    """The tail's backward of the mug decoder's narrow Linear stack (8 -> 20 -> 50) as ONE wave out of LDS, the other
    waves reducing the views meanwhile (fc_stack_backward_one_wave), against the one-workgroup form of wider stacks
    (sdfr_decoder_set_option(SDFR_DECODER_OPT_FC_ONE_WAVE, 0)): same fmaf chains in the same order.  Nine views (records form: the tail
    works from the records) in the deterministic d/dSDF mode -- no float atomics anywhere on the way --: a
    shape-optimising run agrees bit for bit, every iteration, eager and replayed.  One view (the tail reduces the view
    itself, three waves beside the Linear stack's one): the float atomics of d/dSDF leave their ~1e-7 of noise, as
    between any two runs of that form; tests/test_decoder_gpu.py compares the two Linear backward forms bit for bit."""
    from sdfest_amd import Camera, render_depth_gpu
    from sdfest_amd.differentiable_renderer import SDF_GRAD_DETERMINISTIC
    from sdfest_amd.pipeline import FusedRenderAndCompare
    from sdfest_amd._lib import lib
    L = lib()
    dec, d = mug_decoder
    cam = Camera(160, 120, 150.0, 150.0, 80.0, 60.0, pixel_center=0.5)
    dev = "cuda"
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    rng = np.random.default_rng(21)
    cp = rng.uniform(-0.04, 0.04, (views, 3)); cp[0] = 0.0
    cqs = np.concatenate([rng.uniform(-0.04, 0.04, (views, 3)), np.ones((views, 1))], axis=1); cqs[0] = [0, 0, 0, 1]
    cam_pos, cam_quat = t(cp), t(cqs / np.linalg.norm(cqs, axis=1, keepdims=True))
    z_true = t(d["z"][9:10]) * 0.5
    p_true = t([[0.02, -0.01, -0.5]]); s_true = t([0.055])
    q_true = t([[0.2, 0.6, -0.15, 0.75]]); q_true = q_true / q_true.norm()
    with torch.no_grad():
        sdf = dec.decode(z_true)[0, 0]
        obs = torch.stack([render_depth_gpu(sdf, *_to_camera(p_true[0], q_true[0], cam_pos[v], cam_quat[v]),
                                            1 / s_true[0], None, None, None, 0.005, cam) for v in range(views)])
    assert (obs > 0).sum(dim=(1, 2)).min() > 200
    cfg = {"threshold": 0.005, "max_iterations": 6, "depth_weight": 1.0, "pc_weight": 3.0}
    q0 = q_true + t([[0.05, -0.04, 0.03, 0.0]])
    args = (p_true + 0.008, q0 / q0.norm(), t([0.058]), torch.zeros(1, 8, device=dev))
    runs = {}
    for on in (1, 0):
        old = dec.set_option("fc_one_wave", on)
        try:
            loop = FusedRenderAndCompare(dec, cam, cfg, obs.contiguous(), cam_pos, cam_quat,
                                         sdf_grad_mode=SDF_GRAD_DETERMINISTIC if views >= 8 else 0)
            assert loop.records_form == (views >= 8)
            for use_graph in (False, True):
                h = []
                loop(*args, use_graph=use_graph, history=h)
                torch.cuda.synchronize()
                runs[(on, use_graph)] = h
        finally:
            dec.set_option("fc_one_wave", old)
    ref = runs[(0, False)]
    assert (ref[-1]["latent"]).abs().max().item() > 1e-3 and (ref[-1]["position"] - args[0]).abs().max().item() > 1e-3
    for key, h in runs.items():
        for it in range(cfg["max_iterations"]):
            for name in ("position", "orientation", "scale", "latent", "loss"):
                if views >= 8:
                    assert torch.equal(h[it][name], ref[it][name]), (key, it, name, h[it][name], ref[it][name])
                else:
                    err = (h[it][name] - ref[it][name]).abs().max().item()
                    assert err <= 2e-6 * (it + 1) * max(1.0, ref[it][name].abs().max().item()), (key, it, name, err)


def _to_camera(p, q, cam_p, cam_q):
    """object pose in the frame of a camera at (cam_p, cam_q), as simple_setup.py:424-430"""
    from sdfest_amd import pipeline as Q
    qc = cam_q * torch.tensor([-1.0, -1.0, -1.0, 1.0], device=cam_q.device)
    return Q.quaternion_apply(qc, p - cam_p), Q.quaternion_multiply(qc, q / q.norm())
