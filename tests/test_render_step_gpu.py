"""GPU: a step (sdfr_render_step_forward + sdfr_render_step_backward, include/sdfr.h) against the two stand-alone
calls it replaces -- which the other tests pin against the oracle and the goldens -- and against the oracle itself.

The step's forward has a one-launch prologue (plane minima -> view set-up inside one launch, through tagged entries
in the workspace) and zero-fills the gradient volume; its backward starts from the forward's view records."""
import numpy as np
import pytest
import torch

import oracle
from helpers import check_sdf_grad, rel_err

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def make(B, W, H, f, seed, per_view_sdf=False, sdf_grad_mode=0):
    from sdfest_amd import BatchRenderPlan, Camera
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    pos, quat, isc = oracle.random_poses(B, seed=seed, width=W, height=H, f=f)
    g = np.random.default_rng(seed + 100).uniform(-1, 1, (B, H, W)).astype(np.float32)
    plans = [BatchRenderPlan(64, B, cam, per_view_sdf=per_view_sdf, sdf_grad_mode=sdf_grad_mode) for _ in range(2)]
    return plans, (dev(pos), dev(quat), dev(isc)), dev(g), (pos, quat, isc)


def run_step(plan, sdf, pose, g, thr=0.005):
    d = plan.forward(sdf, *pose, thr, prepare_backward=True)
    assert plan._step is not None
    before = plan.g_sdf
    out = plan.backward(g, sdf, *pose)
    assert plan._step is None and plan.g_sdf is not before            # the step path ran (volumes alternate)
    return d.clone(), [o.clone() for o in out]


def run_separate(plan, sdf, pose, g, thr=0.005):
    d = plan.forward(sdf, *pose, thr)
    out = plan.backward(g, sdf, *pose)
    return d.clone(), [o.clone() for o in out]


def assert_same(step, ref, name=""):
    (d, (gs, gp, gq, gi)), (d0, (gs0, gp0, gq0, gi0)) = step, ref
    assert torch.equal(d, d0), name
    assert (d0 > 0).sum().item() > 100, name
    # pose gradients: fixed-order sums of the same per-tile partials, but the order follows the view's rectangle
    # (the step's is the tighter may-hit rectangle): equal up to the rounding of a sum of a few hundred terms
    for a, b in ((gp, gp0), (gq, gq0), (gi, gi0)):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 2e-5, name
    assert rel_err(gs.cpu().numpy(), gs0.cpu().numpy()) <= 1e-5, name      # float-atomic order only


@pytest.mark.parametrize("B,W,H,f", [(256, 200, 136, 300.0),     # batch tiles, packed records, one-launch prologue
                                     (18, 320, 240, 160.0),      # small tiles, packed records (>= 17 views), a last
                                                                 # set-up block with two views
                                     (8, 640, 480, 320.0),       # small tiles, plain grid behind the set-up launch
                                     (5, 320, 240, 160.0),       # a last set-up block with one view
                                     (2, 160, 120, 80.0),        # plain grid (no records, no plane minima):
                                     (1, 640, 480, 320.0)])      # ... zero fill + set-up are ONE prologue launch
def test_step_equals_the_two_standalone_calls(B, W, H, f):
    (p_step, p_ref), pose, g, _ = make(B, W, H, f, seed=11)
    grids = [dev(oracle.blobs_sdf(0)), dev(oracle.blobs_sdf(1)), dev(oracle.blobs_sdf(0) + 0.08)]
    # several steps on ONE workspace with a different grid each time: the plane minima a step's set-up reads must
    # be the ones its own launch published (epoch), never an earlier step's
    for k in range(6):
        sdf = grids[k % 3]
        assert_same(run_step(p_step, sdf, pose, g), run_separate(p_ref, sdf, pose, g), f"step {k}")


def test_step_on_a_workspace_full_of_other_bytes():
    """The sync region of the workspace is never initialised by the caller: whatever it holds -- zeros, ones, random
    bytes, another call's partial sums -- cannot read as this launch's plane minima."""
    (p_step, p_ref), pose, g, _ = make(16, 320, 240, 160.0, seed=12)
    sdf = dev(oracle.blobs_sdf(0))
    ref = run_separate(p_ref, sdf, pose, g)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    for fill in ("zeros", "ones", "random", "standalone"):
        if fill == "zeros":
            p_step.workspace.zero_()
        elif fill == "ones":
            p_step.workspace.fill_(255)
        elif fill == "random":
            p_step.workspace.copy_(torch.randint(0, 256, p_step.workspace.shape, dtype=torch.uint8, device="cuda",
                                                 generator=gen))
        else:
            run_separate(p_step, sdf, pose, g)      # the stand-alone layouts overlap the step's sync region
        assert_same(run_step(p_step, sdf, pose, g), ref, fill)


def test_step_with_one_grid_per_view():
    (p_step, p_ref), pose, g, _ = make(6, 160, 120, 80.0, seed=13, per_view_sdf=True)
    sdf = dev(np.stack([oracle.blobs_sdf(k % 3) for k in range(6)]))
    step, ref = run_step(p_step, sdf, pose, g), run_separate(p_ref, sdf, pose, g)
    assert step[1][0].shape == (6, 64, 64, 64)
    assert_same(step, ref)


def test_batch_tiles_with_one_grid_per_view_against_the_oracle():
    """One SDF per view at a size that takes the 64 x 8 batch tiles (>= 16 384 of them: the view generator's shape,
    plain grids through the batch kernels -- no face records, no plane minima): the step equals the two stand-alone
    calls, and sampled views equal the oracle (depth, d/dSDF of the view's own volume, pose gradients)."""
    B, W, H, f = 132, 320, 240, 200.0
    (p_step, p_ref), pose, g, (pos, quat, isc) = make(B, W, H, f, seed=31, per_view_sdf=True)
    grids = [oracle.blobs_sdf(k) for k in range(3)]
    sdf = dev(np.stack([grids[k % 3] for k in range(B)]))
    step, ref = run_step(p_step, sdf, pose, g), run_separate(p_ref, sdf, pose, g)
    assert step[1][0].shape == (B, 64, 64, 64)
    assert_same(step, ref)
    d, (gs, gp, gq, gi) = step
    g_np = g.cpu().numpy()
    for v in (0, 57, 131):
        o = oracle.render_forward(grids[v % 3], pos[v:v + 1], quat[v:v + 1], isc[v:v + 1], W, H, W / 2.0, H / 2.0, f, f,
                                  0.005)[0]
        dv = d[v].cpu().numpy()
        assert ((o > 0) != (dv > 0)).sum() <= 2
        both = (o > 0) & (dv > 0)
        assert both.sum() > 200 and np.max(np.abs(o[both] - dv[both]) / o[both]) <= 2e-5
        og = oracle.render_backward(g_np[v:v + 1], dv[None], grids[v % 3], pos[v:v + 1], quat[v:v + 1], isc[v:v + 1],
                                    W / 2.0, H / 2.0, f, f, dtype=np.float64)
        assert rel_err(gs[v].cpu().numpy(), og[0]) <= 2e-4
        # (random-sign upstream gradient: the pose sums cancel heavily -- the scale is the largest component)
        scale = max(np.abs(og[1]).max(), np.abs(og[2]).max(), np.abs(og[3]).max(), 1.0)
        for a, b_ in ((gp[v], og[1][0]), (gq[v], og[2][0]), (gi[v:v + 1], og[3])):
            assert np.max(np.abs(a.cpu().numpy() - b_)) <= 2e-3 * scale


@pytest.mark.parametrize("R", [32, 100, 33])
def test_batch_step_at_other_resolutions(R):
    """The batch tiles with a shared grid of another size: 32 and 100 (records, one-launch prologue, the generic
    kernels), 33 (odd: two-launch prologue).  Step = stand-alone pair; a view's depth against the oracle."""
    from sdfest_amd import BatchRenderPlan, Camera
    B, W, H, f = 128, 320, 240, 200.0
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    pos, quat, isc = oracle.random_poses(B, seed=40 + R, width=W, height=H, f=f)
    pose = (dev(pos), dev(quat), dev(isc))
    g = dev(np.random.default_rng(R).uniform(-1, 1, (B, H, W)))
    grid = oracle.blobs_sdf(1, R=R)
    sdf = dev(grid)
    p_step, p_ref = BatchRenderPlan(R, B, cam), BatchRenderPlan(R, B, cam)
    step, ref = run_step(p_step, sdf, pose, g), run_separate(p_ref, sdf, pose, g)
    assert step[1][0].shape == (R, R, R)
    assert_same(step, ref, f"R={R}")
    for v in (3, 90):
        o = oracle.render_forward(grid, pos[v:v + 1], quat[v:v + 1], isc[v:v + 1], W, H, W / 2.0, H / 2.0, f, f, 0.005)[0]
        dv = step[0][v].cpu().numpy()
        assert ((o > 0) != (dv > 0)).sum() <= 2
        both = (o > 0) & (dv > 0)
        assert both.sum() > 200 and np.max(np.abs(o[both] - dv[both]) / o[both]) <= 2e-5


def test_step_replayed_from_a_graph_sees_each_replays_grid():
    """A captured step is replayed with the SAME kernel arguments: the epoch that separates one launch's plane
    minima from the next lives in the workspace and advances on the device."""
    (p_step, p_ref), pose, g, _ = make(32, 200, 136, 150.0, seed=14)
    grids = [oracle.blobs_sdf(0), oracle.blobs_sdf(1), oracle.blobs_sdf(0) + 0.1]
    sdf = dev(grids[0])
    run_step(p_step, sdf, pose, g)          # warm-up outside the capture
    run_step(p_step, sdf, pose, g)          # (two: the captured step then uses the first volume again)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            d = p_step.forward(sdf, *pose, 0.005, prepare_backward=True)
            gs, gp, gq, gi = p_step.backward(g, sdf, *pose)
    version = sdf._version
    for k in range(5):
        sdf.copy_(dev(grids[k % 3]))        # same storage, new values: the graph reads what is there now
        graph.replay()
        torch.cuda.synchronize()
        ref = run_separate(p_ref, sdf, pose, g)
        assert_same((d, (gs, gp, gq, gi)), ref, f"replay {k}")
    assert sdf._version > version


def test_backward_after_changed_inputs_is_the_standalone_call():
    (p_step, p_ref), pose, g, _ = make(8, 160, 120, 80.0, seed=15)
    sdf = dev(oracle.blobs_sdf(0))
    p_step.forward(sdf, *pose, 0.005, prepare_backward=True)
    pose[0].add_(0.0)                       # an in-place write: not the tensors the forward prepared any more
    prepared = p_step._step[2]              # the volume the forward zero-filled
    out = p_step.backward(g, sdf, *pose)
    # stand-alone path, but into the volume the forward prepared (the previous one may still be in an exchange)
    assert p_step._step is None and p_step.g_sdf is prepared and out[0] is prepared
    ref = run_separate(p_ref, sdf, pose, g)
    assert rel_err(out[1].cpu().numpy(), ref[1][1].cpu().numpy()) <= 2e-5 and rel_err(out[0].cpu().numpy(), ref[1][0].cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("mode", [0, 1])
def test_step_at_the_bench_configuration_against_the_oracle(mode):
    """C3 as bench.py runs it (step path): depth hit count, d/dSDF and the pose gradients of a sample of views
    against the oracle -- with the exact d/dSDF weights and with the reference extension's (SDF_GRAD_CUDA_COMPAT,
    sdf_renderer_cuda.cu:373-388)."""
    import os
    (p_step, _), pose, g, (pos, quat, isc) = make(256, 640, 480, 320.0, seed=1, sdf_grad_mode=mode)
    sdf_np = oracle.blobs_sdf(0)
    d, (gs, gp, gq, gi) = run_step(p_step, dev(sdf_np), pose, g)
    assert int((d > 0).sum().item()) in range(4207800, 4207900)
    oracle.set_threads(min(64, os.cpu_count() or 1))
    dn, gn = d.cpu().numpy(), g.cpu().numpy()
    ob = oracle.render_backward(gn, dn, sdf_np, pos, quat, isc, 320.0, 240.0, 320.0, 320.0, dtype=np.float32,
                                sdf_grad_mode=mode)
    check_sdf_grad(gs.cpu().numpy(), ob[0], mode, int((dn > 0).sum()), 1e-4)
    pose_hip = np.concatenate([gp.cpu().numpy(), gq.cpu().numpy(), gi.cpu().numpy()[:, None]], axis=1)
    ref = np.concatenate([ob[1], ob[2], ob[3][:, None]], axis=1)
    sl = slice(0, 32)
    dimg = oracle.render_derivative_images(dn[sl], sdf_np, pos[sl], quat[sl], isc[sl], 320.0, 240.0, 320.0, 320.0,
                                           dtype=np.float32)
    l1 = np.abs(dimg * gn[sl][..., None]).sum(axis=(1, 2), dtype=np.float64)
    assert np.all(np.abs(pose_hip[sl] - ref[sl]) <= 1e-4 * l1)


def test_autograd_functions_run_the_step_and_survive_a_second_backward():
    """render_depth_gpu / render_depth_batch take the step path when a gradient is wanted (the step owns its
    workspace, so other renders may run between its two halves); a second backward through a retained graph
    takes the stand-alone call and gives the same gradients; without requires_grad the plain forward runs."""
    from sdfest_amd import Camera, render_depth_gpu, render_depth_batch
    from sdfest_amd import differentiable_renderer as dr
    W, H, f = 160, 120, 80.0
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    pos, quat, isc = oracle.random_poses(3, seed=21, width=W, height=H, f=f)
    sdf_np = oracle.blobs_sdf(0)
    g = dev(np.random.default_rng(3).uniform(-1, 1, (H, W)))
    calls = {"step": 0, "plain": 0}
    orig_step, orig_bwd = dr.step_backward_raw, dr.backward_raw

    def count_step(state, *a, **k):
        calls["step" if state is not None else "plain"] += 1
        return orig_step(state, *a, **k)
    dr.step_backward_raw = count_step
    try:
        leaves = [dev(sdf_np).requires_grad_(), dev(pos[0]).requires_grad_(), dev(quat[0]).requires_grad_(),
                  dev(isc[0]).requires_grad_()]
        d = render_depth_gpu(*leaves, None, None, None, 0.005, cam)
        other = render_depth_gpu(dev(sdf_np), dev(pos[1]), dev(quat[1]), dev(isc[1]), None, None, None, 0.005, cam)
        assert not other.requires_grad
        first = torch.autograd.grad((d * g).sum(), leaves, retain_graph=True)
        second = torch.autograd.grad((d * g).sum(), leaves)
        assert calls == {"step": 1, "plain": 1}
        ref = dr.backward_raw(g[None].contiguous(), d.detach()[None].contiguous(), dev(sdf_np), dev(pos[:1]),
                              dev(quat[:1]), dev(isc[:1]), W, H, W / 2.0, H / 2.0, f, f)
        for a, b, c in zip(first, second, ref):
            assert rel_err(a.cpu().numpy().ravel(), c.cpu().numpy().ravel()) <= 2e-5
            assert rel_err(b.cpu().numpy().ravel(), c.cpu().numpy().ravel()) <= 2e-5
        # batched function: two graphs alive at once, backward in the opposite order
        bl = [dev(sdf_np).requires_grad_(), dev(pos).requires_grad_(), dev(quat).requires_grad_(),
              dev(isc).requires_grad_()]
        d1 = render_depth_batch(*bl, 0.005, cam)
        d2 = render_depth_batch(bl[0], bl[1] * 1.01, bl[2], bl[3], 0.005, cam)
        gb = dev(np.random.default_rng(4).uniform(-1, 1, (3, H, W)))
        g2 = torch.autograd.grad((d2 * gb).sum(), bl)
        g1 = torch.autograd.grad((d1 * gb).sum(), bl)
        r1 = dr.backward_raw(gb, d1.detach(), dev(sdf_np), dev(pos), dev(quat), dev(isc), W, H, W / 2.0, H / 2.0, f, f)
        for a, c in zip(g1, r1):
            assert rel_err(a.cpu().numpy().ravel(), c.cpu().numpy().ravel()) <= 2e-5
        assert all(torch.isfinite(t).all() for t in g2)
        assert calls["step"] == 3
    finally:
        dr.step_backward_raw = orig_step


def test_plain_calls_sharing_one_workspace_with_a_changing_grid():
    """Round-2 advisor finding: plain forward + stand-alone backward on ONE workspace, 256 views of 640x480 (the
    backward's tile partials then cover everything behind the view records), a different grid every step.  The
    forward's set-up must never build its may-hit boxes from an earlier call's plane minima: depth equals a
    render on a fresh workspace, bit for bit.  (The sync region now has one place in every layout and the forward
    wipes its entries after use.)"""
    from sdfest_amd import BatchRenderPlan, Camera
    B, W, H, f = 256, 640, 480, 320.0
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    pos, quat, isc = oracle.random_poses(B, seed=1, width=W, height=H, f=f)
    pose = (dev(pos), dev(quat), dev(isc))
    g = torch.rand((B, H, W), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 2 - 1
    shared = BatchRenderPlan(64, B, cam)
    base = oracle.blobs_sdf(0)
    # grids whose surfaces differ a lot: a box built from the previous grid would cull rays of this one
    grids = [base, np.minimum(base + 0.25, oracle.blobs_sdf(1)), oracle.blobs_sdf(2) - 0.05, base + 0.1]
    for k, grid in enumerate(grids):
        sdf = dev(np.ascontiguousarray(grid, dtype=np.float32))
        d = shared.forward(sdf, *pose, 0.005).clone()
        shared.backward(g, sdf, *pose)                     # writes its partials over the forward's scratch
        fresh = BatchRenderPlan(64, B, cam)
        d0 = fresh.forward(sdf, *pose, 0.005)
        assert torch.equal(d, d0), f"grid {k}: depth differs from a render on a fresh workspace"
        assert (d0 > 0).sum().item() > 10000
        del fresh
    assert shared.prologue_fallbacks() == 0


def test_prologue_fallback_path_gives_the_same_depth_and_is_counted():
    """The set-up blocks of the one-launch prologue wait a bounded number of rounds for the plane minima and then
    set their views up without them (whole cube as the may-hit box).  Forced here through the workspace's poll bound: depth is
    bit-identical at C3, and the workspace's counter reports one fall-back per view."""
    from sdfest_amd import BatchRenderPlan, Camera, _lib
    B, W, H, f = 256, 640, 480, 320.0
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    pos, quat, isc = oracle.random_poses(B, seed=1, width=W, height=H, f=f)
    pose = (dev(pos), dev(quat), dev(isc))
    sdf = dev(oracle.blobs_sdf(0))
    plan = BatchRenderPlan(64, B, cam)
    d0 = plan.forward(sdf, *pose, 0.005).clone()
    assert plan.prologue_fallbacks() == 0
    # the workspace's own bound of the wait: sync header words 6 / 7 = {SDFR_SYNC_POLLS_MAGIC, rounds} (include/sdfr.h)
    o = _lib.lib().sdfr_render_sync_offset(B)
    words = plan.workspace[o + 24:o + 32].view(torch.int32)
    assert words.tolist() == [0, 0]
    words.copy_(torch.tensor([0x504F4C4C, 0], dtype=torch.int32))
    try:
        d1 = plan.forward(sdf, *pose, 0.005).clone()
        torch.cuda.synchronize()
    finally:
        words.zero_()
    assert plan.prologue_fallbacks() == B
    assert torch.equal(d0, d1)
    d2 = plan.forward(sdf, *pose, 0.005, prepare_backward=True)      # back on the normal path, step layout
    assert torch.equal(d0, d2) and plan.prologue_fallbacks() == B


@pytest.mark.parametrize("seed", [31, 32])
def test_half_grid_hint_changes_nothing_but_the_launch(seed):
    """SDFR_BWD_HALF_GRID is a performance hint: with it a batch launch has half the rows and a view that is not
    close takes its 64 x 8 tiles two per workgroup.  Same tiles, same per-tile sums: pose gradients bit for bit,
    d/dSDF up to the order of its float atomics -- for close views, far views and a mix, stand-alone and step."""
    from sdfest_amd import BatchRenderPlan, Camera
    from sdfest_amd.differentiable_renderer import close_view_fraction, views_are_close
    B, W, H, f = 64, 640, 480, 320.0
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    pos, quat, isc = oracle.random_poses(B, seed=seed, width=W, height=H, f=f)
    assert views_are_close(pos, isc, cam, 64) and close_view_fraction(pos, isc, cam, 64) > 0.95
    isc = isc.copy()
    isc[::3] *= 3.0                 # every third object a third of the size: ~1 pixel per voxel, 64 x 8 tiles
    assert not views_are_close(pos, isc, cam, 64) and 0.6 < close_view_fraction(pos, isc, cam, 64) < 0.7
    pose = (dev(pos), dev(quat), dev(isc))
    g = dev(np.random.default_rng(seed).uniform(-1, 1, (B, H, W)).astype(np.float32))
    sdf = dev(oracle.blobs_sdf(0))
    plain, hinted = BatchRenderPlan(64, B, cam), BatchRenderPlan(64, B, cam, close_views=True)
    for step in (False, True):
        outs = []
        for plan in (plain, hinted):
            d = plan.forward(sdf, *pose, 0.005, prepare_backward=step).clone()
            outs.append((d, [o.clone() for o in plan.backward(g, sdf, *pose)]))
        (d0, (gs0, gp0, gq0, gi0)), (d1, (gs1, gp1, gq1, gi1)) = outs
        assert torch.equal(d0, d1) and (d0 > 0).sum().item() > 10000
        assert torch.equal(gp0, gp1) and torch.equal(gq0, gq1) and torch.equal(gi0, gi1)
        assert rel_err(gs1.cpu().numpy(), gs0.cpu().numpy()) <= 1e-5


def test_half_grid_is_chosen_on_the_device_count_without_host_poses():
    """close_views="auto" (the plan's default): the step's forward counts its close views where it sets the views up
    and leaves the count in a pinned host word; a backward issued later reads the word -- no synchronisation, no
    host-side look at the poses -- and takes the half grid when >= 90 % of the views were close.  Before the first
    count has arrived the full grid runs.  Results do not depend on any of it."""
    from sdfest_amd import BatchRenderPlan, Camera
    from sdfest_amd.differentiable_renderer import close_view_fraction
    B, W, H, f = 64, 640, 480, 320.0
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    pos, quat, isc = oracle.random_poses(B, seed=33, width=W, height=H, f=f)
    n_close = int(round(close_view_fraction(pos, isc, cam, 64) * B))
    assert n_close >= 0.9 * B
    far = isc.copy()
    far[::3] *= 3.0
    n_close_far = int(round(close_view_fraction(pos, far, cam, 64) * B))
    assert n_close_far < 0.9 * B
    g = dev(np.random.default_rng(33).uniform(-1, 1, (B, H, W)).astype(np.float32))
    sdf = dev(oracle.blobs_sdf(0))
    for scales, count, expect_half in ((isc, n_close, True), (far, n_close_far, False)):
        pose = (dev(pos), dev(quat), dev(scales))
        ref_plan = BatchRenderPlan(64, B, cam, close_views=False)
        ref_plan.forward(sdf, *pose, 0.005, prepare_backward=True)
        ref = [o.clone() for o in ref_plan.backward(g, sdf, *pose)]
        plan = BatchRenderPlan(64, B, cam)
        assert plan.close_views_seen() == (0, 0)
        for k in range(3):
            plan.forward(sdf, *pose, 0.005, prepare_backward=True)
            if k == 0:
                # nothing has been counted when the first backward is issued... unless the GPU was quicker than the
                # host: either way the result is the same, so only the later steps are pinned down
                pass
            else:
                torch.cuda.synchronize()      # (the test wants a definite state; a real loop never waits)
                assert plan.close_views_seen() == (k + 1, count)
            before = plan.half_grid_steps
            out = plan.backward(g, sdf, *pose)
            if k > 0:
                assert plan.half_grid_steps - before == (1 if expect_half else 0)
            assert torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2]) and torch.equal(out[3], ref[3])
            assert rel_err(out[0].cpu().numpy(), ref[0].cpu().numpy()) <= 1e-5
