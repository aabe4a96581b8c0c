"""GPU: K estimates optimised side by side (pipeline.MultiObjectRenderAndCompare, SDFPipeline.estimate_objects) -- the
K detected objects of one frame in ONE launch sequence per iteration -- against what the reference does with them: one
pipeline call per object, one after the other (simple_setup.py:213-225).  Same arithmetic per object, so every row must
follow the single-object loop's trajectory."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu

T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda")


@pytest.fixture(scope="module")
def frame():
    """one 320x240 frame with 5 objects (different shapes, poses, sizes) seen from one camera, their instance masks,
    and a perturbed initial estimate per object"""
    import _loop_scenes as S
    from sdfest_amd import Camera, render_depth_gpu
    dec, d = S.mug_decoder()
    W, H, f = 320, 240, 300.0
    cam = Camera(W, H, f, f, W / 2, H / 2, pixel_center=0.5)
    rng = np.random.default_rng(12)
    K = 5
    objs = []
    for k in range(K):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        objs.append(dict(z=d["z"][9 + k % 3] * (0.3 + 0.1 * k), p=np.array([-0.12 + 0.06 * k, 0.03 * (k % 2) - 0.02, -0.5 - 0.03 * k]),
                         q=q, s=0.05 + 0.004 * k))
    images = []
    with torch.no_grad():
        for o in objs:
            sdf = dec.decode(T(o["z"][None]))[0, 0]
            images.append(render_depth_gpu(sdf, T(o["p"]), T(o["q"]), T(1.0 / o["s"]), None, None, None, 0.005, cam))
    images = torch.stack(images)
    assert (images > 0).sum(dim=(1, 2)).min() > 300
    # the frame: the nearest surface per pixel; the masks: which object it belongs to
    big = torch.where(images > 0, images, torch.full_like(images, 1e9))
    nearest = big.argmin(dim=0)
    depth = big.min(dim=0).values
    depth = torch.where(depth < 1e8, depth, torch.zeros_like(depth))
    masks = torch.stack([(nearest == k) & (depth > 0) for k in range(K)])
    assert masks.sum(dim=(1, 2)).min() > 200
    init = []
    for o in objs:
        q0 = o["q"] + rng.normal(0, 0.03, 4)
        init.append((o["p"] + rng.normal(0, 0.006, 3), q0 / np.linalg.norm(q0), o["s"] * 1.07, np.zeros(8)))
    return dec, cam, depth.contiguous(), masks.contiguous(), init, objs


@pytest.mark.parametrize("shape_opt", [False, True])
@pytest.mark.parametrize("use_graph", [False, True])
def test_rows_follow_the_single_object_loop(frame, shape_opt, use_graph):
    from sdfest_amd.pipeline import FusedRenderAndCompare, MultiObjectRenderAndCompare, preprocess_depth
    dec, cam, depth, masks, init, _ = frame
    K = masks.shape[0]
    cfg = {"threshold": 0.005, "max_iterations": 7, "depth_weight": 1.0, "pc_weight": 3.0}
    frames = depth[None].expand(K, -1, -1).contiguous()
    preprocess_depth(frames, masks, 2.0)
    p0 = T(np.stack([i[0] for i in init])); q0 = T(np.stack([i[1] for i in init]))
    s0 = T([i[2] for i in init]); z0 = T(np.stack([i[3] for i in init]))
    multi = MultiObjectRenderAndCompare(dec, cam, cfg, K, shape_optimization=shape_opt, graph_iterations=3)
    multi.rebind(frames)
    outs = []
    for rep in range(2):                       # the second run replays what the first captured
        hist = [] if rep == 0 else None
        outs.append([t.clone() for t in multi(p0, q0, s0, z0, use_graph=use_graph, history=hist)])
        torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b) or (a - b).abs().max().item() < 5e-6       # (history: single-iteration replays)
    assert multi.step.tolist() == [7] * K and multi.counts.tolist() == masks.sum(dim=(1, 2)).tolist()
    moved = 0.0
    for k in range(K):
        # (the single loop in its two-launch form: the K-object loop's kernels.  Its default for one view, the render
        # pair as ONE launch, applies weight / count to sums instead of terms -- equal to rounding, asserted below)
        single = FusedRenderAndCompare(dec, cam, cfg, frames[k:k + 1].contiguous(), shape_optimization=shape_opt,
                                       fused_render=False)
        ref = single(p0[k:k + 1], q0[k:k + 1], s0[k:k + 1], z0[k:k + 1], use_graph=False)
        if k == 0:
            one = FusedRenderAndCompare(dec, cam, cfg, frames[k:k + 1].contiguous(), shape_optimization=shape_opt)
            assert one.fused_render
            for a, b, lr in zip(one(p0[k:k + 1], q0[k:k + 1], s0[k:k + 1], z0[k:k + 1], use_graph=False), ref,
                                (1e-3, 1e-2, 1e-3, 1e-2)):
                assert (a - b).abs().max().item() <= 0.01 * lr * 7
        got = [outs[0][0][k], outs[0][1][k], outs[0][2][k], outs[0][3][k]]
        # Adam's steps are ~lr (1e-3 position / scale, 1e-2 orientation / latent): 1 % of a step per iteration
        for name, a, b, lr in zip(("position", "orientation", "scale", "latent"), got, ref, (1e-3, 1e-2, 1e-3, 1e-2)):
            err = (a.reshape(-1) - b.reshape(-1)).abs().max().item()
            assert err <= 0.01 * lr * 7, (k, name, err, a, b)
            if not shape_opt:     # pose only: no float atomics anywhere -- the same sums in the same order, bit for bit
                assert torch.equal(a.reshape(-1), b.reshape(-1)), (k, name, err)
        moved = max(moved, (got[0] - p0[k]).abs().max().item())
        if shape_opt:
            assert got[3].abs().max().item() > 1e-3
        else:
            assert torch.equal(got[3], z0[k])
    assert moved > 1e-3
    # a new frame (the objects in another order) on the same buffers and graphs
    graph = multi.graph
    perm = torch.tensor([2, 0, 4, 1, 3], device="cuda")
    multi.rebind(frames[perm].contiguous())
    out2 = multi(p0[perm], q0[perm], s0[perm], z0[perm], use_graph=use_graph)
    assert multi.graph is graph
    for a, b, lr in zip(out2, outs[0], (1e-3, 1e-2, 1e-3, 1e-2)):
        assert (a - b[perm]).abs().max().item() <= 0.01 * lr * 7


def test_front_door_estimate_objects(frame):
    """SDFPipeline.estimate_objects(depth, masks) = K calls of pipeline(depth, masks[k], color), side by side"""
    from sdfest_amd import NoDepthError, SDFPipeline
    from test_sdfpipeline_gpu import make_config, mug_weights, plausible_init_state
    dec, cam, depth, masks, init, objs = frame
    K = masks.shape[0]
    cfg = make_config(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, 0.005, 6, far_field=2.0)
    pipe = SDFPipeline(cfg, vae_state_dict=mug_weights(), init_state_dict=plausible_init_state())
    keep = depth.clone()
    out = pipe.estimate_objects(depth, masks)
    assert torch.equal(depth, keep)                                   # the frame is not modified
    assert [tuple(t.shape) for t in out] == [(K, 3), (K, 4), (K,), (K, 8)]
    color = torch.zeros(depth.shape + (3,), device="cuda")
    for k in range(K):
        ref = pipe(depth.clone(), masks[k], color)
        for name, a, b, lr in zip(("position", "orientation", "scale", "latent"),
                                  (out[0][k], out[1][k], out[2][k], out[3][k]), ref, (1e-3, 1e-2, 1e-3, 1e-2)):
            err = (a.reshape(-1) - b.reshape(-1)).abs().max().item()
            assert err <= 0.01 * lr * 6, (k, name, err)
        assert (out[0][k] - T(objs[k]["p"])).abs().max().item() < 0.1         # at the object it belongs to (6 iterations
        #                                                                           from whatever cell the seeded network picks)
    # the next frame on the same object: nothing is built or captured again
    loop, graph = pipe._last_multi_loop, pipe._last_multi_loop.graph
    out2 = pipe.estimate_objects(depth, masks)
    assert pipe._last_multi_loop is loop and loop.graph is graph
    for a, b in zip(out, out2):
        assert (a - b).abs().max().item() < 1e-4
    empty = masks.clone(); empty[3] = False
    with pytest.raises(NoDepthError):
        pipe.estimate_objects(depth, empty)
