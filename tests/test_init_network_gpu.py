"""GPU: forward of the initialisation network against goldens of the reference's torch
modules on seeded random weights (tools/make_goldens.py::make_init_network -> tests/golden/init_network.npz):
the imported VanillaPointNet backbone (dense + residual mug architecture, and a plain residual one), the
head's layers, softmax / prior adjustment / argmax, and SDFPipeline._nn_init's frame handling."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from sdfest_amd.synthetic import MUG_INIT_BACKBONE, MUG_INIT_HEAD, init_network_state

pytestmark = pytest.mark.gpu

PLAIN_BACKBONE = {"in_size": 3, "mlp_out_sizes": [64, 64, 200], "batchnorm": False, "dense": False, "residual": True}
PLAIN_HEAD = {"in_size": 200, "mlp_out_sizes": [96], "batchnorm": False, "orientation_repr": "quaternion"}


@pytest.fixture(scope="module")
def gold():
    d = np.load(os.path.join(GOLDEN, "init_network.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="module")
def mug_net():
    from sdfest_amd.init_network import SDFPoseNet
    return SDFPoseNet(MUG_INIT_BACKBONE, MUG_INIT_HEAD, 8, init_network_state(7))


def test_backbone_and_head_match_torch(gold, mug_net):
    from sdfest_amd.init_network import SDFPoseNet
    plain = SDFPoseNet(PLAIN_BACKBONE, PLAIN_HEAD, 8, init_network_state(8, PLAIN_BACKBONE, PLAIN_HEAD))
    for tag, net in (("mug0", mug_net), ("mug1", mug_net), ("plain0", plain)):
        pts = torch.tensor(gold[f"{tag}_points"], device="cuda")
        feat = net.features(pts)
        ref = gold[f"{tag}_feature"]
        assert feat.shape == ref.shape
        assert np.max(np.abs(feat.cpu().numpy() - ref)) < 1e-4 * np.abs(ref).max(), tag     # 1e-4 relative (north star)
        latent, position, scale, orientation = net(pts[None])
        head = gold[f"{tag}_head"]
        got = torch.cat([latent[0], position[0], scale, orientation[0]]).cpu().numpy()
        if net.orientation_repr == "quaternion":
            q = head[12:] / np.sqrt(np.sum(head[12:] ** 2))                                # sdf_pose_network.py:97-101
            head = np.concatenate([head[:12], q])
            assert abs(np.linalg.norm(got[12:]) - 1.0) < 1e-6
        assert np.max(np.abs(got - head)) < 1e-4 * np.abs(head).max(), tag
        assert latent.shape == (1, 8) and position.shape == (1, 3) and scale.shape == (1,)
    # the set feature does not depend on the order of the points, nor on the padding of the last point tile
    perm = torch.randperm(777, generator=torch.Generator().manual_seed(0))
    pts = torch.tensor(gold["mug0_points"], device="cuda")
    assert torch.equal(mug_net.features(pts[perm.cuda()]), mug_net.features(pts))


def test_orientation_posterior(gold, mug_net):
    logits = torch.tensor(gold["mug0_head"][12:], device="cuda")
    post, idx, mx = mug_net.orientation_posterior(logits)
    assert np.allclose(post.cpu().numpy(), gold["post"], rtol=1e-5, atol=1e-9)
    assert int(idx) == int(np.argmax(gold["post"])) and abs(float(mx) - gold["post"].max()) < 1e-6
    post2, idx2, mx2 = mug_net.orientation_posterior(logits, torch.tensor(gold["prior"]), torch.tensor(gold["train_prior"]))
    assert np.allclose(post2.cpu().numpy(), gold["post_adjusted"], rtol=2e-5, atol=1e-9)
    assert int(idx2) == int(np.argmax(gold["post_adjusted"])) and abs(float(post2.sum()) - 1.0) < 1e-5
    # the reference suite's known answers (tests/estimation/test_simple_setup.py:6-26), through the kernel:
    # softmax of log p is p
    p = torch.tensor([0.8, 0.2, 1e-30, 1e-30])
    train = torch.tensor([0.4, 0.4, 0.1, 0.1])
    same, _, _ = mug_net.orientation_posterior(torch.log(p), torch.tensor([0.25] * 4), train)
    exp = np.array([0.8 * 0.25 / 0.4, 0.2 * 0.25 / 0.4, 0, 0]); exp /= exp.sum()
    assert np.allclose(same.cpu().numpy(), exp, atol=1e-6)
    changed, i3, _ = mug_net.orientation_posterior(torch.log(p), torch.tensor([0.1, 0.4, 0.25, 0.25]), train)
    exp = np.array([0.8 * 0.1 / 0.4, 0.2 * 0.4 / 0.4, 0.0, 0.0]); exp /= exp.sum()
    assert np.allclose(changed.cpu().numpy(), exp, atol=1e-6) and int(i3) == 0


def test_nn_init_frames_and_strategies(gold, mug_net):
    """simple_setup.py:770-842: centroid handling, grid argmax -> quaternion, camera -> world frame,
    "first" / "best" view selection, mean_shape, NoDepthError."""
    from sdfest_amd import Camera
    from sdfest_amd.init_network import NoDepthError, nn_init
    from sdfest_amd.pipeline import depth_to_pointcloud, quaternion_apply, quaternion_multiply
    W, H = 96, 72
    cam = Camera(W, H, 48.0, 48.0, 48.0, 36.0, pixel_center=0.5)
    g = torch.Generator().manual_seed(4)
    depth = torch.zeros((2, H, W))
    depth[0, 20:50, 30:70] = 0.4 + 0.05 * torch.rand((30, 40), generator=g)
    depth[1, 10:40, 20:50] = 0.5 + 0.05 * torch.rand((30, 30), generator=g)
    depth = depth.cuda()
    cam_pos = torch.tensor([[0.0, 0.0, 0.0], [0.2, -0.1, 0.05]], device="cuda")
    cq = torch.tensor([[0.0, 0.0, 0.0, 1.0], [0.1, 0.3, -0.2, 0.9]], device="cuda")
    cq = cq / cq.norm(dim=1, keepdim=True)
    per_view = []
    for v in range(2):      # the statements of _nn_init, one view at a time
        pts = depth_to_pointcloud(depth[v], cam)
        c = pts.mean(0)
        latent, position, scale, logits = mug_net((pts - c)[None])
        post, idx, mx = mug_net.orientation_posterior(logits)
        q_cam = torch.tensor(mug_net.grid.index_to_quat(int(idx)), dtype=torch.float, device="cuda")[None]
        per_view.append((latent, quaternion_apply(cq[v], position + c) + cam_pos[v], scale,
                         quaternion_multiply(cq[v], q_cam), float(mx)))
    first = nn_init(mug_net, cam, depth, cam_pos, cq, {"init_view": "first"})
    for a, b in zip(first, per_view[0][:4]):
        assert torch.equal(a, b)
    best = nn_init(mug_net, cam, depth, cam_pos, cq, {"init_view": "best", "mean_shape": True})
    w = int(np.argmax([p[4] for p in per_view]))
    assert torch.equal(best[1], per_view[w][1]) and torch.equal(best[3], per_view[w][3])
    assert torch.count_nonzero(best[0]) == 0 and abs(float(best[3].norm()) - 1.0) < 1e-6
    with pytest.raises(NoDepthError):
        nn_init(mug_net, cam, torch.zeros((1, H, W), device="cuda"), cam_pos[:1], cq[:1], {"init_view": "first"})
    with pytest.raises(NotImplementedError):
        nn_init(mug_net, cam, depth, cam_pos, cq, {"init_view": "median"})


def test_resident_init_matches_the_host_driven_form(gold, mug_net):
    """ResidentInit = _nn_init (simple_setup.py:718-844) as a fixed launch sequence with the point count on the device
    (sdfr_pointnet_layer_counted), the cell's quaternion from an uploaded table and camera -> world + "first" / "best"
    in sdfr_init_estimate: against nn_init (same layers; the centroid is a block-sum tree instead of torch.mean, hence
    rounding, not bits), eager == captured bit for bit, priors, mean_shape, and the empty-cloud report."""
    from sdfest_amd import Camera
    from sdfest_amd.init_network import ResidentInit, nn_init
    W, H = 96, 72
    cam = Camera(W, H, 48.0, 48.0, 48.0, 36.0, pixel_center=0.5)
    g = torch.Generator().manual_seed(4)
    depth = torch.zeros((3, H, W))
    depth[0, 20:50, 30:70] = 0.4 + 0.05 * torch.rand((30, 40), generator=g)
    depth[1, 10:40, 20:50] = 0.5 + 0.05 * torch.rand((30, 30), generator=g)
    depth[2, 5:66, 3:90] = 0.3 + 0.1 * torch.rand((61, 87), generator=g)       # > 4096 points: several row blocks
    depth = depth.cuda().contiguous()
    cam_pos = torch.tensor([[0.0, 0.0, 0.0], [0.2, -0.1, 0.05], [-0.1, 0.1, 0.0]], device="cuda")
    cq = torch.tensor([[0.0, 0.0, 0.0, 1.0], [0.1, 0.3, -0.2, 0.9], [-0.2, 0.1, 0.1, 0.95]], device="cuda")
    cq = (cq / cq.norm(dim=1, keepdim=True)).contiguous()
    C = mug_net.grid.num_cells()
    prior = torch.rand((3, C), generator=g).cuda()
    train = (0.5 + torch.rand(C, generator=g)).cuda()
    for cfg, kw in (({"init_view": "first"}, {}), ({"init_view": "best"}, {}), ({"init_view": "best", "mean_shape": True}, {}),
                    ({"init_view": "best"}, dict(prior_orientation_distribution=prior)),
                    ({"init_view": "first"}, dict(prior_orientation_distribution=prior,
                                                  training_orientation_distribution=train))):
        for normalize in (True, False):
            ri = ResidentInit(mug_net, cam, 3, cfg, normalize_pose=normalize)
            ref = nn_init(mug_net, cam, depth, cam_pos, cq, cfg, normalize_pose=normalize, **kw)
            eager = [t.clone() for t in ri(depth, cam_pos, cq, kw.get("prior_orientation_distribution"),
                                           kw.get("training_orientation_distribution"), use_graph=False)]
            assert ri.empty_views() == []
            for rep in range(2):          # captured, then replayed
                got = [t.clone() for t in ri(depth, cam_pos, cq, kw.get("prior_orientation_distribution"),
                                             kw.get("training_orientation_distribution"), use_graph=True)]
                for a, b in zip(got, eager):
                    assert torch.equal(a, b), (cfg, normalize, rep)
            assert len(ri._graphs) == 1
            for name, a, b in zip(("latent", "position", "scale", "orientation"), eager, ref):
                assert a.shape == b.shape, (name, a.shape, b.shape)
                tol = 1e-4 * max(1.0, float(b.abs().max()))
                assert (a - b).abs().max().item() <= tol, (cfg, normalize, name, a, b)
            assert torch.equal(eager[3], ref[3])                      # the same orientation cell, the same product
    # an image without a point: reported after the launches (the front door raises NoDepthError, :780-781)
    ri = ResidentInit(mug_net, cam, 3, {"init_view": "best"})
    empty = depth.clone(); empty[1] = 0
    ri(empty, cam_pos, cq, use_graph=False)
    assert ri.empty_views() == [1]
    ri = ResidentInit(mug_net, cam, 3, {"init_view": "first"})
    ri(empty, cam_pos, cq)
    assert ri.empty_views() == []                                      # "first" never looks at view 1


def test_nan_points_reach_the_outputs(gold, mug_net):
    """torch's relu and max propagate NaN (pointnet.py:64-96): a NaN coordinate (or weight) must come out as NaN, not
    as a plausible pose computed from `fmaxf(NaN, 0) = 0` (round-2 advisor finding)."""
    pts = torch.tensor(gold["mug0_points"], device="cuda").clone()
    ok = mug_net.features(pts)
    assert torch.isfinite(ok).all()
    pts[5, 1] = float("nan")
    feat = mug_net.features(pts)
    assert torch.isnan(feat).any()
    latent, position, scale, orientation = mug_net(pts[None])
    assert torch.isnan(latent).any() or torch.isnan(position).any() or torch.isnan(orientation).any()
