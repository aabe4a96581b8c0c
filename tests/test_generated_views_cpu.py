"""CPU: the host-side pieces of the batched view generator (sdfest_amd/generated_views.py) against
per-sample restatements of generated_dataset.py:187-207, :262-271, :296-308, :366-373."""
import math

import numpy as np
import torch

from sdfest_amd import Camera
from sdfest_amd import generated_views as gv
from sdfest_amd.pipeline import depth_to_pointcloud


def test_pose_sampler_distributions():
    cam = Camera(640, 480, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
    g = torch.Generator().manual_seed(0)
    p, q, s = gv.sample_poses(20000, cam, 0.3, 1.2, 0.2, 0.02, g)
    z = -p[:, 2]
    assert z.min() >= 0.3 and z.max() <= 1.2 and abs(z.mean().item() - 0.75) < 0.01
    x_pix, y_pix = p[:, 0] / z * cam.fx, p[:, 1] / z * cam.fy
    # the reference's own bounds: x in [-W/2, H/2], y in [-H/2, H/2]  (generated_dataset.py:263-266)
    assert x_pix.min() >= -320.01 and x_pix.max() <= 240.01 and x_pix.max() > 235 and x_pix.min() < -315
    assert y_pix.min() >= -240.01 and y_pix.max() <= 240.01
    assert torch.allclose(q.norm(dim=1), torch.ones(20000), atol=1e-6)
    assert q.mean(0).abs().max() < 0.02 and abs((q ** 2).mean().item() - 0.25) < 0.01   # uniform on S^3
    assert abs(s.mean().item() - 0.1) < 5e-4 and abs(s.std().item() - 0.01) < 5e-4       # extent / 2
    g2 = torch.Generator().manual_seed(0)
    p2, q2, s2 = gv.sample_poses(20000, cam, 0.3, 1.2, 0.2, 0.02, g2)
    assert torch.equal(p, p2) and torch.equal(q, q2) and torch.equal(s, s2)


def test_gaussian_kernel_and_smoothing_follow_the_reference_steps():
    k = gv.gaussian_kernel(1.0, 5)
    assert k.shape == (1, 1, 5, 5) and abs(k.sum().item() - 1.0) < 1e-6
    assert torch.allclose(k, k.flip(-1)) and torch.allclose(k, k.transpose(-1, -2))
    rng = np.random.default_rng(0)
    depth = torch.tensor(rng.uniform(0.5, 1.0, (3, 24, 32)).astype(np.float32))
    depth[:, :6] = 0
    depth[1, 12:15, 10:13] = 0
    ref = depth.clone()
    out = gv.smooth_depth(depth.clone(), k, torch.tensor([True, True, False]))
    for b in range(2):   # generated_dataset.py:296-308, one sample at a time
        d = ref[b].clone()
        d[d == 0] = torch.nan
        f = torch.nn.functional.conv2d(d[None, None], k, padding="same")[0, 0]
        m = torch.logical_or(f.isnan(), f.isinf())
        d[~m] = f[~m]
        d[d.isnan()] = 0.0
        assert torch.allclose(out[b], d, rtol=0, atol=2e-6) and torch.equal(out[b] == 0, d == 0)
        assert (out[b] != ref[b]).sum() > 50 and torch.equal(out[b][:6], ref[b][:6])
    assert torch.equal(out[2], ref[2])


def test_batched_back_projection_equals_per_view():
    cam = Camera(32, 24, 30.0, 31.0, 15.5, 12.25, pixel_center=0.5)
    rng = np.random.default_rng(1)
    depth = torch.tensor(rng.uniform(0.5, 1.0, (4, 24, 32)).astype(np.float32))
    depth[depth < 0.7] = 0
    depth[2] = 0
    pts, counts = gv.depth_to_pointsets(depth, cam)
    parts = torch.split(pts, counts.tolist())
    for b in range(4):
        assert torch.equal(parts[b], depth_to_pointcloud(depth[b], cam))
    assert counts[2] == 0


def test_unsupported_options_raise():
    import pytest
    base = {"z_min": 0.3, "z_max": 1.0, "extent_mean": 0.2, "extent_std": 0.02}
    with pytest.raises(NotImplementedError):    # generated_dataset.py:361-364
        gv.SDFVAEViewGenerator({**base, "orientation_repr": "euler"}, None)
    with pytest.raises(KeyError):
        gv.SDFVAEViewGenerator({"z_min": 0.3}, None)
    with pytest.raises(ValueError):
        gv.gaussian_kernel(1.0, 4)


def test_mask_affine_parameters_and_matrices():
    """RandomAffine.get_params for degrees=(0,1), translate=(0,0.01), scale=(0.999,1.001) at 640x480 and
    the inverse matrix about the image centre (torchvision functional._get_inverse_affine_matrix)."""
    import math
    g = torch.Generator().manual_seed(0)
    angle, translate, scale = gv.sample_mask_affine(4000, 640, 480, g)
    assert angle.min() >= 0 and angle.max() <= 1 and abs(angle.mean().item() - 0.5) < 0.03
    assert torch.all(translate[:, 0] == 0)                           # 0.00 * width
    assert torch.all(translate[:, 1] == translate[:, 1].round()) and translate[:, 1].abs().max() == 5   # round(U(-4.8, 4.8))
    assert scale.min() >= 0.999 and scale.max() <= 1.001
    m = gv.inverse_affine_matrices(torch.tensor([0.0, 90.0, 30.0]), torch.tensor([[3.0, -2.0], [0.0, 0.0], [1.0, 4.0]]),
                                   torch.tensor([1.0, 2.0, 0.5])).double()
    assert torch.allclose(m[0], torch.tensor([1.0, 0, -3, 0, 1, 2]).double())          # inverse of a shift
    assert torch.allclose(m[1], torch.tensor([0.0, 0.5, 0, -0.5, 0, 0]).double(), atol=1e-7)
    # the forward map (rotate by +angle: [[c, -s], [s, c]], scale, then shift) composed with it is the identity
    c, s_ = math.cos(math.radians(30.0)), math.sin(math.radians(30.0))
    fwd = torch.tensor([[0.5 * c, -0.5 * s_, 1.0], [0.5 * s_, 0.5 * c, 4.0], [0, 0, 1]]).double()
    inv = torch.cat([m[2].reshape(2, 3), torch.tensor([[0.0, 0, 1]]).double()])
    assert torch.allclose(inv @ fwd, torch.eye(3).double(), atol=1e-6)
    # two parameter tuples of the dataset's range worked out by hand from torchvision's documented formula
    # (functional._get_inverse_affine_matrix, center (0, 0), no shear: M^-1 = [[cos, sin], [-sin, cos]] / scale, then
    # M^-1 (-t) in the last column):  (0.5 deg, t = (0, 3), s = 1.0005)  and  (0.75 deg, t = (0, -5), s = 0.9992)
    k = gv.inverse_affine_matrices(torch.tensor([0.5, 0.75]), torch.tensor([[0.0, 3.0], [0.0, -5.0]]),
                                   torch.tensor([1.0005, 0.9992])).double()
    assert torch.allclose(k[0], torch.tensor([0.999462192, 0.0087221744, -0.0261665232, -0.0087221744, 0.999462192,
                                              -2.9983865759]).double(), atol=2e-7)
    assert torch.allclose(k[1], torch.tensor([1.0007148995, 0.0131000756, 0.0655003782, -0.0131000756, 1.0007148995,
                                              5.0035744975]).double(), atol=2e-7)
