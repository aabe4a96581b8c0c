"""GPU: seeded random initialisation networks -- PointNet backbones of 1 ... 5 per-point layers whose widths are no
multiple of the MFMA tile, with and without BatchNorm, dense links and residual links (equal consecutive widths so that
the links are taken, and a last layer twice as wide as its predecessor: the one shape in which a dense link meets a
residual one), point sets of 1 ... 3000 points, heads of 1 ... 3 layers with quaternion or discretised orientation --
against the same layers written out in torch float64 (eval-mode BatchNorm; the sequences of pointnet.py:61-96 and
sdf_pose_network.py:70-115).  The fixed test pins the mug architecture and one plain one on goldens of the imported
reference modules; which epilogue a layer takes depends on these flags and widths."""
import os

import numpy as np
import pytest
import torch

from sdfest_amd.synthetic import init_network_state

pytestmark = pytest.mark.gpu


def torch_pose_net(state, backbone, head, shape_dimension, points):
    """points (M, C) float64 -> (latent, position, scale, orientation) of SDFPoseNet.forward on that one set."""
    F = torch.nn.functional
    st = {k: torch.tensor(v, dtype=torch.float64) for k, v in state.items()}

    def block(x, lin, bn):
        y = F.linear(x, st[lin + ".weight"], st[lin + ".bias"])
        if bn is not None:      # BatchNorm1d in eval mode
            y = (y - st[bn + ".running_mean"]) / torch.sqrt(st[bn + ".running_var"] + 1e-5) * st[bn + ".weight"] \
                + st[bn + ".bias"]
        return F.relu(y)
    n = len(backbone["mlp_out_sizes"])
    out = prev = points
    for i in range(n):
        out = block(out, f"_backbone._linear_layers.{i}", f"_backbone._bn_layers.{i}" if backbone["batchnorm"] else None)
        if backbone["dense"] and i != n - 1:
            out = torch.cat([out, out.max(dim=0, keepdim=True).values.expand(out.shape[0], -1)], dim=1)
        if backbone["residual"] and prev.shape == out.shape:
            out = prev + out
        prev = out
    feat = out.max(dim=0).values
    h = feat[None]
    for i in range(len(head["mlp_out_sizes"])):
        h = block(h, f"_head._linear_layers.{i}", f"_head._bn_layers.{i}" if head["batchnorm"] else None)
    h = F.linear(h, st["_head._final_layer.weight"], st["_head._final_layer.bias"])
    sd = shape_dimension
    ori = h[:, sd + 4:]
    if head["orientation_repr"] == "quaternion":
        ori = ori / torch.sqrt(torch.sum(ori ** 2, 1, keepdim=True))
    return feat, h[:, :sd], h[:, sd:sd + 3], h[:, sd + 3], ori


def draw(seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(1, 6))
    sizes = []
    for i in range(n):
        if i > 0 and rng.uniform() < 0.4:
            sizes.append(sizes[-1])                       # a residual link can be taken
        elif i == n - 1 and i > 0 and rng.uniform() < 0.3:
            sizes.append(2 * sizes[-1])                   # dense + residual into the last layer
        else:
            sizes.append(int(rng.integers(8, 300)) if rng.uniform() < 0.8 else int(rng.integers(300, 1100)))
    backbone = {"in_size": 3, "mlp_out_sizes": sizes, "batchnorm": bool(rng.uniform() < 0.5),
                "dense": bool(rng.uniform() < 0.5), "residual": bool(rng.uniform() < 0.6)}
    hs = [int(rng.integers(16, 300)) for _ in range(int(rng.integers(1, 4)))]
    repr_ = "quaternion" if rng.uniform() < 0.5 else "discretized"
    head = {"in_size": sizes[-1], "mlp_out_sizes": hs, "batchnorm": bool(rng.uniform() < 0.5),
            "orientation_repr": repr_, "orientation_grid_resolution": int(rng.integers(0, 2))}
    sd = int(rng.integers(1, 13))
    cells = 72 * 8 ** head["orientation_grid_resolution"]
    state = init_network_state(100 + seed, backbone, head, sd, cells)
    M = int(rng.choice([1, 5, 64, 777, 3000]))
    pts = (rng.normal(size=(M, 3)) * 0.1).astype(np.float32)
    return backbone, head, sd, state, pts


@pytest.mark.parametrize("seed", range(int(os.environ.get("SDFR_FUZZ_SEEDS", "16"))))
def test_random_network(seed):
    from sdfest_amd.init_network import SDFPoseNet
    backbone, head, sd, state, pts = draw(seed)
    name = f"seed {seed}: {backbone} head={head} shape_dimension={sd} M={len(pts)}"
    net = SDFPoseNet(backbone, head, sd, state)
    feat_ref, lat_ref, pos_ref, sc_ref, ori_ref = torch_pose_net(state, backbone, head, sd,
                                                                 torch.tensor(pts, dtype=torch.float64))
    p = torch.tensor(pts, device="cuda")
    feat = net.features(p).cpu().numpy()
    fr = feat_ref.numpy()
    assert feat.shape == fr.shape, name
    assert np.max(np.abs(feat - fr)) <= 1e-4 * max(np.abs(fr).max(), 1e-6), name
    latent, position, scale, orientation = net(p[None])
    got = torch.cat([latent[0], position[0], scale, orientation[0]]).cpu().numpy()
    ref = torch.cat([lat_ref[0], pos_ref[0], sc_ref, ori_ref[0]]).numpy()
    assert got.shape == ref.shape, name
    assert np.max(np.abs(got - ref)) <= 1e-4 * max(np.abs(ref).max(), 1e-6), name
