"""Single-shot initialisation network, inference only: host-side mirror of
``sdfest/initialization/pointnet.py::VanillaPointNet`` (:7-96),
``sdfest/initialization/sdf_pose_network.py::SDFPoseHead`` / ``SDFPoseNet`` (:9-147) and of
``SDFPipeline._nn_init`` (estimation/simple_setup.py:718-844), which runs it once before the
render-and-compare loop.  The arithmetic runs in ``libsdfr_hip.so`` (initnet.hip): per-point layers
as fp32 MFMA GEMMs with bias / BatchNorm / ReLU / residual / set-maximum epilogues, the head as
wave-per-row products, softmax + prior adjustment + argmax in one kernel.

Weights come as the state dict of the reference's ``SDFPoseNet`` (keys ``_backbone._linear_layers.i.*``,
``_backbone._bn_layers.i.*``, ``_head._linear_layers.i.*``, ``_head._bn_layers.i.*``,
``_head._final_layer.*``).  BatchNorm is applied in inference mode (running statistics), folded to a
scale and a shift per channel.  The trained weights of the paper are not in the reference repository
(download URLs only, mug.yaml:115-116); tests use seeded random weights of the mug architecture.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .differentiable_renderer import Camera
from .pipeline import depth_to_pointcloud, quaternion_apply, quaternion_multiply
from .so3grid import SO3Grid

_BN_EPS = 1e-5  # torch.nn.BatchNorm1d default


class NoDepthError(Exception):
    """Raised when a depth image holds no valid pixel (simple_setup.py:30-32, :780-781)."""


def adjust_categorical_posterior(posterior: torch.Tensor, prior: torch.Tensor,
                                 train_prior: Optional[torch.Tensor]) -> torch.Tensor:
    """simple_setup.py:978-1009: posterior * prior / train_prior, L1-normalised along the last axis."""
    adjusted = posterior.clone() * prior
    if train_prior is not None:
        adjusted = adjusted / train_prior
    return torch.nn.functional.normalize(adjusted, p=1, dim=-1)


def _f32(t, dev):
    return torch.as_tensor(np.asarray(t) if not torch.is_tensor(t) else t).to(device=dev, dtype=torch.float32).contiguous()


class _Layer:
    def __init__(self, state: Dict, prefix_lin: str, prefix_bn: Optional[str], dev):
        self.w = _f32(state[prefix_lin + ".weight"], dev)
        self.b = _f32(state[prefix_lin + ".bias"], dev)
        self.cout, self.cin_total = self.w.shape
        if prefix_bn is not None:
            g, beta = _f32(state[prefix_bn + ".weight"], dev), _f32(state[prefix_bn + ".bias"], dev)
            mu, var = _f32(state[prefix_bn + ".running_mean"], dev), _f32(state[prefix_bn + ".running_var"], dev)
            self.scale = (g / torch.sqrt(var + _BN_EPS)).contiguous()
            self.shift = (beta - mu * self.scale).contiguous()
        else:
            self.scale = torch.ones(self.cout, device=dev)
            self.shift = torch.zeros(self.cout, device=dev)


class SDFPoseNet:
    """``backbone`` / ``head``: the reference's config dictionaries (mug.yaml:98-111)."""

    def __init__(self, backbone: Dict, head: Dict, shape_dimension: int, state_dict: Dict, device="cuda"):
        self.dev = torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.L = _lib.lib()
        self.in_size = int(backbone["in_size"])
        self.pn_sizes = list(backbone["mlp_out_sizes"])
        self.dense, self.residual = bool(backbone.get("dense", False)), bool(backbone.get("residual", False))
        bn_b = bool(backbone["batchnorm"])
        self.pn = [_Layer(state_dict, f"_backbone._linear_layers.{i}", f"_backbone._bn_layers.{i}" if bn_b else None,
                          self.dev) for i in range(len(self.pn_sizes))]
        bn_h = bool(head["batchnorm"])
        self.head = [_Layer(state_dict, f"_head._linear_layers.{i}", f"_head._bn_layers.{i}" if bn_h else None, self.dev)
                     for i in range(len(head["mlp_out_sizes"]))]
        self.final = _Layer(state_dict, "_head._final_layer", None, self.dev)
        self.shape_dimension = int(shape_dimension)
        self.orientation_repr = head.get("orientation_repr", "quaternion")
        if self.orientation_repr == "discretized":
            self.grid = SO3Grid(head["orientation_grid_resolution"])
            n_out = self.shape_dimension + 4 + self.grid.num_cells()
        elif self.orientation_repr == "quaternion":
            self.grid, n_out = None, self.shape_dimension + 8
        else:
            raise NotImplementedError(f"orientation_repr {self.orientation_repr} is not supported.")
        if self.final.cout != n_out or self.head[0].cin_total != self.pn_sizes[-1]:
            raise RuntimeError("state dict does not match the head configuration")

    def _st(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _vec(self, layer: _Layer, koff: int, x: Optional[torch.Tensor], k: int, bn: bool, relu: bool) -> torch.Tensor:
        y = torch.empty(layer.cout, dtype=torch.float32, device=self.dev)
        rc = self.L.sdfr_linear_vec(layer.w.data_ptr(), layer.cin_total, koff, x.data_ptr() if x is not None else None, k,
                                    layer.b.data_ptr(), layer.scale.data_ptr() if bn else None,
                                    layer.shift.data_ptr() if bn else None, int(relu), y.data_ptr(), layer.cout,
                                    self.dev.index, self._st())
        _lib.check(rc, "sdfr_linear_vec")
        return y

    def features(self, points: torch.Tensor) -> torch.Tensor:
        """VanillaPointNet.forward on ONE set (M, in_size) -> (C_last,) (pointnet.py:61-96)."""
        x = points.to(device=self.dev, dtype=torch.float32).contiguous()
        if x.dim() != 2 or x.shape[1] != self.in_size or x.shape[0] < 1:
            raise RuntimeError(f"points must have shape (M >= 1, {self.in_size})")
        M = x.shape[0]
        F, F_width = x, self.in_size          # the per-point part of `prev_out`
        G = None                              # the broadcast part of `prev_out` (dense links)
        prev_width = self.in_size             # width of prev_out including a concatenated maximum
        n = len(self.pn)
        for i, layer in enumerate(self.pn):
            last = i == n - 1
            if layer.cin_total != F_width + (G.numel() if G is not None else 0):
                raise RuntimeError("state dict does not match the backbone configuration")
            # bias of the layer, plus the constant the concatenated maximum contributes to every point
            cvec = self._vec(layer, F_width, G, G.numel(), False, False) if G is not None else layer.b
            out_width = layer.cout * (2 if (self.dense and not last) else 1)
            use_res = self.residual and prev_width == out_width     # `prev_out.shape == out.shape` (:88-90)
            colmax = torch.empty(layer.cout, dtype=torch.float32, device=self.dev)
            store = not last or use_res
            Y = torch.empty((M, layer.cout), dtype=torch.float32, device=self.dev) if store else None
            res_F = None
            if use_res and F_width == layer.cout:
                res_F = F
            elif use_res:
                # prev_out = [F | G] (a dense link) into a last layer as wide as both halves together
                res_F = torch.cat([F, G.reshape(1, -1).expand(M, -1)], dim=1).contiguous()
            rc = self.L.sdfr_pointnet_layer(F.data_ptr(), M, F_width, F.shape[1], layer.w.data_ptr(), layer.cin_total,
                                            cvec.data_ptr(), layer.scale.data_ptr(), layer.shift.data_ptr(),
                                            res_F.data_ptr() if res_F is not None else None,
                                            Y.data_ptr() if Y is not None else None, layer.cout, layer.cout,
                                            colmax.data_ptr(), self.dev.index, self._st())
            _lib.check(rc, "sdfr_pointnet_layer")
            if self.dense and not last:
                # out = cat(out, max(out)); a residual adds prev_out = [F | G] to both halves
                G = colmax + G if (use_res and G is not None) else colmax
            else:
                G = None
            if last:
                # torch.max over the points of the final `out` (after a residual, if any)
                return colmax if not use_res else Y.max(dim=0).values
            F, F_width, prev_width = Y, layer.cout, out_width
        raise AssertionError

    def head_forward(self, feature: torch.Tensor):
        """SDFPoseHead.forward on one feature vector (sdf_pose_network.py:70-115): (latent (1,L), position
        (1,3), scale (1,), orientation (1,4) normalised quaternion or (1,C) logits)."""
        out = feature
        for layer in self.head:
            out = self._vec(layer, 0, out, layer.cin_total, True, True)
        out = self._vec(self.final, 0, out, self.final.cin_total, False, False)[None]
        sd = self.shape_dimension
        orientation = out[:, sd + 4:]
        if self.orientation_repr == "quaternion":
            orientation = orientation / torch.sqrt(torch.sum(orientation ** 2, 1, keepdim=True))
        return out[:, 0:sd], out[:, sd:sd + 3], out[:, sd + 3], orientation

    def __call__(self, points: torch.Tensor):
        """SDFPoseNet.forward for a batch of one point set: (1,M,3) or (M,3)."""
        if points.dim() == 3:
            if points.shape[0] != 1:
                raise NotImplementedError("one point set per call (the estimator's use, simple_setup.py:786-787)")
            points = points[0]
        return self.head_forward(self.features(points))

    def orientation_posterior(self, logits: torch.Tensor, prior: Optional[torch.Tensor] = None,
                              train_prior: Optional[torch.Tensor] = None):
        """softmax, optional prior adjustment, argmax (simple_setup.py:795-812): (posterior (C,), index, max)."""
        lg = logits.reshape(-1).to(device=self.dev, dtype=torch.float32).contiguous()
        C = lg.numel()
        post = torch.empty(C, dtype=torch.float32, device=self.dev)
        idx = torch.empty(1, dtype=torch.int32, device=self.dev)
        mx = torch.empty(1, dtype=torch.float32, device=self.dev)
        pr = prior.reshape(-1).to(device=self.dev, dtype=torch.float32).contiguous() if prior is not None else None
        tp = (train_prior.reshape(-1).to(device=self.dev, dtype=torch.float32).contiguous()
              if (train_prior is not None and prior is not None) else None)
        rc = self.L.sdfr_orientation_posterior(lg.data_ptr(), C, pr.data_ptr() if pr is not None else None,
                                               tp.data_ptr() if tp is not None else None, post.data_ptr(),
                                               idx.data_ptr(), mx.data_ptr(), self.dev.index, self._st())
        _lib.check(rc, "sdfr_orientation_posterior")
        return post, idx, mx


def nn_init(network: SDFPoseNet, camera: Camera, depth_images: torch.Tensor, camera_positions: torch.Tensor,
            camera_orientations: torch.Tensor, config: Dict, normalize_pose: bool = True,
            prior_orientation_distribution: Optional[torch.Tensor] = None,
            training_orientation_distribution: Optional[torch.Tensor] = None):
    """``SDFPipeline._nn_init`` (simple_setup.py:718-844) for the VanillaPointNet backbone: per view
    depth -> point cloud (-> centroid removed) -> network -> estimate in the camera frame -> world frame;
    ``config["init_view"]`` "first" | "best" (largest posterior maximum, discretised orientations only);
    ``config["mean_shape"]`` zeroes the latent.  Returns (latent (1,L), position (1,3), scale (1,),
    orientation (1,4))."""
    if prior_orientation_distribution is not None and network.orientation_repr != "discretized":
        raise ValueError("prior_orientation_distribution only supported for discretized orientation representation.")
    best, best_result = 0, None
    for i, (depth_image, camera_orientation, camera_position) in enumerate(
            zip(depth_images, camera_orientations, camera_positions)):
        inp = depth_to_pointcloud(depth_image, camera)
        if len(inp) == 0:
            raise NoDepthError
        centroid = None
        if normalize_pose:
            centroid = torch.mean(inp, dim=-2)
            inp = inp - centroid
        latent_shape, position, scale, orientation_repr = network(inp.unsqueeze(0))
        if config.get("mean_shape", False):
            latent_shape = latent_shape.new_zeros(latent_shape.shape)
        if centroid is not None:
            position = position + centroid
        if network.orientation_repr == "discretized":
            prior = prior_orientation_distribution[i] if prior_orientation_distribution is not None else None
            _, index, maximum = network.orientation_posterior(orientation_repr, prior, training_orientation_distribution)
            orientation_camera = torch.tensor(network.grid.index_to_quat(int(index.item())), dtype=torch.float,
                                              device=network.dev).unsqueeze(0)
        else:
            orientation_camera, maximum = orientation_repr, None
        position_world = quaternion_apply(camera_orientation, position) + camera_position
        orientation_world = quaternion_multiply(camera_orientation, orientation_camera)
        strategy = config.get("init_view", "first")
        if strategy == "first":
            return latent_shape, position_world, scale, orientation_world
        elif strategy == "best":
            if network.orientation_repr != "discretized":
                raise NotImplementedError('"best" init strategy only supported with discretized '
                                          "orientation representation")
            if maximum.item() > best:
                best, best_result = maximum.item(), (latent_shape, position_world, scale, orientation_world)
        else:
            raise NotImplementedError('Only "first" and "best" strategies are currently supported')
    return best_result
