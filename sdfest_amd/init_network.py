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


class ResidentInit:
    """``SDFPipeline._nn_init`` (simple_setup.py:718-844) as a FIXED launch sequence with nothing read back: the number
    of observed points stays on the device (``sdfr_depth_count_centroid`` leaves count and centroid, the per-point
    layers are ``sdfr_pointnet_layer_counted``), the orientation cell's quaternion comes from a table uploaded once
    (``SO3Grid.index_to_quat`` of every cell), the camera -> world transform and the "first" / "best" choice are one
    small kernel (``sdfr_init_estimate``) that writes the estimate where the loop reads its parameters.  Every buffer
    is allocated at construction (points and activations at capacity, W * H rows), so the sequence can be captured
    into a hipGraph (``use_graph``) and the front door's call has no host synchronisation between the observation
    and the result.

    ``nn_init`` above stays the host-driven form (one call of ~25 launches, two host reads): same layers, same
    arithmetic per point; the centroid here is the count pass's fixed-order block-sum tree instead of ``torch.mean``
    (a few 1e-7 relative apart), so the two agree to rounding, not to the bit (tests/test_init_network_gpu.py).
    The empty-cloud check of :780-781 cannot raise before the launches it would prevent: ``empty_views()`` reads the
    counts afterwards (one small copy; the caller raises ``NoDepthError``)."""

    def __init__(self, network: SDFPoseNet, camera: Camera, views: int, config: Dict, normalize_pose: bool = True,
                 objects: bool = False):
        """objects: the `views` images are `views` different OBJECTS seen from one camera (cam_pos / cam_quat hold one
        row): every image gets its own estimate, ``self.params`` is (views, 8 + L) -- the initialisation of
        ``pipeline.MultiObjectRenderAndCompare``; the "first" / "best" choice among views does not apply."""
        net = self.net = network
        self.cam, self.V = camera, int(views)
        self.objects = bool(objects)
        self.dev, self.L = net.dev, net.L            # (L: the loaded library, as in SDFPoseNet)
        self.normalize_pose = bool(normalize_pose)
        self.mean_shape = bool(config.get("mean_shape", False))
        self.strategy = config.get("init_view", "first")
        if self.strategy not in ("first", "best"):
            raise NotImplementedError('Only "first" and "best" strategies are currently supported')
        if self.strategy == "best" and net.orientation_repr != "discretized":
            raise NotImplementedError('"best" init strategy only supported with discretized orientation representation')
        if net.in_size != 3:
            raise NotImplementedError("the resident form feeds back-projected points (in_size = 3)")
        H, W = camera.height, camera.width
        self.H, self.W, M = H, W, H * W
        f32 = dict(dtype=torch.float32, device=self.dev)
        # the plan of the backbone: what `features` decides per call, decided once
        self.plan = []
        F_width, G_width, prev_width, n = net.in_size, 0, net.in_size, len(net.pn)
        for i, layer in enumerate(net.pn):
            last = i == n - 1
            if layer.cin_total != F_width + G_width:
                raise RuntimeError("state dict does not match the backbone configuration")
            out_width = layer.cout * (2 if (net.dense and not last) else 1)
            use_res = net.residual and prev_width == out_width
            if use_res and F_width != layer.cout:
                raise NotImplementedError("a residual link over a concatenated dense input: use nn_init")
            if last and use_res:
                raise NotImplementedError("a residual link into the last backbone layer: use nn_init")
            self.plan.append(dict(layer=layer, F_width=F_width, G_width=G_width, use_res=use_res, last=last,
                                  store=not last))
            G_width = layer.cout if (net.dense and not last) else 0
            F_width, prev_width = layer.cout, out_width
        width = max([p["layer"].cout for p in self.plan if p["store"]] + [4])
        self.points = torch.zeros((M, 3), **f32)
        self.act = [torch.zeros((M, width), **f32) for _ in range(2)]       # ping-pong per-point activations
        self.colmax = [torch.zeros(p["layer"].cout, **f32) for p in self.plan]
        self.cvec = [torch.zeros(p["layer"].cout, **f32) for p in self.plan]
        self.G = [torch.zeros(p["layer"].cout, **f32) for p in self.plan]
        self.head_vec = [torch.zeros(l.cout, **f32) for l in net.head] + [torch.zeros(net.final.cout, **f32)]
        i32 = dict(dtype=torch.int32, device=self.dev)
        self.count = torch.zeros(1, **i32)
        self.offset1 = torch.zeros(1, **i32)
        self.counts_all = torch.zeros(self.V, **i32)
        self.centroid = torch.zeros(3, **f32)
        self.ws = torch.zeros(max(self.L.sdfr_depth_centroid_workspace_bytes(1, W, H), 256) + 16, dtype=torch.uint8,
                              device=self.dev)
        self.discretized = net.orientation_repr == "discretized"
        if self.discretized:
            C = net.grid.num_cells()
            self.C = C
            self.grid_quats = torch.tensor(np.stack([np.asarray(net.grid.index_to_quat(i), dtype=np.float64)
                                                     for i in range(C)]), **f32).contiguous()
            self.posterior = torch.zeros(C, **f32)
            self.prior = torch.zeros((self.V, C), **f32)
            self.train_prior = torch.zeros(C, **f32)
        self.index = torch.zeros(1, **i32)
        self.post_max = torch.zeros(1, **f32)
        self.best = torch.zeros(1, **f32)
        self.params = torch.zeros((self.V, 8 + net.shape_dimension) if self.objects else (8 + net.shape_dimension,), **f32)
        self._graphs = {}
        self._host_counts = torch.zeros(self.V, dtype=torch.int32).pin_memory()
        self._counts_event = None

    def _vec(self, layer, koff, x, k, bn, relu, out):
        rc = self.L.sdfr_linear_vec(layer.w.data_ptr(), layer.cin_total, koff, x.data_ptr() if x is not None else None, k,
                                    layer.b.data_ptr(), layer.scale.data_ptr() if bn else None,
                                    layer.shift.data_ptr() if bn else None, int(relu), out.data_ptr(), layer.cout,
                                    self.dev.index, self._st)
        _lib.check(rc, "sdfr_linear_vec")

    def _sequence(self, depth: torch.Tensor, cam_pos: torch.Tensor, cam_quat: torch.Tensor, has_prior: bool,
                  has_train: bool):
        """the launches, on the current stream; depth (V,H,W) float32 contiguous on the device"""
        L, net, d = self.L, self.net, self.dev.index
        self._st = st = torch.cuda.current_stream(self.dev).cuda_stream
        fx, fy, cx0, cy0, _ = self.cam.get_pinhole_camera_parameters(0.0)
        W, H, M = self.W, self.H, self.W * self.H
        ws = self.ws[(-self.ws.data_ptr()) % 16:]
        views = range(self.V) if (self.strategy == "best" or self.objects) else range(1)
        best = self.strategy == "best" and not self.objects
        if best:
            self.best.zero_()
        for v in views:
            img = depth[v]
            _lib.check(L.sdfr_depth_count_centroid(img.data_ptr(), 1, W, H, 0, 1.0 / fx, 1.0 / fy, cx0, cy0,
                                                   self.count.data_ptr(), self.offset1.data_ptr(), self.centroid.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), d, st), "sdfr_depth_count_centroid")
            _lib.check(L.sdfr_depth_to_points_shifted(img.data_ptr(), 1, W, H, 0, 1.0 / fx, 1.0 / fy, cx0, cy0,
                                                      self.offset1.data_ptr(), ws.data_ptr(),
                                                      self.centroid.data_ptr() if self.normalize_pose else None, None,
                                                      self.points.data_ptr(), d, st), "sdfr_depth_to_points_shifted")
            self.counts_all[v:v + 1].copy_(self.count)
            F, ldF = self.points, 3
            G = None
            for i, p in enumerate(self.plan):
                layer = p["layer"]
                if G is not None:     # bias + what the concatenated set maximum contributes to every point
                    self._vec(layer, p["F_width"], G, p["G_width"], False, False, self.cvec[i])
                    cvec = self.cvec[i]
                else:
                    cvec = layer.b
                Y = self.act[i % 2] if p["store"] else None
                ldy = Y.shape[1] if Y is not None else layer.cout
                rc = L.sdfr_pointnet_layer_counted(
                    F.data_ptr(), self.count.data_ptr(), M, p["F_width"], ldF, layer.w.data_ptr(), layer.cin_total,
                    cvec.data_ptr(), layer.scale.data_ptr(), layer.shift.data_ptr(),
                    F.data_ptr() if p["use_res"] else None, Y.data_ptr() if Y is not None else None, ldy, layer.cout,
                    self.colmax[i].data_ptr(), d, st)
                _lib.check(rc, "sdfr_pointnet_layer_counted")
                if net.dense and not p["last"]:
                    if p["use_res"] and G is not None:
                        torch.add(self.colmax[i], G, out=self.G[i])
                        G = self.G[i]
                    else:
                        G = self.colmax[i]
                else:
                    G = None
                if not p["last"]:
                    F, ldF = Y, ldy
            out = self.colmax[-1]
            for j, layer in enumerate(net.head):
                self._vec(layer, 0, out, layer.cin_total, True, True, self.head_vec[j])
                out = self.head_vec[j]
            self._vec(net.final, 0, out, net.final.cin_total, False, False, self.head_vec[-1])
            head = self.head_vec[-1]
            sd = net.shape_dimension
            if self.discretized:
                _lib.check(L.sdfr_orientation_posterior(
                    head.data_ptr() + 4 * (sd + 4), self.C, self.prior[v].data_ptr() if has_prior else None,
                    self.train_prior.data_ptr() if (has_prior and has_train) else None, self.posterior.data_ptr(),
                    self.index.data_ptr(), self.post_max.data_ptr(), d, st), "sdfr_orientation_posterior")
            _lib.check(L.sdfr_init_estimate(
                head.data_ptr(), sd, self.grid_quats.data_ptr() if self.discretized else None,
                self.index.data_ptr() if self.discretized else None,
                self.centroid.data_ptr() if self.normalize_pose else None,
                cam_pos[0 if self.objects else v].data_ptr(), cam_quat[0 if self.objects else v].data_ptr(),
                int(self.mean_shape), int(best), self.post_max.data_ptr(), self.best.data_ptr(),
                (self.params[v] if self.objects else self.params).data_ptr(), d, st), "sdfr_init_estimate")

    def __call__(self, depth: torch.Tensor, cam_pos: torch.Tensor, cam_quat: torch.Tensor,
                 prior_orientation_distribution: Optional[torch.Tensor] = None,
                 training_orientation_distribution: Optional[torch.Tensor] = None, use_graph: bool = True):
        """depth (V,H,W), cam_pos (V,3), cam_quat (V,4): float32, contiguous, on the device, and THE SAME TENSORS on
        every call when ``use_graph`` (their addresses are part of the captured sequence: the loop's target and camera
        buffers are such tensors).  Returns (latent (1,L), position (1,3), scale (1,), orientation (1,4)) as views of
        ``self.params``."""
        if prior_orientation_distribution is not None and not self.discretized:
            raise ValueError("prior_orientation_distribution only supported for discretized orientation representation.")
        has_prior = prior_orientation_distribution is not None
        has_train = has_prior and training_orientation_distribution is not None
        if has_prior:
            self.prior.copy_(prior_orientation_distribution.reshape(self.V, self.C))
        if has_train:
            self.train_prior.copy_(training_orientation_distribution.reshape(self.C))
        if use_graph:
            key = (depth.data_ptr(), cam_pos.data_ptr(), cam_quat.data_ptr(), has_prior, has_train)
            g = self._graphs.get(key)
            if g is None:
                side = torch.cuda.Stream(self.dev)        # warm-up on a side stream (lazy module loads), then capture
                side.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(side):
                    self._sequence(depth, cam_pos, cam_quat, has_prior, has_train)
                torch.cuda.current_stream(self.dev).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._sequence(depth, cam_pos, cam_quat, has_prior, has_train)
                if len(self._graphs) >= 4:      # (a caller that hands other tensors every call: re-capture, no hoard)
                    self._graphs.pop(next(iter(self._graphs)))
                self._graphs[key] = g
            g.replay()
        else:
            self._sequence(depth, cam_pos, cam_quat, has_prior, has_train)
        # the counts for the empty-cloud check, without waiting for them here
        self._host_counts.copy_(self.counts_all, non_blocking=True)
        self._counts_event = torch.cuda.Event()
        self._counts_event.record(torch.cuda.current_stream(self.dev))
        sd = self.net.shape_dimension
        p = self.params
        if self.objects:
            return p[:, 8:8 + sd], p[:, 0:3], p[:, 7], p[:, 3:7]
        return p[8:8 + sd][None], p[0:3][None], p[7:8], p[3:7][None]

    def empty_views(self):
        """the views (of those the strategy looked at) without a single observed point -- waits for the counts of the
        last call only (they were copied behind its launches), not for whatever was enqueued after them"""
        if self._counts_event is None:
            return []
        self._counts_event.synchronize()
        n = self.V if (self.strategy == "best" or self.objects) else 1
        return [v for v in range(n) if int(self._host_counts[v]) == 0]
