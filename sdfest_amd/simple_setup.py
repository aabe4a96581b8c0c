"""``SDFPipeline``: the front door of the render-and-compare estimator, with the call signature of
``sdfest/estimation/simple_setup.py::SDFPipeline`` (:35-89 constructor, :213-226 ``__call__``, :583-596 result).

A caller of the reference replaces ``from sdfest.estimation.simple_setup import SDFPipeline`` by
``from sdfest_amd import SDFPipeline`` and keeps its code: same config dictionary
(``estimation/configs/default.yaml`` + a model file such as ``configs/models/mug.yaml``), same arguments, same
4-tuple back.  Inside, one call is

    _preprocess_depth   one in-place kernel (``sdfr_preprocess_depth``; :671-693)
    _nn_init            the initialisation network's forward (``init_network.nn_init``; :718-844)
    the 50 iterations   the graph-captured loop (``FusedRenderAndCompare``), built and captured on the FIRST call
                        for a number of views and re-bound to every later observation (``rebind``): no allocation,
                        no capture, no Python per iteration (:408-470)
    result selection    :583-596

What is NOT here (out of scope): plots (``visualize``), step logs (``log_path``), animations
(``animation_path``), mesh export (``generate_mesh``) and weight downloads -- the first three arguments are accepted
and ignored with a warning, weights come as state dicts or from a local file.
"""
import os
import warnings
from typing import Dict, Mapping, Optional, Tuple

import torch

from .differentiable_renderer import Camera, parse_sdf_grad_mode, render_depth_gpu
from .init_network import NoDepthError, ResidentInit, SDFPoseNet, adjust_categorical_posterior, nn_init
from .pipeline import FusedRenderAndCompare, MultiObjectRenderAndCompare, _selection_strategy, preprocess_depth
from .vae import SDFDecoder

__all__ = ["SDFPipeline", "NoDepthError"]


def _load_state(source, what: str) -> Mapping:
    """a state dict as given, or ``torch.load`` of a LOCAL file (the reference downloads a missing file from
    ``model_url``, sdfest/utils.py:10-43: there is no network here, and a product should not fetch weights)"""
    if isinstance(source, Mapping):
        return source
    for base in ("", os.path.expanduser("~/.sdfest/model_weights/")):
        path = os.path.expanduser(os.path.join(base, str(source)))
        if os.path.exists(path):
            return torch.load(path, map_location="cpu")
    raise FileNotFoundError(f"{what}: no state dict given and {source!r} is not a local file "
                            "(weights are not downloaded; pass vae_state_dict= / init_state_dict=)")


class SDFPipeline:
    """SDF pose and shape estimation pipeline (reference: simple_setup.py:35-89).

    config: the reference's dictionary -- ``device``, ``camera`` {width, height, fx, fy, cx, cy, pixel_center},
    ``threshold``, ``max_iterations``, ``depth_weight``, ``pc_weight``, ``nn_weight``, ``mean_shape``, ``init_view``,
    ``init`` {backbone_type, backbone, head_type, head, normalize_pose, model}, ``vae`` {latent_size, decoder, tsdf,
    model} (or ``init["vae"]``), optionally ``far_field``, ``result_selection_strategy``,
    ``relative_inlier_threshold``.
    vae_state_dict / init_state_dict: the state dicts of the reference's ``SDFVAE`` (keys ``decoder.*``; encoder
    keys are ignored) and ``SDFPoseNet``; default: ``torch.load`` of ``config[...]["model"]`` if that file exists.
    init_network: an object called like ``nn_init``'s network, or a callable
    ``(depth_images, camera_positions, camera_orientations, prior, training_prior) -> (latent (1,L), position (1,3),
    scale (1,), orientation (1,4))`` replacing ``_nn_init`` altogether (tests, or a caller with its own initialiser).
    """

    def __init__(self, config: Dict, vae_state_dict: Optional[Mapping] = None,
                 init_state_dict: Optional[Mapping] = None, init_network=None, resident_init: bool = True) -> None:
        self._parse_config(config)
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("sdfest_amd runs on the GPU only (config['device'] must be a cuda device)")
        self._dev = torch.device("cuda", torch.cuda.current_device()) if dev.index is None else dev
        _selection_strategy(config)     # (the reference raises at the END of a call, :592-596; a typo fails early here)

        self.resolution = 64
        vae_state = _load_state(vae_state_dict if vae_state_dict is not None else self.vae_config.get("model"),
                                "vae")
        self.vae = SDFDecoder.from_config(self.vae_config, vae_state, device=self._dev, sdf_size=self.resolution)

        self._nn_init_override = None
        if callable(init_network) and not isinstance(init_network, SDFPoseNet):
            self._nn_init_override = init_network
            self.init_network = None
        elif init_network is not None:
            self.init_network = init_network
        else:
            if self.init_config["backbone_type"] != "VanillaPointNet" or self.init_config["head_type"] != "SDFPoseHead":
                raise NotImplementedError("the initialisation network is VanillaPointNet + SDFPoseHead "
                                          "(simple_setup.py:27)")
            init_state = _load_state(init_state_dict if init_state_dict is not None
                                     else self.init_config.get("model"), "init")
            self.init_network = SDFPoseNet(self.init_config["backbone"], self.init_config["head"],
                                           self.vae_config["latent_size"], init_state, device=self._dev)

        self.cam = Camera(**self.camera_config)
        # a plain attribute, as in the reference (:84-86; real_data.py:230-243 replaces it at run time)
        self.render = lambda sdf, pos, quat, i_s: render_depth_gpu(
            sdf, pos, quat, i_s, None, None, None, config["threshold"], self.cam, self.sdf_grad_mode)
        self.config = config
        self.log_data = []
        self._loops = {}     # (views, shape_optimization) -> the captured loop, re-bound per call
        # resident_init: the initialisation network as a captured launch sequence with nothing read back
        # (init_network.ResidentInit) where its architecture allows; False: the host-driven nn_init (two host reads)
        self._resident_wanted = bool(resident_init)
        self._residents = {}
        self._multi_loops, self._resident_objects = {}, {}     # estimate_objects: per (K, shape_optimization) / per K
        self._ignored_warned = False

    def _parse_config(self, config: Dict) -> None:
        """simple_setup.py:91-109"""
        self.device = config["device"]
        self.init_config = config["init"]
        self.vae_config = config["vae"] if "vae" in config else self.init_config["vae"]
        self.camera_config = config["camera"]
        self.result_selection_strategy = config.get("result_selection_strategy", "last_iteration")
        self._relative_inlier_threshold = config.get("relative_inlier_threshold", 0.03)
        # (the reference sets _far_field only when the key exists and then reads it unconditionally, :106-107, :692:
        # without the key its _preprocess_depth raises AttributeError; None here = no far-field clipping)
        self._far_field = config.get("far_field")
        self.sdf_grad_mode = parse_sdf_grad_mode(config.get("sdf_grad_mode"))
        self.config = config

    # ---- the reference's helpers, same names ------------------------------------------------------------------
    def _preprocess_depth(self, depth_images: torch.Tensor, masks: torch.Tensor) -> None:
        """simple_setup.py:671-693, in place"""
        preprocess_depth(depth_images, masks, self._far_field)

    _adjust_categorical_posterior = staticmethod(adjust_categorical_posterior)    # :977-1009

    def _nn_init(self, depth_images, camera_positions, camera_orientations, prior_orientation_distribution=None,
                 training_orientation_distribution=None) -> Tuple:
        """simple_setup.py:718-844: (latent (1,L), position (1,3), scale (1,), orientation (1,4)), world frame"""
        if self._nn_init_override is not None:
            return self._nn_init_override(depth_images, camera_positions, camera_orientations,
                                          prior_orientation_distribution, training_orientation_distribution)
        return nn_init(self.init_network, self.cam, depth_images, camera_positions, camera_orientations, self.config,
                       normalize_pose=bool(self.init_config.get("normalize_pose", False)),
                       prior_orientation_distribution=prior_orientation_distribution,
                       training_orientation_distribution=training_orientation_distribution)

    def _resident_init(self, views: int):
        """the captured form of ``_nn_init`` for `views` images, or None where only the host-driven form applies"""
        if not self._resident_wanted or self._nn_init_override is not None or not isinstance(self.init_network, SDFPoseNet):
            return None
        if views not in self._residents:
            self._bounded(self._residents, self.MAX_CACHED_LOOPS)
            try:
                self._residents[views] = ResidentInit(self.init_network, self.cam, views, self.config,
                                                      normalize_pose=bool(self.init_config.get("normalize_pose", False)))
            except NotImplementedError:
                self._residents[views] = None
        return self._residents[views]

    def generate_depth(self, position, orientation, scale, latent) -> torch.Tensor:
        """simple_setup.py:609-619"""
        sdf = self.vae.decode(latent)
        return self.render(sdf[0, 0], position, orientation, 1 / scale)

    MAX_CACHED_LOOPS = 4     # per kind: a loop holds its point clouds at capacity (3.7 MB per 640x480 view) and its graphs

    @staticmethod
    def _bounded(cache: dict, limit: int) -> None:
        """drop the least recently built entry before a new one goes in (a caller whose number of views or objects
        varies from call to call re-builds instead of collecting buffers without bound)"""
        while len(cache) >= limit:
            cache.pop(next(iter(cache)))

    def _loop(self, views: int, shape_optimization: bool) -> FusedRenderAndCompare:
        key = (int(views), bool(shape_optimization))
        loop = self._loops.get(key)
        if loop is None:
            self._bounded(self._loops, self.MAX_CACHED_LOOPS)
            loop = FusedRenderAndCompare(self.vae, self.cam, self.config, views=views,
                                         shape_optimization=shape_optimization, device=self._dev,
                                         sdf_grad_mode=self.sdf_grad_mode)
            self._loops[key] = loop
        return loop

    def __call__(self, depth_images: torch.Tensor, masks: torch.Tensor, color_images: torch.Tensor,
                 visualize: bool = False, camera_positions: Optional[torch.Tensor] = None,
                 camera_orientations: Optional[torch.Tensor] = None, log_path: Optional[str] = None,
                 shape_optimization: bool = True, animation_path: Optional[str] = None,
                 point_constraint: Optional[Tuple[torch.Tensor]] = None,
                 prior_orientation_distribution: Optional[torch.Tensor] = None,
                 training_orientation_distribution: Optional[torch.Tensor] = None) -> tuple:
        """Infer pose, size and latent shape from depth and mask -- arguments as simple_setup.py:213-296.

        depth_images (N,H,W) or (H,W): masked and far-field-clipped IN PLACE (pass a copy if the full depth is used
        afterwards); masks: same shape, bool; color_images: unused (visualisation only in the reference).
        Returns (position (1,3), orientation (1,4) normalised scalar-last, scale (1,), latent (1,L))."""
        if (visualize or log_path is not None or animation_path is not None) and not self._ignored_warned:
            warnings.warn("sdfest_amd.SDFPipeline: visualize / log_path / animation_path are accepted and ignored "
                          "(plots, step logs and animations are outside the hot path)")
            self._ignored_warned = True
        # batch dimension (:306-318)
        if depth_images.dim() == 2:
            depth_images = depth_images.unsqueeze(0)
            masks = masks.unsqueeze(0)
            if camera_positions is not None:
                camera_positions = camera_positions.unsqueeze(0)
            if camera_orientations is not None:
                camera_orientations = camera_orientations.unsqueeze(0)
            if prior_orientation_distribution is not None:
                prior_orientation_distribution = prior_orientation_distribution.unsqueeze(0)
        n_imgs = depth_images.shape[0]
        dev = self._dev
        # (the default cameras -- the origin, the identity: :327-331 -- are left as None for ``rebind``, which writes them
        # into the loop's own buffers; only the host-driven initialisation below needs them as tensors)

        loop = self._loop(n_imgs, shape_optimization)
        with torch.no_grad():
            # :333-334, and the observation into the loop's buffers in the same pass
            if (depth_images.is_cuda and depth_images.dtype is torch.float32 and depth_images.is_contiguous()
                    and depth_images.device == dev):
                loop.rebind(depth_images, camera_positions, camera_orientations, point_constraint, masks=masks,
                            far_field=self._far_field)
            else:
                # any other tensor the reference would take (another dtype or device, a strided view): its own two
                # in-place assignments, then the copy into the loop's buffers
                depth_images[~masks.to(device=depth_images.device, dtype=torch.bool)] = 0
                if self._far_field is not None:
                    depth_images[depth_images > self._far_field] = 0
                loop.rebind(depth_images, camera_positions, camera_orientations, point_constraint)
                depth_images = loop.target if loop.group is None else depth_images.to(dev, torch.float32)
            # :352-359
            resident = self._resident_init(n_imgs)
            if resident is not None:
                # on the loop's own (address-stable) buffers: the preprocessed images and the cameras rebind left there
                latent_shape, position, scale, orientation = resident(
                    loop.target, loop.cam_pos_all, loop.cam_quat_all, prior_orientation_distribution,
                    training_orientation_distribution)
            else:
                latent_shape, position, scale, orientation = self._nn_init(
                    depth_images, loop.cam_pos_all if camera_positions is None else camera_positions,
                    loop.cam_quat_all if camera_orientations is None else camera_orientations,
                    prior_orientation_distribution, training_orientation_distribution)
            # :381-470
            position, orientation, scale, latent_shape = loop(position, orientation, scale, latent_shape)
            if resident is not None and resident.empty_views():
                # (:780-781 -- known only now: the counts were never waited for in front of the launches)
                raise NoDepthError
        self._last_loop = loop
        # :583-596 -- "best_inlier_ratio" returns the tensors the reference stored, which its optimiser went on
        # updating in place: the last iterate as well (pipeline._BestEstimate); the snapshot at the best ratio is
        # self.best_estimate()
        return position, orientation, scale, latent_shape

    def estimate_objects(self, depth_image: torch.Tensor, masks: torch.Tensor,
                         camera_position: Optional[torch.Tensor] = None,
                         camera_orientation: Optional[torch.Tensor] = None, shape_optimization: bool = True,
                         prior_orientation_distribution: Optional[torch.Tensor] = None,
                         training_orientation_distribution: Optional[torch.Tensor] = None) -> tuple:
        """The K detected objects of ONE frame at once: what a caller of the reference does with a loop of K
        ``pipeline(depth, mask_k, color)`` calls (one per instance mask), as one call whose K estimates are optimised side
        by side (``pipeline.MultiObjectRenderAndCompare``: one launch sequence per iteration for all of them).

        depth_image (H,W): the frame (NOT modified: every object gets its own masked copy); masks (K,H,W) bool: the
        instance masks; camera_position (3,) / camera_orientation (4,): the camera in the world (default: the origin);
        prior_orientation_distribution (K,C) / training_orientation_distribution (C,): as in ``__call__``, one row per
        object.  Returns position (K,3), orientation (K,4), scale (K,), latent (K,L) -- row k follows what
        ``pipeline(depth, masks[k], color)`` estimates for object k (`result_selection_strategy` "last_iteration") TO
        ROUNDING, tested to 1 % of an Adam step per iteration: the single call runs its render pair as one launch
        (``FusedRenderAndCompare(fused_render=)``: the depth loss's weight / count multiplies sums instead of terms), and with
        shape optimisation a batch of latents takes other, equivalent decoder kernels than a single one (the direct
        convolution from 8 latents on, the tiled resize from 2, no split-K above 16).  Against the single loop in its
        two-launch form the pose-only rows are BIT FOR BIT the same (tests/test_multi_object_gpu.py).  ``NoDepthError`` for an object without a valid depth pixel is raised AFTER the call's work was
        enqueued (the counts are read behind the launches), where the reference raises before optimising (:780-781)."""
        dev = self._dev
        if depth_image.dim() != 2 or masks.dim() != 3 or tuple(masks.shape[1:]) != tuple(depth_image.shape):
            raise ValueError("depth_image (H,W) and masks (K,H,W) are expected")
        K = int(masks.shape[0])
        key = (K, bool(shape_optimization))
        loop = self._multi_loops.get(key)
        if loop is None:
            self._bounded(self._multi_loops, self.MAX_CACHED_LOOPS)
            loop = self._multi_loops[key] = MultiObjectRenderAndCompare(
                self.vae, self.cam, self.config, K, shape_optimization=shape_optimization, device=dev,
                sdf_grad_mode=self.sdf_grad_mode)
        with torch.no_grad():
            frames = depth_image.to(device=dev, dtype=torch.float32)[None].expand(K, -1, -1).contiguous()
            loop.rebind(frames, camera_position, camera_orientation, masks=masks, far_field=self._far_field)
            resident = None
            if self._resident_wanted and self._nn_init_override is None and isinstance(self.init_network, SDFPoseNet):
                if K not in self._resident_objects:
                    self._bounded(self._resident_objects, self.MAX_CACHED_LOOPS)
                    try:
                        self._resident_objects[K] = ResidentInit(
                            self.init_network, self.cam, K, dict(self.config, init_view="first"),
                            normalize_pose=bool(self.init_config.get("normalize_pose", False)), objects=True)
                    except NotImplementedError:
                        self._resident_objects[K] = None
                resident = self._resident_objects[K]
            if resident is not None:
                latent, position, scale, orientation = resident(loop.target, loop.cam_pos, loop.cam_quat,
                                                                prior_orientation_distribution,
                                                                training_orientation_distribution)
            else:     # the host-driven form, object by object
                rows = []
                for k in range(K):
                    prior = prior_orientation_distribution[k:k + 1] if prior_orientation_distribution is not None else None
                    rows.append(self._nn_init(loop.target[k:k + 1], loop.cam_pos, loop.cam_quat, prior,
                                              training_orientation_distribution))
                latent, position, scale, orientation = (torch.cat([r[i].reshape(1, -1) for r in rows]) for i in range(4))
                scale = scale.reshape(K)
            out = loop(position, orientation, scale, latent)
            if resident is not None and resident.empty_views():
                raise NoDepthError
        self._last_multi_loop = loop
        return out

    def prepare(self, views: int = 1, shape_optimization: bool = True) -> "SDFPipeline":
        """Optional: build the loop for `views` images and capture its graphs NOW (on an empty observation) instead of
        inside the first call -- a service that must answer its first request in the steady-state time calls this
        after construction.  The reference has no counterpart (it re-uses nothing between calls)."""
        loop = self._loop(views, shape_optimization)
        with torch.no_grad():
            H, W = self.cam.height, self.cam.width
            loop.rebind(torch.zeros((views, H, W), device=self._dev))
            z = torch.zeros((1, self.vae.latent_size), device=self._dev)
            loop(torch.tensor([[0.0, 0.0, -1.0]], device=self._dev), torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=self._dev),
                 torch.tensor([0.1], device=self._dev), z)
            resident = self._resident_init(views)
            if resident is not None:
                resident(loop.target, loop.cam_pos_all, loop.cam_quat_all)
            torch.cuda.synchronize(self._dev)
        return self

    def best_estimate(self):
        """(ratio, 1-based iteration, (position, orientation, scale, latent)) at the best inlier ratio of the last call
        (``result_selection_strategy == "best_inlier_ratio"``); None otherwise"""
        loop = getattr(self, "_last_loop", None)
        return loop.best_estimate() if loop is not None and loop.track_inliers else None
