"""Batched synthetic-view generator: the forward-only caller of the renderer.

Mirror of ``SDFVAEViewDataset._generate_sample`` (sdfest/initialization/datasets/
generated_dataset.py:247-342): sample a latent shape, decode it, draw a random pose inside the
frustum, render the depth image, optionally smooth it, back-project it to a point set.  The
reference produces one sample per call (one decode, one render launch, ~20 small torch ops);
here a whole batch goes through one batched decode and ONE render launch with one SDF per view
(``sdf_view_stride = R^3``), and the per-sample post-processing runs on the packed batch.

Random numbers: the reference draws from Python's ``random`` and ``torch.randn``; this module
draws the same distributions from one ``torch.Generator`` (reproducible from a seed).  Every
sampled quantity can also be passed in, which is how the tests compare with per-sample calls.
``mask_noise`` (generated_dataset.py:234-245, :286-291): the reference perturbs the exact mask with
torchvision's ``RandomAffine(degrees=(0, 1), translate=(0.00, 0.01), scale=(0.999, 1.001))``
(torchvision==0.12.0, requirements.txt:148 -- not in this image, not under /root/reference).  Its
published algorithm is restated: the parameter distributions of ``RandomAffine.get_params``, the
inverse matrix of ``functional._get_inverse_affine_matrix`` and the nearest-neighbour resampling of
``functional_tensor.affine`` (= torch's ``grid_sample``), the last as one HIP kernel
(``sdfr_affine_mask``) pinned against ``torch.nn.functional.grid_sample``.
``orientation_repr="discretized"`` (:169-175, :358-360): :mod:`sdfest_amd.so3grid`.
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch

from .differentiable_renderer import BatchRenderPlan, Camera

# generated_dataset.py:96-115 (the keys this generator understands)
DEFAULT_CONFIG = {
    "width": 640, "height": 480, "fov_deg": 90, "render_threshold": 0.004,
    "pointcloud": True, "normalize_pose": None, "orientation_repr": "quaternion",
    "mask_noise": False, "mask_noise_min": 0.1, "mask_noise_max": 2.0, "orientation_grid_resolution": 2,
    "norm_noise": False, "norm_noise_min": -0.2, "norm_noise_max": 0.2,
    "scale_to_unit_ball": False, "gaussian_noise_probability": 0.0,
    "gaussian_noise_kernel_size": 5, "gaussian_noise_kernel_std": 1,
}


def sample_uniform_quaternions(n: int, gen: Optional[torch.Generator] = None) -> torch.Tensor:
    """(n,4) uniformly distributed unit quaternions, Shoemake 1992 (generated_dataset.py:187-207)."""
    u = torch.rand((n, 3), generator=gen, dtype=torch.float64)
    a, b = torch.sqrt(1 - u[:, 0]), torch.sqrt(u[:, 0])
    t2, t3 = 2 * math.pi * u[:, 1], 2 * math.pi * u[:, 2]
    return torch.stack((a * torch.sin(t2), a * torch.cos(t2), b * torch.sin(t3), b * torch.cos(t3)), 1).float()


def sample_poses(n: int, camera: Camera, z_min: float, z_max: float, extent_mean: float,
                 extent_std: float, gen: Optional[torch.Generator] = None):
    """Positions with the centre inside the frustum, uniform orientations, Gaussian half-extents
    (generated_dataset.py:262-271).  Returns (position (n,3), quaternion (n,4), scale (n,)) on CPU.

    Note the reference's own bounds: x_pix ~ U(-width/2, height/2) (sic), y_pix ~ U(-height/2,
    height/2); kept as they are."""
    u = torch.rand((n, 3), generator=gen, dtype=torch.float64)
    z = z_min + (z_max - z_min) * u[:, 0]
    x_pix = -camera.width / 2 + (camera.height / 2 + camera.width / 2) * u[:, 1]
    y_pix = -camera.height / 2 + camera.height * u[:, 2]
    position = torch.stack((x_pix / camera.fx * z, y_pix / camera.fy * z, -z), 1).float()
    scale = (extent_mean + extent_std * torch.randn(n, generator=gen, dtype=torch.float64)) / 2.0
    return position, sample_uniform_quaternions(n, gen), scale.float()


def inverse_affine_matrices(angle_deg: torch.Tensor, translate: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """(n,6) row-major 2x3 inverse affine maps about the image centre, in pixels, no shear:
    torchvision 0.12 ``functional._get_inverse_affine_matrix(center=[0, 0], angle, translate, scale,
    shear=[0, 0])`` -- rotation by ``angle`` degrees, then the inverse translation."""
    rot = angle_deg.double() * (math.pi / 180.0)
    a, b, c, d = torch.cos(rot), -torch.sin(rot), torch.sin(rot), torch.cos(rot)
    sc = scale.double()
    m0, m1, m3, m4 = d / sc, -b / sc, -c / sc, a / sc
    tx, ty = translate[:, 0].double(), translate[:, 1].double()
    m2 = m0 * (-tx) + m1 * (-ty)
    m5 = m3 * (-tx) + m4 * (-ty)
    return torch.stack((m0, m1, m2, m3, m4, m5), 1).float()


def sample_mask_affine(n: int, width: int, height: int, gen: Optional[torch.Generator] = None):
    """``RandomAffine.get_params`` for degrees=(0, 1), translate=(0.00, 0.01), scale=(0.999, 1.001)
    (generated_dataset.py:242-244): angle ~ U(0, 1) degrees, whole-pixel shifts round(U(-t W, t W)),
    round(U(-t H, t H)), scale ~ U(0.999, 1.001).  Returns (angle (n,), translate (n,2), scale (n,))."""
    u = torch.rand((n, 4), generator=gen, dtype=torch.float64)
    angle = u[:, 0] * 1.0
    max_dx, max_dy = 0.00 * width, 0.01 * height
    translate = torch.stack((torch.round(-max_dx + 2 * max_dx * u[:, 1]), torch.round(-max_dy + 2 * max_dy * u[:, 2])), 1)
    scale = 0.999 + 0.002 * u[:, 3]
    return angle, translate, scale


def perturb_masks(depth: torch.Tensor, matrices: torch.Tensor) -> torch.Tensor:
    """(B,H,W) bool: the exact masks ``depth != 0`` under the inverse affine maps ``matrices`` (B,6),
    nearest neighbour, False outside (generated_dataset.py:234-245 on a batch)."""
    from . import _lib
    B, H, W = depth.shape
    dev = depth.device
    depth = depth.contiguous()
    m = matrices.to(device=dev, dtype=torch.float32).contiguous()
    if tuple(m.shape) != (B, 6):
        raise RuntimeError(f"matrices must have shape ({B}, 6)")
    out = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().sdfr_affine_mask(depth.data_ptr(), B, W, H, m.data_ptr(), out.data_ptr(), dev.index,
                                           torch.cuda.current_stream(dev).cuda_stream), "sdfr_affine_mask")
    return out.bool()


def gaussian_kernel(std: float, kernel_size: int) -> torch.Tensor:
    """The smoothing kernel of generated_dataset.py:366-373: scipy's gaussian_filter applied to a
    unit impulse."""
    from scipy.ndimage import gaussian_filter
    if kernel_size % 2 != 1:
        raise ValueError("Kernel size should be odd.")
    impulse = np.zeros((kernel_size, kernel_size))
    impulse[kernel_size // 2, kernel_size // 2] = 1
    return torch.tensor(gaussian_filter(impulse, std)[None, None], dtype=torch.float32)


def smooth_depth(depth: torch.Tensor, kernel: torch.Tensor, apply: torch.Tensor) -> torch.Tensor:
    """generated_dataset.py:296-308 on a batch: where a pixel's whole kernel window is valid the
    depth becomes the filtered value, every other pixel keeps its value (invalid stays 0).
    ``apply`` (B,) bool selects the samples that are filtered at all.  In place; returns depth."""
    # The reference marks invalid pixels NaN and lets conv2d spread them; which outputs a NaN
    # reaches depends on the convolution algorithm (a transform-based one contaminates more than
    # the window).  Same result without NaNs: a window is usable iff it holds no invalid pixel.
    k = kernel.to(depth)[0, 0]
    kh, kw = k.shape
    H, W = depth.shape[-2:]
    pad = torch.nn.functional.pad
    dp = pad(depth, (kw // 2, kw // 2, kh // 2, kh // 2))                       # zero padding = "same"
    ip = pad((depth == 0).to(depth.dtype), (kw // 2, kw // 2, kh // 2, kh // 2))
    filtered = torch.zeros_like(depth)
    window_invalid = torch.zeros_like(depth)
    for i in range(kh):      # direct correlation, tap by tap (no convolution library: 25 fused ops)
        for j in range(kw):
            filtered += k[i, j] * dp[:, i:i + H, j:j + W]
            window_invalid += ip[:, i:i + H, j:j + W]
    ok = (window_invalid == 0) & apply.to(depth.device)[:, None, None]
    depth[ok] = filtered[ok]
    return depth


def depth_to_pointsets(depth: torch.Tensor, camera: Camera, tiled: bool = False):
    """Back-projection of every non-zero pixel of a (B,H,W) batch (pointset_utils.py:57-77,
    convention "opengl", pixel-centre-0 intrinsics): packed points (N,3) in view-major, row-major
    order and the per-view counts.  tiled (GPU only): the same points of each view in the order of
    ``SDFR_POINT_ORDER_TILED`` (include/sdfr.h) -- for consumers that only sum over a view's points."""
    fx, fy, cx, cy, _ = camera.get_pinhole_camera_parameters(0.0)
    if depth.is_cuda:
        # two passes over the images (count, then a stable compaction) instead of torch.nonzero and
        # three gathers over B*H*W elements: sdfr_depth_count / sdfr_depth_to_points
        from . import _lib
        L = _lib.lib()
        depth = depth.contiguous()
        V, H, W = depth.shape
        dev = depth.device
        counts = torch.empty(V, dtype=torch.int32, device=dev)
        ws = torch.empty(max(L.sdfr_depth_points_workspace_bytes(V, W, H), 256), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        order = 1 if tiled else 0
        _lib.check(L.sdfr_depth_count_ordered(depth.data_ptr(), V, W, H, order, counts.data_ptr(), ws.data_ptr(),
                                              ws.numel(), dev.index, st), "sdfr_depth_count")
        counts64 = counts.to(torch.int64)
        offsets = (counts64.cumsum(0) - counts64).to(torch.int32)
        pts = torch.empty((int(counts64.sum()), 3), dtype=torch.float32, device=dev)   # (the one sync)
        if pts.shape[0]:
            _lib.check(L.sdfr_depth_to_points_ordered(depth.data_ptr(), V, W, H, order, 1.0 / fx, 1.0 / fy, cx, cy,
                                                      offsets.data_ptr(), ws.data_ptr(), pts.data_ptr(), dev.index, st),
                       "sdfr_depth_to_points")
        return pts, counts64
    if tiled:
        raise RuntimeError("the tiled point order is the GPU path's")
    b, rows, cols = torch.nonzero(depth, as_tuple=True)
    z = depth[b, rows, cols]
    pts = torch.stack(((cols.float() - cx) * z / fx, -(rows.float() - cy) * z / fy, -z), dim=1)
    return pts, torch.bincount(b, minlength=depth.shape[0])


class PointSets:
    """The per-view point sets of a packed batch as a read-only sequence: ``sets[b]`` is the (N_b, 3) view of
    ``points`` that ``torch.split`` would return, made when it is asked for (a batch of 256 views costs ~0.1 ms of
    host time to split eagerly, whether or not anybody reads the views)."""

    def __init__(self, points: torch.Tensor, counts_host):
        self.points = points
        self.bounds = np.concatenate([[0], np.cumsum(np.asarray(counts_host, dtype=np.int64))])

    def __len__(self):
        return len(self.bounds) - 1

    def __getitem__(self, b):
        if isinstance(b, slice):
            return [self[i] for i in range(*b.indices(len(self)))]
        if b < 0:
            b += len(self)
        if not 0 <= b < len(self):
            raise IndexError(b)
        return self.points[int(self.bounds[b]):int(self.bounds[b + 1])]

    def __iter__(self):
        return (self[b] for b in range(len(self)))


def depth_to_centred_pointsets(depth: torch.Tensor, camera: Camera, noise: Optional[torch.Tensor] = None,
                               while_waiting=None):
    """``depth_to_pointsets`` for the generator's normalised samples (generated_dataset.py:318-326): every view's
    points minus their centroid (plus ``noise`` (B,3), if given: ``(points - centroid) + noise``, the reference's two
    roundings), without a pass over the packed points -- the count pass leaves the centroids
    (``sdfr_depth_count_centroid``), the compaction subtracts them (``sdfr_depth_to_points_shifted``).  The centroid
    itself is a fixed-order float32 block-sum tree, not ``torch.mean``'s: equal to it within float32 rounding of the
    mean (a few 1e-7 relative), which the tests state.  Returns (points (N,3), counts (B,) int64 on the device, counts on the host,
    centroid (B,3)); one synchronisation (the caller sizes the output).  ``while_waiting``: called on the host
    after the count pass has been issued and before its result is waited for (CPU work that hides behind the GPU's)."""
    from . import _lib
    L = _lib.lib()
    fx, fy, cx, cy, _ = camera.get_pinhole_camera_parameters(0.0)
    depth = depth.contiguous()
    V, H, W = depth.shape
    dev = depth.device
    counts_offsets = torch.empty((2, V), dtype=torch.int32, device=dev)
    counts, offsets = counts_offsets[0], counts_offsets[1]
    centroid = torch.empty((V, 3), dtype=torch.float32, device=dev)
    ws = torch.empty(max(L.sdfr_depth_centroid_workspace_bytes(V, W, H), 256), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(L.sdfr_depth_count_centroid(depth.data_ptr(), V, W, H, 0, 1.0 / fx, 1.0 / fy, cx, cy, counts.data_ptr(),
                                           offsets.data_ptr(), centroid.data_ptr(), ws.data_ptr(), ws.numel(),
                                           dev.index, st), "sdfr_depth_count_centroid")
    noise_dev = None if noise is None else noise.to(centroid).contiguous()
    host = torch.empty(V, dtype=torch.int32).pin_memory()
    host.copy_(counts, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    if while_waiting is not None:
        while_waiting()
    done.synchronize()                                                  # (the one sync)
    counts_host = host.numpy()
    counts64 = counts.to(torch.int64)
    pts = torch.empty((int(counts_host.sum()), 3), dtype=torch.float32, device=dev)
    if pts.shape[0]:
        _lib.check(L.sdfr_depth_to_points_shifted(depth.data_ptr(), V, W, H, 0, 1.0 / fx, 1.0 / fy, cx, cy,
                                                  offsets.data_ptr(), ws.data_ptr(), centroid.data_ptr(),
                                                  noise_dev.data_ptr() if noise_dev is not None else None,
                                                  pts.data_ptr(), dev.index, st), "sdfr_depth_to_points_shifted")
    return pts, counts64, counts_host, centroid


class SDFVAEViewGenerator:
    """Batched ``SDFVAEViewDataset`` (generated_dataset.py:26-342).

    ``decoder`` is a :class:`sdfest_amd.SDFDecoder`; ``config`` uses the reference's keys (see
    DEFAULT_CONFIG; ``z_min``, ``z_max``, ``extent_mean``, ``extent_std`` are required)."""

    def __init__(self, config: Dict, decoder, batch_size: int = 64, device="cuda", seed: Optional[int] = None,
                 prefetch_draws: bool = False, decode_ahead: bool = False):
        """prefetch_draws: a ``generate()`` that draws its own latents and poses also draws the NEXT batch's while the
        GPU is busy with this one (the numbers and their order in the generator's stream stay what they were; a call
        that is given a latent or a pose discards what was drawn ahead for it).  For consumers that call
        ``generate()`` over and over -- iteration turns it on.
        decode_ahead (with prefetch_draws): the next batch's latents are also uploaded and DECODED ahead, on a second
        HIP stream, while this batch's render, noise and point-set kernels run on the caller's stream -- the decoder's
        kernels are bound by the vector ALUs, the march by the CU's vector-memory pipeline, the point-set passes by
        HBM, and the host's wait for the point counts no longer leaves the GPU idle.  Same kernels on the same
        numbers: the samples do not change."""
        self.prefetch_draws = bool(prefetch_draws)
        self.decode_ahead = bool(decode_ahead)
        self._ahead = None
        self._side = None          # the second stream and the batch decoded ahead on it
        self._decoded = None
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(config)
        for k in ("z_min", "z_max", "extent_mean", "extent_std"):
            if k not in cfg:
                raise KeyError(f"config key {k!r} is required")
        if cfg["orientation_repr"] == "discretized":                                 # :169-172
            from .so3grid import SO3Grid
            self.orientation_grid = SO3Grid(cfg["orientation_grid_resolution"])
        elif cfg["orientation_repr"] != "quaternion":
            raise NotImplementedError(f"Orientation representation {cfg['orientation_repr']} is not supported.")
        self.cfg = cfg
        self.decoder = decoder
        self.B = int(batch_size)
        self.device = torch.device(device)
        f = cfg["width"] / math.tan(cfg["fov_deg"] * math.pi / 180.0 / 2.0) / 2      # :123-132
        self.camera = Camera(cfg["width"], cfg["height"], f, f, cfg["width"] / 2, cfg["height"] / 2,
                             pixel_center=0.5)
        self.gen = torch.Generator()
        if seed is not None:
            self.gen.manual_seed(seed)
        self.kernel = gaussian_kernel(cfg["gaussian_noise_kernel_std"], cfg["gaussian_noise_kernel_size"])
        self.plan = BatchRenderPlan(decoder._volume_size, self.B, self.camera, device=self.device,
                                    per_view_sdf=True)

    # -- the GPU part: decode -> render (one launch each for the whole batch) ------------------
    def render(self, latent: torch.Tensor, position: torch.Tensor, quaternion: torch.Tensor,
               scale: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(B,H,W) depth images of decode(latent[b]) at pose b (generated_dataset.py:258-280).
        Without ``out`` the returned tensor is the plan's buffer (overwritten by the next call)."""
        with torch.no_grad():
            f32 = dict(device=self.device, dtype=torch.float32)   # caller-supplied poses may be float64
            sdf = self.decoder.decode(latent.to(**f32))[:, 0].contiguous()
            return self.plan.forward(sdf, position.to(**f32).contiguous(), quaternion.to(**f32).contiguous(),
                                     (1.0 / scale.to(**f32)).contiguous(), self.cfg["render_threshold"], out=out)

    def _decode_ahead(self):
        """Upload and decode the batch drawn ahead on the second stream (see ``decode_ahead``)."""
        dev = self.device
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        z0, p, q, s = self._ahead
        Lz = z0.shape[1]
        with torch.cuda.stream(self._side), torch.no_grad():
            packed = torch.cat((z0, p, q, s[:, None]), 1).to(dev, non_blocking=True)
            latent, position, quaternion, scale = (packed[:, :Lz], packed[:, Lz:Lz + 3].clone(),
                                                   packed[:, Lz + 3:Lz + 7], packed[:, Lz + 7].clone())
            sdf = self.decoder.decode(latent.contiguous())[:, 0].contiguous()
            ready = torch.cuda.Event()
            ready.record(self._side)
        self._decoded = (latent, position, quaternion, scale, sdf, ready)

    def generate(self, latent=None, position=None, quaternion=None, scale=None, smooth=None,
                 mask_affine=None, mask_noise_value=None) -> Dict:
        """One batch of samples; any of the sampled quantities may be given instead of drawn
        (mask_affine: (angle (B,), translate (B,2), scale (B,)); mask_noise_value: (B,) background depth).

        Returns a dict of batched tensors with the reference's keys: "depth" (B,H,W),
        "latent_shape" (B,L), "position" (B,3), "orientation" = "quaternion" (B,4), "scale" (B,),
        and with ``pointcloud``: "pointset" (list of (N_b,3) views of the packed "points") and
        "valid" (B,) = the reference's _is_valid (at least one depth pixel)."""
        cfg, B, dev = self.cfg, self.B, self.device
        ahead, self._ahead = self._ahead, None
        decoded, self._decoded = self._decoded, None
        draws_all = latent is None and position is None and quaternion is None and scale is None
        if not draws_all:
            decoded = None          # (a caller's latent or pose: what was decoded ahead is not this batch)
        if ahead is not None:
            z0, p, q, s = ahead
            latent = z0 if latent is None else latent
        else:
            if latent is None:
                latent = torch.randn((B, self.decoder.latent_size), generator=self.gen)     # SDFVAE.sample
            p, q, s = sample_poses(B, self.camera, cfg["z_min"], cfg["z_max"], cfg["extent_mean"],
                                   cfg["extent_std"], self.gen)

        def draw_ahead():
            # (after every draw of THIS batch, so the stream's order is the one of back-to-back calls)
            if self.prefetch_draws and draws_all:
                self._ahead = (torch.randn((B, self.decoder.latent_size), generator=self.gen),) + tuple(
                    sample_poses(B, self.camera, cfg["z_min"], cfg["z_max"], cfg["extent_mean"], cfg["extent_std"],
                                 self.gen))
                if self.decode_ahead and dev.type == "cuda":
                    self._decode_ahead()
        position, quaternion, scale = (p if position is None else position, q if quaternion is None else quaternion,
                                       s if scale is None else scale)
        host = [t for t in (latent, position, quaternion, scale) if not t.is_cuda]
        sdf_ahead = None
        if decoded is not None:
            # decoded ahead on the second stream: the device copies of the draws and the volumes; this stream waits
            latent, position, quaternion, scale, sdf_ahead, ready = decoded
            torch.cuda.current_stream(dev).wait_event(ready)
            for t in (latent, position, quaternion, scale, sdf_ahead):
                t.record_stream(torch.cuda.current_stream(dev))
        elif len(host) == 4 and all(t.dtype == torch.float32 for t in host):
            # one packed transfer instead of four small ones (each costs ~15 us of host time)
            Lz = latent.shape[1]
            packed = torch.cat((latent, position, quaternion, scale[:, None]), 1).to(dev, non_blocking=True)
            latent, position, quaternion, scale = (packed[:, :Lz], packed[:, Lz:Lz + 3].clone(),
                                                   packed[:, Lz + 3:Lz + 7], packed[:, Lz + 7].clone())
        else:
            position = position.to(dev).clone()
            quaternion = quaternion.to(dev)
            scale = scale.to(dev).clone()
        depth_out = torch.empty((B, cfg["height"], cfg["width"]), dtype=torch.float32, device=dev)
        if sdf_ahead is not None:
            with torch.no_grad():
                depth = self.plan.forward(sdf_ahead, position.contiguous(), quaternion.contiguous(),
                                          (1.0 / scale).contiguous(), cfg["render_threshold"], out=depth_out)
        else:
            # (a decode still running ahead on the second stream has its own scratch: SDFDecoder keeps one per stream)
            depth = self.render(latent, position, quaternion, scale, out=depth_out)
        final_mask = None
        if cfg["mask_noise"]:                                                          # :286-291
            if mask_affine is None:
                mask_affine = sample_mask_affine(B, cfg["width"], cfg["height"], self.gen)
            final_mask = perturb_masks(depth, inverse_affine_matrices(*mask_affine))
            if mask_noise_value is None:
                lo, hi = cfg["mask_noise_min"], cfg["mask_noise_max"]
                mask_noise_value = lo + (hi - lo) * torch.rand(B, generator=self.gen)
            # depth[~exact_mask] = one background depth per sample
            depth = torch.where(depth != 0, depth, mask_noise_value.to(depth)[:, None, None].expand_as(depth)).contiguous()
        if cfg["gaussian_noise_probability"] > 0.0:                                    # :296-308
            if smooth is None:
                smooth = torch.rand(B, generator=self.gen) < cfg["gaussian_noise_probability"]
            smooth_depth(depth, self.kernel, smooth)
        if final_mask is not None:
            depth = (depth * final_mask).contiguous()                                  # :310  depth[~final_mask] = 0
        out = {"depth": depth, "latent_shape": latent.to(dev)}
        if cfg["pointcloud"] and cfg["normalize_pose"] and not cfg["scale_to_unit_ball"] and depth.is_cuda:
            # :312-326 with the centroid formed and subtracted inside the two passes over the images
            noise = None
            if cfg["norm_noise"]:
                lo, hi = cfg["norm_noise_min"], cfg["norm_noise_max"]
                noise = (lo + (hi - lo) * torch.rand((B, 3), generator=self.gen)).to(dev)
            pts, counts, counts_host, centroid = depth_to_centred_pointsets(depth, self.camera, noise,
                                                                            while_waiting=draw_ahead)
            out["valid"] = counts > 0
            position -= centroid
            if noise is not None:
                position += noise
            out["points"], out["counts"] = pts, counts
            out["pointset"] = PointSets(pts, counts_host)
        elif cfg["pointcloud"]:                                                        # :312-334
            pts, counts = depth_to_pointsets(depth, self.camera)
            out["valid"] = counts > 0          # = the reference's depth.max() != 0, from the counts
            owner = torch.repeat_interleave(torch.arange(B, device=dev), counts)
            if cfg["normalize_pose"]:
                centroid = torch.zeros((B, 3), device=dev).index_add_(0, owner, pts)
                centroid = centroid / counts.clamp(min=1)[:, None]
                pts = pts - centroid[owner]
                position -= centroid
                if cfg["norm_noise"]:
                    lo, hi = cfg["norm_noise_min"], cfg["norm_noise_max"]
                    noise = (lo + (hi - lo) * torch.rand((B, 3), generator=self.gen)).to(dev)
                    position += noise
                    pts = pts + noise[owner]
                if cfg["scale_to_unit_ball"]:
                    # the reference divides by torch.max(torch.linalg.norm(pointset)): the norm of
                    # the whole (N,3) matrix, one number per sample (:330-332)
                    fro = torch.sqrt(torch.zeros(B, device=dev).index_add_(0, owner, (pts * pts).sum(1)))
                    fro = torch.where(counts > 0, fro, torch.ones_like(fro))
                    pts = pts / fro[owner][:, None]
                    scale /= fro
            out["points"], out["counts"] = pts, counts
            out["pointset"] = list(torch.split(pts, counts.tolist()))
        if "valid" not in out:
            out["valid"] = depth.amax(dim=(1, 2)) != 0
        out["position"], out["scale"] = position, scale
        out["quaternion"] = out["orientation"] = quaternion
        if cfg["orientation_repr"] == "discretized":                                   # :358-360
            index = self.orientation_grid.quat_to_index(quaternion.detach().cpu().numpy().astype(np.float64))
            out["orientation"] = torch.as_tensor(np.atleast_1d(index), device=dev, dtype=torch.long)
        return out

    def samples(self, out: Dict) -> List[Dict]:
        """Split a batch into the reference's per-sample dictionaries (valid samples only)."""
        keys = ("depth", "latent_shape", "position", "orientation", "quaternion", "scale")
        res = []
        for b in range(self.B):
            if not bool(out["valid"][b]):
                continue  # the reference redraws an empty sample (generated_dataset.py:222-235)
            d = {k: out[k][b] for k in keys}
            if "pointset" in out:
                d["pointset"] = out["pointset"][b]
            res.append(d)
        return res

    def __iter__(self):
        # (the consumer may use the same decoder on its own stream between two samples -- a validation decode, a VJP, a
        # FusedRenderAndCompare: the decode running ahead on the second stream has a scratch buffer of its own)
        self.prefetch_draws = True
        self.decode_ahead = True
        while True:
            for s in self.samples(self.generate()):
                yield s
