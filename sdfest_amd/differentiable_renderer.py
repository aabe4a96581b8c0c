"""PyTorch interface of the HIP depth renderer.

Host-side mirror of ``sdfest/differentiable_renderer/sdf_renderer.py``: same
``Camera`` (:31-133), same ``SDFRendererFunctionGPU`` (:267-357) and
``render_depth_gpu`` (:360-424) signatures, argument meaning and exceptions.
Tensors are only buffers here: the work happens in ``libsdfr_hip.so`` through
the C ABI of ``include/sdfr.h``.
"""
import math
import threading
from typing import Optional, Tuple

import torch

from . import _lib


class Camera:
    """Pinhole camera parameters (reference: sdf_renderer.py:31-133).

    ``pixel_center`` states where, inside a pixel, the principal point's
    coordinate system puts the pixel centre (0 = computer vision, 0.5 = graphics).
    """

    def __init__(self, width: int, height: int, fx: float, fy: float, cx: float, cy: float,
                 s: float = 0.0, pixel_center: float = 0.0):
        self.fx = fx
        self.fy = fy
        self.cx = cx
        self.cy = cy
        self.pixel_center = pixel_center
        self.s = s
        self.width = width
        self.height = height

    def get_pinhole_camera_parameters(self, pixel_center: float) -> Tuple:
        """(fx, fy, cx, cy, s) with the principal point re-expressed for `pixel_center`."""
        shift = pixel_center - self.pixel_center
        return self.fx, self.fy, self.cx + shift, self.cy + shift, self.s


# sdf_grad_mode values (include/sdfr.h)
SDF_GRAD_EXACT = 0
SDF_GRAD_CUDA_COMPAT = 1
SDF_GRAD_DETERMINISTIC = 0x100    # flag bit: integer accumulation of d/dSDF, bitwise reproducible
FIXED_QUANTUM_BITS = 40
BWD_HALF_GRID = 0x200             # flag bit: performance hint "every view is close" (views_are_close)
BWD_SMALL_TILES = 0x400           # flag bit: 32 x 8 backward tiles whatever the batch (pose sums independent of it)


def parse_sdf_grad_mode(value) -> int:
    """``config["sdf_grad_mode"]`` of the pipelines -> SDF_GRAD_EXACT / SDF_GRAD_CUDA_COMPAT: "exact" (default; the
    numpy twin's weights, simple_renderer.py:399-408) or "cuda_compat" (the weights the reference's GPU extension adds,
    sdf_renderer_cuda.cu:373-388 -- what produced the published system's latent trajectories); 0 / 1 are accepted too."""
    if value is None:
        return SDF_GRAD_EXACT
    if isinstance(value, str):
        names = {"exact": SDF_GRAD_EXACT, "cuda_compat": SDF_GRAD_CUDA_COMPAT}
        if value not in names:
            raise ValueError(f"sdf_grad_mode {value!r}: expected 'exact' or 'cuda_compat'")
        return names[value]
    if int(value) not in (SDF_GRAD_EXACT, SDF_GRAD_CUDA_COMPAT):
        raise ValueError(f"sdf_grad_mode {value!r}: expected 0 (exact) or 1 (cuda_compat)")
    return int(value)
VIEW_RECORD_FLOATS = 20           # SDFR_VIEW_RECORD_FLOATS: the sharded loop's exchange record of one view


def close_view_fraction(position, inv_scale, camera: "Camera", R: int) -> float:
    """The caller's side of the ``SDFR_BWD_HALF_GRID`` hint (include/sdfr.h): the share of the views whose object
    spans at least two pixels per voxel on the screen, sqrt(|fx fy|) * (2 / (R - 1)) / (inv_scale * |position|)
    >= 2 -- the views the backward gives 32 x 32-pixel tiles.  Takes host arrays (or tensors: a CUDA tensor costs a
    synchronising copy -- evaluate it where the poses are made, not per step)."""
    import numpy as np
    pos = position.detach().cpu().numpy() if isinstance(position, torch.Tensor) else np.asarray(position)
    isc = inv_scale.detach().cpu().numpy() if isinstance(inv_scale, torch.Tensor) else np.asarray(inv_scale)
    pos = pos.reshape(-1, 3).astype(np.float64)
    isc = isc.reshape(-1).astype(np.float64)
    fx, fy, _, _, _ = camera.get_pinhole_camera_parameters(0.5)
    dist = np.maximum(np.linalg.norm(pos, axis=1), 1e-20)
    r = np.sqrt(abs(fx * fy)) * (2.0 / (R - 1)) / (isc * dist)
    return float(np.mean(r >= 2.0)) if r.size else 0.0


def views_are_close(position, inv_scale, camera: "Camera", R: int, share: float = 0.9) -> bool:
    """True when at least `share` of the views are close (``close_view_fraction``): then the half grid pays --
    the few views that are not take their tiles two per workgroup."""
    return close_view_fraction(position, inv_scale, camera, R) >= share


_ws_lock = threading.Lock()
_ws_cache = {}


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Per (device, stream, thread) scratch buffer, grown on demand.  Forward runs on the caller's
    thread and backward on an autograd worker thread; keying by thread keeps them apart, keying by
    stream keeps two streams driven from one thread from sharing view records and partial sums
    without any ordering between them.  A buffer that is outgrown goes back to the caching
    allocator, which keeps it out of circulation until the work queued on ITS stream (the stream it
    was allocated on = the only one that used it) has finished."""
    key = (device.index, _stream(device), threading.get_ident())
    with _ws_lock:
        buf = _ws_cache.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
            _ws_cache[key] = buf
    return buf


def _check_input(t: torch.Tensor, name: str) -> None:
    # reference: CHECK_CUDA / CHECK_CONTIGUOUS (sdf_renderer.cpp:9-13) raise RuntimeError
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


try:    # the raw handle of the current stream without building a torch.cuda.Stream object around it: 0.3 instead of
    # 8 us per call, four calls per forward + backward pair of the drop-in path (tools/microbench/dropin_time.py)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:   # (a torch build without it)
    _raw_stream = None


def _stream(device: torch.device) -> int:
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def _check_inputs(*pairs) -> None:
    """``_check_input`` for several tensors: one combined test first (the drop-in path is host-bound: 2.4 us per
    tensor added up), the per-tensor messages only when something is wrong."""
    f32 = torch.float32
    for t, _ in pairs:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype is f32 and t.is_contiguous()):
            break
    else:
        return
    for t, n in pairs:
        _check_input(t, n)


def forward_raw(sdf, position, orientation, inv_scale, width, height, cx, cy, fx, fy, threshold):
    """Equivalent of ``sdf_renderer_cpp.forward`` (sdf_renderer.cpp:42-61), batched.

    sdf (R,R,R) shared by all views or (B,R,R,R); position (B,3); orientation (B,4);
    inv_scale (B,).  Returns depth (B,H,W).
    """
    _check_inputs((sdf, "sdf"), (position, "position"), (orientation, "orientation"), (inv_scale, "inv_scale"))
    B = position.shape[0]
    if position.shape != (B, 3) or orientation.shape != (B, 4) or inv_scale.shape != (B,):
        raise RuntimeError("expected position (B,3), orientation (B,4), inv_scale (B,)")
    R = sdf.shape[-1]
    if sdf.dim() == 3:
        stride = 0
    elif sdf.dim() == 4 and sdf.shape[0] == B:
        stride = R * R * R
    else:
        raise RuntimeError("sdf must be (R,R,R) or (B,R,R,R)")
    if tuple(sdf.shape[-3:]) != (R, R, R):
        raise RuntimeError("sdf must be cubic")
    dev = sdf.device
    depth = torch.empty((B, height, width), dtype=torch.float32, device=dev)
    L = _lib.lib()
    nbytes = L.sdfr_render_forward_workspace_bytes(R, B, width, height)
    ws = _workspace(dev, nbytes)
    rc = L.sdfr_render_forward(_ptr(sdf), R, stride, _ptr(position), _ptr(orientation),
                               _ptr(inv_scale), B, width, height, cx, cy, fx, fy, threshold,
                               _ptr(depth), _ptr(ws), ws.numel(), dev.index, _stream(dev))
    _lib.check(rc, "sdfr_render_forward")
    return depth


def backward_raw(grad_depth, depth, sdf, position, orientation, inv_scale, width, height, cx, cy,
                 fx, fy, sdf_grad_mode=0):
    """Equivalent of ``sdf_renderer_cpp.backward`` (sdf_renderer.cpp:63-86), batched.

    Returns (g_sdf like sdf, g_position (B,3), g_orientation (B,4), g_inv_scale (B,)).
    """
    _check_inputs((grad_depth, "grad_depth_image"), (depth, "depth_image"), (sdf, "sdf"), (position, "position"),
                  (orientation, "orientation"), (inv_scale, "inv_scale"))
    B = position.shape[0]
    R = sdf.shape[-1]
    per_view = sdf.dim() == 4
    stride = R * R * R if per_view else 0
    dev = sdf.device
    g_sdf = torch.empty_like(sdf)
    g_pos = torch.empty_like(position)
    g_quat = torch.empty_like(orientation)
    g_isc = torch.empty_like(inv_scale)
    L = _lib.lib()
    nbytes = L.sdfr_render_backward_workspace_bytes(R, B, width, height)
    ws = _workspace(dev, nbytes)
    rc = L.sdfr_render_backward(_ptr(grad_depth), _ptr(depth), _ptr(sdf), R, stride,
                                _ptr(position), _ptr(orientation), _ptr(inv_scale), B, width,
                                height, cx, cy, fx, fy, sdf_grad_mode, _ptr(g_sdf), stride,
                                _ptr(g_pos), _ptr(g_quat), _ptr(g_isc), _ptr(ws), ws.numel(),
                                dev.index, _stream(dev))
    _lib.check(rc, "sdfr_render_backward")
    return g_sdf, g_pos, g_quat, g_isc


def step_forward_raw(sdf, position, orientation, inv_scale, width, height, cx, cy, fx, fy, threshold):
    """``forward_raw`` as the first half of a step (``sdfr_render_step_forward``, include/sdfr.h): the same
    depth images, and the prologue launch also zero-fills the gradient volume and leaves the view records
    for ``step_backward_raw``.  The step owns its workspace (the shared scratch buffer may be re-used by
    other renders before this one's backward runs).  Returns (depth, state)."""
    _check_inputs((sdf, "sdf"), (position, "position"), (orientation, "orientation"), (inv_scale, "inv_scale"))
    B = position.shape[0]
    if position.shape != (B, 3) or orientation.shape != (B, 4) or inv_scale.shape != (B,):
        raise RuntimeError("expected position (B,3), orientation (B,4), inv_scale (B,)")
    R = sdf.shape[-1]
    if sdf.dim() == 3:
        stride = 0
    elif sdf.dim() == 4 and sdf.shape[0] == B:
        stride = R * R * R
    else:
        raise RuntimeError("sdf must be (R,R,R) or (B,R,R,R)")
    if tuple(sdf.shape[-3:]) != (R, R, R):
        raise RuntimeError("sdf must be cubic")
    dev = sdf.device
    depth = torch.empty((B, height, width), dtype=torch.float32, device=dev)
    g_sdf = torch.empty_like(sdf)
    L = _lib.lib()
    ws = torch.empty(max(L.sdfr_render_step_workspace_bytes(R, B, width, height), 256), dtype=torch.uint8,
                     device=dev)
    rc = L.sdfr_render_step_forward(_ptr(sdf), R, stride, _ptr(position), _ptr(orientation), _ptr(inv_scale),
                                    B, width, height, cx, cy, fx, fy, threshold, _ptr(depth), _ptr(g_sdf),
                                    stride, _ptr(ws), ws.numel(), dev.index, _stream(dev))
    _lib.check(rc, "sdfr_render_step_forward")
    return depth, (ws, g_sdf, _stream(dev))


def step_backward_raw(state, grad_depth, depth, sdf, position, orientation, inv_scale, width, height, cx, cy,
                      fx, fy, sdf_grad_mode=0):
    """Second half of the step begun by ``step_forward_raw`` (same return value as ``backward_raw``).  A state
    that has been used already (a second backward through a retained graph), or a backward on another stream,
    takes the stand-alone call."""
    dev = sdf.device
    if state is None or state[2] != _stream(dev):
        return backward_raw(grad_depth, depth, sdf, position, orientation, inv_scale, width, height, cx, cy,
                            fx, fy, sdf_grad_mode)
    _check_inputs((grad_depth, "grad_depth_image"), (depth, "depth_image"), (sdf, "sdf"))
    ws, g_sdf, _ = state
    B = position.shape[0]
    R = sdf.shape[-1]
    stride = R * R * R if sdf.dim() == 4 else 0
    g_pos = torch.empty_like(position)
    g_quat = torch.empty_like(orientation)
    g_isc = torch.empty_like(inv_scale)
    L = _lib.lib()
    rc = L.sdfr_render_step_backward(_ptr(grad_depth), _ptr(depth), _ptr(sdf), R, stride, B, width, height,
                                     cx, cy, fx, fy, sdf_grad_mode, _ptr(g_sdf), stride, _ptr(g_pos),
                                     _ptr(g_quat), _ptr(g_isc), _ptr(ws), ws.numel(), dev.index, _stream(dev))
    _lib.check(rc, "sdfr_render_step_backward")
    return g_sdf, g_pos, g_quat, g_isc


class _RenderBatch(torch.autograd.Function):
    """Batched renderer: B poses of one SDF (or of B SDFs) in one launch."""

    @staticmethod
    def forward(ctx, sdf, position, orientation, inv_scale, threshold, camera, sdf_grad_mode):
        fx, fy, cx, cy, _ = camera.get_pinhole_camera_parameters(0.5)
        sdf_c = sdf.detach().contiguous()
        pos = position.detach().contiguous()
        quat = orientation.detach().contiguous()
        isc = inv_scale.detach().contiguous()
        # forward + backward of the same views = one step (one launch less per call, tighter culling in the
        # backward); without a gradient to compute, the plain forward
        ctx.step = None
        if any(ctx.needs_input_grad[:4]):
            image, ctx.step = step_forward_raw(sdf_c, pos, quat, isc, camera.width, camera.height, cx, cy, fx,
                                               fy, threshold)
        else:
            image = forward_raw(sdf_c, pos, quat, isc, camera.width, camera.height, cx, cy, fx, fy,
                                threshold)
        ctx.save_for_backward(image, sdf_c, pos, quat, isc)
        ctx.cam = (camera.width, camera.height, cx, cy, fx, fy)
        ctx.sdf_grad_mode = sdf_grad_mode
        return image

    @staticmethod
    def backward(ctx, grad_depth_image):
        image, sdf, pos, quat, isc = ctx.saved_tensors
        w, h, cx, cy, fx, fy = ctx.cam
        step, ctx.step = ctx.step, None
        g_sdf, g_p, g_q, g_is = step_backward_raw(step, grad_depth_image.contiguous(), image, sdf, pos, quat,
                                                  isc, w, h, cx, cy, fx, fy, ctx.sdf_grad_mode)
        return g_sdf, g_p, g_q, g_is, None, None, None


class SDFRendererFunctionGPU(torch.autograd.Function):
    """Single-view renderer function; signature of sdf_renderer.py:267-357."""

    @staticmethod
    def forward(ctx, sdf: torch.Tensor, position: torch.Tensor, orientation: torch.Tensor,
                inv_scale: torch.Tensor, threshold: Optional[float] = 0.0,
                camera: Optional[Camera] = None, sdf_grad_mode: Optional[int] = None) -> torch.Tensor:
        # sdf_grad_mode (not in the reference's signature): None = the module-wide render_depth_gpu.sdf_grad_mode
        _check_inputs((sdf, "sdf"), (position, "position"), (orientation, "orientation"), (inv_scale, "inv_scale"))
        ctx.sdf_grad_mode = sdf_grad_mode
        fx, fy, cx, cy, _ = camera.get_pinhole_camera_parameters(0.5)
        # the pair forward / backward of one view is one step (sdfr_render_step_*: 4 launches instead of 5)
        ctx.step = None
        args = (sdf, position.reshape(1, 3), orientation.reshape(1, 4), inv_scale.reshape(1), camera.width,
                camera.height, cx, cy, fx, fy, threshold)
        if any(ctx.needs_input_grad[:4]):
            image, ctx.step = step_forward_raw(*args)
            image = image[0]
        else:
            image = forward_raw(*args)[0]
        ctx.save_for_backward(image, sdf, position, orientation, inv_scale)
        ctx.cam = (camera.width, camera.height, cx, cy, fx, fy)
        return image

    @staticmethod
    def backward(ctx, grad_depth_image: torch.Tensor):
        image, sdf, position, orientation, inv_scale = ctx.saved_tensors
        w, h, cx, cy, fx, fy = ctx.cam
        step, ctx.step = ctx.step, None
        g_sdf, g_p, g_q, g_is = step_backward_raw(
            step, grad_depth_image.contiguous().reshape(1, h, w), image.reshape(1, h, w), sdf,
            position.reshape(1, 3), orientation.reshape(1, 4), inv_scale.reshape(1), w, h, cx, cy,
            fx, fy, render_depth_gpu.sdf_grad_mode if ctx.sdf_grad_mode is None else ctx.sdf_grad_mode)
        # gradients come back in the shape of the inputs ((4,) or (1,4); () or (1,))
        return (g_sdf, g_p.reshape(position.shape), g_q.reshape(orientation.shape),
                g_is.reshape(inv_scale.shape), None, None, None)


def render_depth_gpu(sdf: torch.Tensor, position: torch.Tensor, orientation: torch.Tensor,
                     inv_scale: torch.Tensor, width: Optional[int] = None,
                     height: Optional[int] = None, fov_deg: Optional[float] = None,
                     threshold: Optional[float] = 0.0, camera: Optional[Camera] = None,
                     sdf_grad_mode: Optional[int] = None):
    """Render a depth image of a 7-DOF discrete SDF (drop-in for sdf_renderer.py:360-424).

    The SDF pose is given in the camera frame under the OpenGL convention; the first image
    row is up.  Give either ``camera`` or ``width``+``height``+``fov_deg`` (horizontal fov,
    square pixels).  Differentiable w.r.t. sdf, position, orientation and inv_scale.
    sdf_grad_mode (beyond the reference's signature): which d depth / d sdf weights the backward uses for THIS call
    -- ``SDF_GRAD_EXACT`` (the numpy twin's, simple_renderer.py:399-408) or ``SDF_GRAD_CUDA_COMPAT`` (what the
    reference's GPU extension really adds, sdf_renderer_cuda.cu:373-388); None: ``render_depth_gpu.sdf_grad_mode``.
    """
    if None not in [width, height, fov_deg] and camera is not None:
        raise ValueError("Either width+height+fov_dev or camera must be provided.")
    if camera is None:
        f = width / math.tan(fov_deg * math.pi / 180.0 / 2.0) / 2
        camera = Camera(width, height, f, f, width / 2, height / 2, pixel_center=0.5)
    return SDFRendererFunctionGPU.apply(sdf, position, orientation, inv_scale, threshold, camera, sdf_grad_mode)


# d depth / d sdf weights: 0 = exact (numpy twin), 1 = the CUDA kernel's permutation
render_depth_gpu.sdf_grad_mode = 0


def render_depth_batch(sdf: torch.Tensor, positions: torch.Tensor, orientations: torch.Tensor,
                       inv_scales: torch.Tensor, threshold: float, camera: Camera,
                       sdf_grad_mode: int = 0) -> torch.Tensor:
    """B views in one launch: view b equals render_depth_gpu(sdf, positions[b], ...).

    The reference loops over views in Python (estimation/simple_setup.py:420-446); this is
    the same arithmetic with the loop moved onto the GPU.  sdf: (R,R,R) shared or (B,R,R,R).
    """
    return _RenderBatch.apply(sdf, positions, orientations, inv_scales, threshold, camera,
                              sdf_grad_mode)


class _RenderL1Batch(torch.autograd.Function):
    """Batched render with the masked depth-L1 folded in: returns the per-view loss
    and the depth images; the gradient image never exists."""

    @staticmethod
    def forward(ctx, sdf, position, orientation, inv_scale, target, threshold, camera, sdf_grad_mode):
        fx, fy, cx, cy, _ = camera.get_pinhole_camera_parameters(0.5)
        sdf_c = sdf.detach().contiguous()
        pos = position.detach().contiguous()
        quat = orientation.detach().contiguous()
        isc = inv_scale.detach().contiguous()
        tgt = target.detach().contiguous()
        for t, n in ((sdf_c, "sdf"), (pos, "position"), (quat, "orientation"), (isc, "inv_scale"),
                     (tgt, "target")):
            _check_input(t, n)
        B, R = pos.shape[0], sdf_c.shape[-1]
        W, H = camera.width, camera.height
        if tuple(tgt.shape) != (B, H, W):
            raise RuntimeError(f"target must have shape {(B, H, W)}, got {tuple(tgt.shape)}")
        stride = R * R * R if sdf_c.dim() == 4 else 0
        dev = sdf_c.device
        depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        stats = torch.empty((B, 2), dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws = _workspace(dev, L.sdfr_render_forward_l1_workspace_bytes(R, B, W, H))
        rc = L.sdfr_render_forward_l1(_ptr(sdf_c), R, stride, _ptr(pos), _ptr(quat), _ptr(isc), B, W, H,
                                      cx, cy, fx, fy, threshold, _ptr(tgt), _ptr(depth), _ptr(loss),
                                      _ptr(stats), _ptr(ws), ws.numel(), dev.index, _stream(dev))
        _lib.check(rc, "sdfr_render_forward_l1")
        ctx.save_for_backward(depth, stats, tgt, sdf_c, pos, quat, isc)
        ctx.cam = (W, H, cx, cy, fx, fy)
        ctx.sdf_grad_mode = sdf_grad_mode
        ctx.mark_non_differentiable(depth)
        return loss, depth

    @staticmethod
    def backward(ctx, grad_loss, _grad_depth_unused):
        depth, stats, tgt, sdf, pos, quat, isc = ctx.saved_tensors
        W, H, cx, cy, fx, fy = ctx.cam
        B, R = pos.shape[0], sdf.shape[-1]
        stride = R * R * R if sdf.dim() == 4 else 0
        dev = sdf.device
        g_sdf = torch.empty_like(sdf)
        g_pos = torch.empty_like(pos)
        g_quat = torch.empty_like(quat)
        g_isc = torch.empty_like(isc)
        gl = grad_loss.to(torch.float32).contiguous()
        L = _lib.lib()
        ws = _workspace(dev, L.sdfr_render_backward_workspace_bytes(R, B, W, H))
        rc = L.sdfr_render_backward_l1(_ptr(gl), 1.0, _ptr(stats), _ptr(tgt), _ptr(depth), _ptr(sdf), R,
                                       stride, _ptr(pos), _ptr(quat), _ptr(isc), B, W, H, cx, cy, fx,
                                       fy, ctx.sdf_grad_mode, _ptr(g_sdf), stride, _ptr(g_pos),
                                       _ptr(g_quat), _ptr(g_isc), _ptr(ws), ws.numel(), dev.index,
                                       _stream(dev))
        _lib.check(rc, "sdfr_render_backward_l1")
        return g_sdf, g_pos, g_quat, g_isc, None, None, None, None


def render_depth_l1_batch(sdf: torch.Tensor, positions: torch.Tensor, orientations: torch.Tensor,
                          inv_scales: torch.Tensor, targets: torch.Tensor, threshold: float,
                          camera: Camera, sdf_grad_mode: int = 0):
    """``render_depth_batch`` and the depth term of ``SDFPipeline._compute_view_losses``
    (estimation/simple_setup.py:129-135) in one pass over the images.

    Returns ``(loss, depth)``: ``loss[b] = mean |depth[b] - targets[b]|`` over the pixels where both
    are > 0 (NaN if there is none), differentiable w.r.t. sdf, positions, orientations and
    inv_scales; ``depth`` (B,H,W) is returned for inspection and carries no gradient.
    """
    return _RenderL1Batch.apply(sdf, positions, orientations, inv_scales, targets, threshold, camera,
                                sdf_grad_mode)


class BatchRenderPlan:
    """Pre-allocated buffers for repeated forward+backward of B views (no per-call allocation).

    This is how a production loop drives the C ABI: every output and the workspace are
    allocated once, each step is a fixed sequence of launches on one stream, so the step can
    be captured into a hipGraph (``capture()``).
    """

    def __init__(self, R: int, B: int, camera: Camera, device="cuda", per_view_sdf: bool = False,
                 sdf_grad_mode: int = 0, grad_volumes: int = 2, close_views="auto",
                 grad_tail_words: int = 0):
        """close_views: the ``SDFR_BWD_HALF_GRID`` hint of a step's backward -- half the workgroups when (nearly) all
        views are close (``views_are_close``); same results either way, views that are not close are slower with it.
        True / False: the caller knows.  "auto" (default): nobody looks at the poses on the host -- the step's
        forward counts its close views on the device (``sdfr_render_step_forward_counted``) into a pinned host word,
        and every backward reads what the word holds at that moment: the count of some earlier step (stale by one or
        a few: the hint cannot change results), none before the first forward has finished -> the full grid.
        grad_tail_words: every gradient volume of the ring is followed by this many float32 words of the caller's
        (``g_tail``): volume and tail are one contiguous bucket (``g_bucket``), e.g. ONE all-reduce over the ranks of a
        sharded batch carries d/dSDF and whatever small per-view results ride along."""
        if grad_volumes < 2:
            raise ValueError("grad_volumes must be >= 2")
        if grad_tail_words < 0 or (grad_tail_words and per_view_sdf):
            raise ValueError("grad_tail_words needs the shared gradient volume")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.R, self.B, self.camera = R, B, camera
        self.fx, self.fy, self.cx, self.cy, _ = camera.get_pinhole_camera_parameters(0.5)
        self.W, self.H = camera.width, camera.height
        self.sdf_stride = R * R * R if per_view_sdf else 0
        if close_views not in (True, False, "auto"):
            raise ValueError("close_views must be True, False or 'auto'")
        self._close_auto = close_views == "auto"
        self.sdf_grad_mode = sdf_grad_mode | (BWD_HALF_GRID if close_views is True else 0)
        # pinned: the device stores into it (one 64-bit word per forward), the host reads it without synchronising
        self._close_word = torch.zeros(1, dtype=torch.int64).pin_memory() if self._close_auto else None
        self.half_grid_steps = 0     # step backwards that went out with the hint ("auto": decided from the word)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.depth = torch.empty((B, self.H, self.W), **f32)
        self.g_sdf = torch.empty((B, R, R, R) if per_view_sdf else (R, R, R), **f32)
        self.g_pos = torch.empty((B, 3), **f32)
        self.g_quat = torch.empty((B, 4), **f32)
        self.g_inv_scale = torch.empty((B,), **f32)
        L = _lib.lib()
        nbytes = max(L.sdfr_render_forward_l1_workspace_bytes(R, B, self.W, self.H),
                     L.sdfr_render_backward_workspace_bytes(R, B, self.W, self.H),
                     L.sdfr_render_step_workspace_bytes(R, B, self.W, self.H), 256)
        # a step (forward(..., prepare_backward=True) + backward) zero-fills the NEXT gradient volume in the
        # forward's prologue: `grad_volumes` volumes take turns, so the g_sdf of the previous grad_volumes - 1
        # steps (e.g. still being all-reduced) are not touched by the next forward
        # (one contiguous buffer, ``grad_ring``: several steps' volumes can then go into ONE all-reduce)
        if grad_tail_words:
            # 16-byte multiple per bucket: every volume of the ring stays aligned for the kernels' vector stores
            self._bucket_ring = torch.zeros((grad_volumes, R * R * R + (grad_tail_words + 3) // 4 * 4), **f32)
            self.grad_ring = None
            self._g_sdf_ring = [b[:R * R * R].view(R, R, R) for b in self._bucket_ring.unbind(0)]
        else:
            self._bucket_ring = None
            self.grad_ring = torch.empty((grad_volumes,) + tuple(self.g_sdf.shape), **f32)
            self._g_sdf_ring = list(self.grad_ring.unbind(0))
        self.g_sdf = self._g_sdf_ring[-1]
        self._g_sdf_next = 0
        self._g_sdf_index = grad_volumes - 1   # which volume of the ring self.g_sdf is
        self._step = None   # what the last forward prepared: (tensor ids / versions, depth tensor)
        self._fixed_layout = None   # deterministic mode: which workspace layout holds the last int64 volume
        self._step_l1 = None        # the volume a forward_l1(prepare_backward=True) zero-filled
        self.partials_offset = 0    # where the last backward_l1_pc left its tile partials (sdfr_loop_tail)
        self._g_depth = None        # step_fused_l1_pc: the views' unscaled depth-term volumes
        self.loss = torch.empty((B,), **f32)
        self.loss_stats = torch.empty((B, 2), **f32)
        # zero-filled once: the sync region's counter of prologue fall-backs (include/sdfr.h) then counts from 0
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self._sync_offset = L.sdfr_render_sync_offset(B)
        self._L = L
        self._shape_sdf = (B, R, R, R) if per_view_sdf else (R, R, R)
        self._shape_pos, self._shape_quat, self._shape_isc = (B, 3), (B, 4), (B,)
        self._shape_img = (B, self.H, self.W)

    def _check(self, sdf, pos, quat, inv_scale, **images) -> None:
        """The C ABI takes raw device pointers: a tensor of another dtype, shape, device or layout
        would be read as garbage or out of bounds on the GPU.  Same exception type as the
        reference's CHECK_CUDA / CHECK_CONTIGUOUS (sdf_renderer.cpp:9-13).  One combined test on the
        hot path (a single-view call is ~14 us of GPU work: per-tensor helper calls and dictionaries
        made the eager loop host-bound); the slow path only names what is wrong."""
        dev, f32 = self.device, torch.float32
        ok = (sdf.device == dev and sdf.dtype is f32 and sdf.shape == self._shape_sdf and sdf.is_contiguous()
              and pos.device == dev and pos.dtype is f32 and pos.shape == self._shape_pos and pos.is_contiguous()
              and quat.device == dev and quat.dtype is f32 and quat.shape == self._shape_quat and quat.is_contiguous()
              and inv_scale.device == dev and inv_scale.dtype is f32 and inv_scale.shape == self._shape_isc
              and inv_scale.is_contiguous())
        for t in images.values():
            ok = ok and (t.device == dev and t.dtype is f32 and t.shape == self._shape_img and t.is_contiguous())
        if ok:
            return
        want = {"sdf": self._shape_sdf, "position": self._shape_pos, "orientation": self._shape_quat,
                "inv_scale": self._shape_isc}
        got = {"sdf": sdf, "position": pos, "orientation": quat, "inv_scale": inv_scale}
        for name, t in images.items():
            want[name] = self._shape_img
            got[name] = t
        for name, t in got.items():
            _check_input(t, name)
            if t.device != self.device:
                raise RuntimeError(f"{name} is on {t.device}, the plan on {self.device}")
            if tuple(t.shape) != want[name]:
                raise RuntimeError(f"{name} must have shape {want[name]}, got {tuple(t.shape)}")

    def ring_reset(self) -> None:
        """The next step writes ``grad_ring[0]`` again (a caller that exchanges halves of the ring aligns them)."""
        self._g_sdf_next = 0
        self._step = None
        self._step_l1 = None

    def close_views_seen(self):
        """(forwards counted so far, close views of the latest counted one) as the pinned word holds them right now
        -- no synchronisation: the forward whose count this is finished some time ago ("auto" plans only)."""
        if self._close_word is None:
            return 0, 0
        v = int(self._close_word[0])
        return (v >> 32) & 0xffffffff, v & 0xffffffff

    def select_volume(self, index: int) -> None:
        """Make ring volume `index` the one the next stand-alone backward writes (``self.g_sdf``); a step's backward
        writes the volume its forward prepared instead."""
        self.g_sdf = self._g_sdf_ring[index]
        self._g_sdf_index = index

    @property
    def g_bucket(self) -> torch.Tensor:
        """The flat bucket [R^3 words of ``g_sdf`` | tail] of the volume the last backward wrote (``grad_tail_words``)."""
        if self._bucket_ring is None:
            raise RuntimeError("the plan was made without grad_tail_words")
        return self._bucket_ring[self._g_sdf_index]

    @property
    def g_tail(self) -> torch.Tensor:
        return self.g_bucket[self.R ** 3:]

    def next_bucket(self) -> torch.Tensor:
        """The bucket the NEXT step (``forward*(prepare_backward=True)`` + its backward) will write."""
        if self._bucket_ring is None:
            raise RuntimeError("the plan was made without grad_tail_words")
        return self._bucket_ring[self._g_sdf_next]

    def prologue_fallbacks(self) -> int:
        """How many view set-ups of this plan's forwards so far had to do without the grid's plane minima
        (``include/sdfr.h``, sync region word 1; synchronises).  0 unless the prologue launch was serialised."""
        o = self._sync_offset + 4
        return int(self.workspace[o:o + 4].view(torch.int32).item())

    @staticmethod
    def _key(sdf, pos, quat, inv_scale):
        return (sdf.data_ptr(), sdf._version, pos.data_ptr(), pos._version, quat.data_ptr(), quat._version,
                inv_scale.data_ptr(), inv_scale._version)

    def forward(self, sdf, pos, quat, inv_scale, threshold: float, out: Optional[torch.Tensor] = None,
                prepare_backward: bool = False) -> torch.Tensor:
        """Render into the plan's depth buffer, or into ``out`` (B,H,W float32 contiguous on the plan's
        device) when the caller keeps the images (the plan's buffer is overwritten by the next call).

        ``prepare_backward``: this forward and the next ``backward`` of the same tensors form one step
        (``sdfr_render_step_forward`` / ``sdfr_render_step_backward``, include/sdfr.h): the forward's
        prologue also zero-fills the gradient volume and the backward starts from the forward's view
        records -- one launch less, fewer live tiles, same results.  ``self.g_sdf`` then cycles
        through the plan's ``grad_volumes`` volumes from step to step."""
        self._check(sdf, pos, quat, inv_scale)
        dst = self.depth if out is None else out
        if out is not None and (out.shape != self.depth.shape or out.dtype != torch.float32
                                or not out.is_contiguous() or out.device != self.depth.device):
            raise RuntimeError("out must be a contiguous float32 tensor of shape (B, H, W) on the plan's device")
        self._step_l1 = None   # a forward_l1(prepare_backward=True) whose backward never came: its view records are gone
        if prepare_backward:
            nxt = self._g_sdf_ring[self._g_sdf_next]
            rc = self._L.sdfr_render_step_forward_counted(
                sdf.data_ptr(), self.R, self.sdf_stride, pos.data_ptr(), quat.data_ptr(),
                inv_scale.data_ptr(), self.B, self.W, self.H, self.cx, self.cy, self.fx, self.fy,
                threshold, dst.data_ptr(), nxt.data_ptr(), self.sdf_stride, self.workspace.data_ptr(),
                self.workspace.numel(), self._close_word.data_ptr() if self._close_auto else None,
                self.device.index, _stream(self.device))
            _lib.check(rc, "sdfr_render_step_forward")
            self._step = (self._key(sdf, pos, quat, inv_scale), dst, nxt)
            return dst
        self._step = None
        rc = self._L.sdfr_render_forward(
            sdf.data_ptr(), self.R, self.sdf_stride, pos.data_ptr(), quat.data_ptr(),
            inv_scale.data_ptr(), self.B, self.W, self.H, self.cx, self.cy, self.fx, self.fy,
            threshold, dst.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(),
            self.device.index, _stream(self.device))
        _lib.check(rc, "sdfr_render_forward")
        return dst

    def backward(self, grad_depth, sdf, pos, quat, inv_scale, defer_pose: bool = False):
        """``defer_pose``: leave the pose gradients as tile partials in ``self.workspace`` for
        ``sdfr_views_to_pose_grad_deferred`` (include/sdfr.h); g_pos / g_quat / g_inv_scale are then not written.

        After ``forward(..., prepare_backward=True)`` on the same (unmodified) tensors this is the step's
        backward; any other call in between, or different inputs, falls back to the stand-alone backward.
        "Unmodified" is judged by address and torch's version counter: a tensor updated through its raw pointer
        between the two halves (``sdfr_adam_step``, ``sdfr_pose_to_views``) looks unmodified, and the step's
        backward would then use the view records of the forward's poses -- do not update poses or the grid
        between the halves of a step."""
        self._check(sdf, pos, quat, inv_scale, grad_depth=grad_depth)
        step, self._step = self._step, None
        self._step_l1 = None
        if step is not None and not defer_pose and step[0] == self._key(sdf, pos, quat, inv_scale):
            _, depth, g_sdf = step
            mode = self.sdf_grad_mode
            if self._close_auto:
                seen, close = self.close_views_seen()
                if seen and close >= 0.9 * self.B:
                    mode |= BWD_HALF_GRID
            self.half_grid_steps += 1 if mode & BWD_HALF_GRID else 0
            rc = self._L.sdfr_render_step_backward(
                grad_depth.data_ptr(), depth.data_ptr(), sdf.data_ptr(), self.R, self.sdf_stride,
                self.B, self.W, self.H, self.cx, self.cy, self.fx, self.fy, mode,
                g_sdf.data_ptr(), self.sdf_stride, self.g_pos.data_ptr(), self.g_quat.data_ptr(),
                self.g_inv_scale.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(),
                self.device.index, _stream(self.device))
            _lib.check(rc, "sdfr_render_step_backward")
            self.g_sdf = g_sdf
            self._g_sdf_index = self._g_sdf_next
            self._g_sdf_next = (self._g_sdf_next + 1) % len(self._g_sdf_ring)
            self._fixed_layout = 1
            return self.g_sdf, self.g_pos, self.g_quat, self.g_inv_scale
        depth = self.depth
        if step is not None:
            # a prepared step whose backward cannot run as such (other tensors, or deferred pose sums): the
            # stand-alone backward, but into the volume the forward prepared -- the previous step's volume may
            # still be in an asynchronous all-reduce -- and of the images that forward rendered
            _, depth, self.g_sdf = step
            self._g_sdf_index = self._g_sdf_next
            self._g_sdf_next = (self._g_sdf_next + 1) % len(self._g_sdf_ring)
        rc = self._L.sdfr_render_backward(
            grad_depth.data_ptr(), depth.data_ptr(), sdf.data_ptr(), self.R, self.sdf_stride,
            pos.data_ptr(), quat.data_ptr(), inv_scale.data_ptr(), self.B, self.W, self.H,
            self.cx, self.cy, self.fx, self.fy, self.sdf_grad_mode, self.g_sdf.data_ptr(),
            self.sdf_stride, None if defer_pose else self.g_pos.data_ptr(),
            None if defer_pose else self.g_quat.data_ptr(),
            None if defer_pose else self.g_inv_scale.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(),
            self.device.index, _stream(self.device))
        _lib.check(rc, "sdfr_render_backward")
        self._fixed_layout = 0
        return self.g_sdf, self.g_pos, self.g_quat, self.g_inv_scale

    def g_sdf_fixed(self) -> torch.Tensor:
        """After a backward in the deterministic mode (``sdf_grad_mode | SDF_GRAD_DETERMINISTIC``): the int64
        (R,R,R) volume in the workspace that ``g_sdf`` was converted from, ``g_sdf = volume * 2**-40``.  The
        ranks of a sharded batch add THESE (integer addition: any order, any grouping gives the same bits) and
        convert afterwards: ``sdfest_amd.parallel.allreduce_fixed_gradients``."""
        if not (self.sdf_grad_mode & SDF_GRAD_DETERMINISTIC) or self._fixed_layout is None:
            raise RuntimeError("g_sdf_fixed() needs a backward in the deterministic mode")
        off = self._L.sdfr_render_fixed_volume_offset(self.R, self.B, self.W, self.H, self._fixed_layout)
        n = self.R ** 3 * 8
        return self.workspace[off:off + n].view(torch.int64).view(self.R, self.R, self.R)

    def forward_l1(self, sdf, pos, quat, inv_scale, threshold: float, target, prepare_backward: bool = False,
                   defer_loss: bool = False):
        """forward + masked depth-L1 against ``target`` (B,H,W): returns (depth, loss (B,)).

        ``prepare_backward``: first half of a step whose second half is ``backward_l1`` or ``backward_l1_pc``
        (``sdfr_render_step_forward_l1`` / ``sdfr_render_step_backward_l1[_pc]``): the gradient volume is zero-filled
        and the view records are left for the backward, which then has no prologue launch.
        ``defer_loss`` (with prepare_backward): the forward does not reduce its per-tile (sum, count) records -- one
        dependent launch less between the two image kernels -- and ``self.loss`` / ``self.loss_stats`` are written
        by the step's backward instead (same bits; include/sdfr.h, "DEFERRED LOSS").  The caller promises that the
        step's backward follows, on the same tensors."""
        self._step = None   # the workspace is about to be re-used
        self._check(sdf, pos, quat, inv_scale, target=target)
        if defer_loss and not prepare_backward:
            raise ValueError("defer_loss goes with prepare_backward")
        if prepare_backward:
            nxt = self._g_sdf_ring[self._g_sdf_next]
            rc = self._L.sdfr_render_step_forward_l1(
                sdf.data_ptr(), self.R, self.sdf_stride, pos.data_ptr(), quat.data_ptr(), inv_scale.data_ptr(),
                self.B, self.W, self.H, self.cx, self.cy, self.fx, self.fy, threshold, target.data_ptr(),
                self.depth.data_ptr(), None if defer_loss else self.loss.data_ptr(),
                None if defer_loss else self.loss_stats.data_ptr(), nxt.data_ptr(),
                self.sdf_stride, self.workspace.data_ptr(), self.workspace.numel(),
                self._close_word.data_ptr() if self._close_auto else None, self.device.index, _stream(self.device))
            _lib.check(rc, "sdfr_render_step_forward_l1")
            self._step_l1 = (self._key(sdf, pos, quat, inv_scale), nxt, bool(defer_loss))
            return self.depth, self.loss
        self._step_l1 = None
        rc = self._L.sdfr_render_forward_l1(
            sdf.data_ptr(), self.R, self.sdf_stride, pos.data_ptr(), quat.data_ptr(),
            inv_scale.data_ptr(), self.B, self.W, self.H, self.cx, self.cy, self.fx, self.fy,
            threshold, target.data_ptr(), self.depth.data_ptr(), self.loss.data_ptr(),
            self.loss_stats.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(),
            self.device.index, _stream(self.device))
        _lib.check(rc, "sdfr_render_forward_l1")
        return self.depth, self.loss

    def backward_l1(self, target, sdf, pos, quat, inv_scale, weight: float = 1.0, loss_grad=None,
                    defer_pose: bool = False):
        """gradients of sum_b weight * loss_grad[b] * loss[b] (after ``forward_l1``); ``defer_pose`` as in
        ``backward``."""
        self._step = None   # the workspace is about to be re-used
        prepared, self._step_l1 = self._step_l1, None
        self._check(sdf, pos, quat, inv_scale, target=target)
        if loss_grad is not None:
            _check_input(loss_grad, "loss_grad")
            if tuple(loss_grad.shape) != (self.B,) or loss_grad.device != self.device:
                raise RuntimeError(f"loss_grad must have shape ({self.B},) on {self.device}")
        if prepared is not None:
            # second half of a step begun by forward_l1(prepare_backward=True): into the volume that forward zero-filled;
            # as the step's backward (no prologue, the forward's records) only for the very tensors the forward saw
            key, g_sdf, deferred = prepared
            self.g_sdf = g_sdf
            self._g_sdf_index = self._g_sdf_next
            self._g_sdf_next = (self._g_sdf_next + 1) % len(self._g_sdf_ring)
            if key == self._key(sdf, pos, quat, inv_scale):
                mode = self.sdf_grad_mode
                if self._close_auto:      # the half-grid hint, as in ``backward``
                    seen, close = self.close_views_seen()
                    if seen and close >= 0.9 * self.B:
                        mode |= BWD_HALF_GRID
                self.half_grid_steps += 1 if mode & BWD_HALF_GRID else 0
                rc = self._L.sdfr_render_step_backward_l1(
                    loss_grad.data_ptr() if loss_grad is not None else None, weight,
                    None if deferred else self.loss_stats.data_ptr(), target.data_ptr(), self.depth.data_ptr(),
                    sdf.data_ptr(), self.R, self.sdf_stride, self.B, self.W, self.H, self.cx, self.cy, self.fx, self.fy,
                    mode, self.g_sdf.data_ptr(), self.sdf_stride,
                    None if defer_pose else self.g_pos.data_ptr(), None if defer_pose else self.g_quat.data_ptr(),
                    None if defer_pose else self.g_inv_scale.data_ptr(), self.workspace.data_ptr(),
                    self.workspace.numel(), self.loss.data_ptr() if deferred else None,
                    self.loss_stats.data_ptr() if deferred else None, self.device.index, _stream(self.device))
                _lib.check(rc, "sdfr_render_step_backward_l1")
                self.partials_offset = self._L.sdfr_render_partials_offset(self.R, self.B, self.W, self.H, 1)
                self._fixed_layout = 1
                return self.g_sdf, self.g_pos, self.g_quat, self.g_inv_scale
            if deferred:
                raise RuntimeError("forward_l1(defer_loss=True) must be followed by the step's backward on the same "
                                   "tensors: its loss statistics were left to that launch")
        rc = self._L.sdfr_render_backward_l1(
            loss_grad.data_ptr() if loss_grad is not None else None, weight,
            self.loss_stats.data_ptr(), target.data_ptr(), self.depth.data_ptr(), sdf.data_ptr(),
            self.R, self.sdf_stride, pos.data_ptr(), quat.data_ptr(), inv_scale.data_ptr(), self.B,
            self.W, self.H, self.cx, self.cy, self.fx, self.fy, self.sdf_grad_mode,
            self.g_sdf.data_ptr(), self.sdf_stride, None if defer_pose else self.g_pos.data_ptr(),
            None if defer_pose else self.g_quat.data_ptr(),
            None if defer_pose else self.g_inv_scale.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(),
            self.device.index, _stream(self.device))
        _lib.check(rc, "sdfr_render_backward_l1")
        return self.g_sdf, self.g_pos, self.g_quat, self.g_inv_scale

    def step_fused_l1_pc(self, sdf, pos, quat, inv_scale, scale, threshold: float, target, points, offsets,
                         max_view_points: int, pc_workspace, pc_weight: float = 1.0, g_sdf=None):
        """``forward_l1(prepare_backward=True, defer_loss=True)`` and ``backward_l1_pc`` of at most 8 views as ONE launch
        (``sdfr_render_step_fused_l1_pc``): a tile runs the backward of its hit pixels while their depths are still
        in registers.  A view's overlap count is not known inside the launch, so the depth term is left UNSCALED --
        d/dSDF in the view's own volume (``g_depth`` (B,R,R,R), allocated at the first call that wants it), the pose sums
        in the tile partials -- beside the count (``view_count``); ``g_sdf`` (None: a loop that does not optimise the shape:
        no d/dSDF at all) receives the point-cloud term of all views.
        Nothing is zero-filled here: the consumer clears what it has read (``sdfr_decoder_backward_latent_deferred_scaled``,
        ``sdfr_loop_tail_fused``).  Returns the depth images."""
        self._step = None
        self._step_l1 = None
        self._check(sdf, pos, quat, inv_scale, target=target)
        rc = self._L.sdfr_render_step_fused_l1_pc(
            sdf.data_ptr(), self.R, self.sdf_stride, pos.data_ptr(), quat.data_ptr(), inv_scale.data_ptr(),
            scale.data_ptr(), self.B, self.W, self.H, self.cx, self.cy, self.fx, self.fy, threshold, target.data_ptr(),
            self.depth.data_ptr(), self.sdf_grad_mode, g_sdf.data_ptr() if g_sdf is not None else None,
            self.g_depth.data_ptr() if g_sdf is not None else None,
            self.workspace.data_ptr(), self.workspace.numel(), pc_weight, points.data_ptr(),
            offsets.data_ptr() if offsets is not None else None, max_view_points, pc_workspace.data_ptr(),
            pc_workspace.numel(), self.device.index, _stream(self.device))
        _lib.check(rc, "sdfr_render_step_fused_l1_pc")
        self.partials_offset = self._L.sdfr_render_partials_offset(self.R, self.B, self.W, self.H, 1)
        return self.depth

    @property
    def g_depth(self) -> torch.Tensor:
        """the views' unscaled d/dSDF of the depth term that ``step_fused_l1_pc`` accumulates, (B,R,R,R) float32"""
        if self._g_depth is None:
            self._g_depth = torch.zeros((self.B, self.R, self.R, self.R), dtype=torch.float32, device=self.device)
        return self._g_depth

    @property
    def view_count(self) -> torch.Tensor:
        """the views' overlap counts that ``step_fused_l1_pc`` adds up, (B,) float32 in the workspace"""
        off = self._L.sdfr_render_fused_view_count_offset(self.B, self.H)
        return self.workspace[off:off + 4 * self.B].view(torch.float32)

    def backward_l1_pc(self, target, sdf, pos, quat, inv_scale, scale, points, offsets, max_view_points: int,
                       pc_workspace, weight: float = 1.0, pc_weight: float = 1.0, loss_grad=None):
        """``backward_l1`` and the sampler's L1 backward (``sdfr_pc_l1_backward_accumulate``) in one launch
        (``sdfr_render_backward_l1_pc``): g_sdf holds both terms, the pose gradients stay deferred in
        ``self.workspace`` and ``pc_workspace`` for ``sdfr_views_to_pose_grad_deferred``."""
        self._step = None   # the workspace is about to be re-used
        self._check(sdf, pos, quat, inv_scale, target=target)
        dev = self.device
        if not (scale.device == dev and scale.dtype is torch.float32 and scale.shape == self._shape_isc
                and scale.is_contiguous()):
            raise RuntimeError(f"scale must be a contiguous float32 tensor of shape {self._shape_isc} on {dev}")
        if not (points.device == dev and points.dtype is torch.float32 and points.dim() == 2 and points.shape[1] == 3
                and points.is_contiguous()):
            raise RuntimeError(f"points must be a contiguous float32 (M, 3) tensor on {dev}")
        if offsets is None:
            if self.B != 1 or points.shape[0] < max_view_points:
                raise RuntimeError("offsets may be None only for one view with max_view_points <= len(points)")
        elif not (offsets.device == dev and offsets.dtype is torch.int32 and offsets.shape == (self.B + 1,)
                  and offsets.is_contiguous()):
            raise RuntimeError(f"offsets must be a contiguous int32 tensor of shape ({self.B + 1},) on {dev}")
        if pc_workspace.device != dev or not pc_workspace.is_contiguous() or pc_workspace.dtype is not torch.uint8:
            raise RuntimeError(f"pc_workspace must be a contiguous uint8 tensor on {dev}")
        prepared, self._step_l1 = self._step_l1, None
        if prepared is not None:
            # second half of a step begun by forward_l1(prepare_backward=True): its backward writes the volume that
            # forward zero-filled.  As a step's backward (no prologue, the forward's view records and rectangles)
            # only for the very tensors the forward saw, judged like ``backward`` does (address and version counter)
            key, self.g_sdf, deferred = prepared
            self._g_sdf_index = self._g_sdf_next
            self._g_sdf_next = (self._g_sdf_next + 1) % len(self._g_sdf_ring)
            if key != self._key(sdf, pos, quat, inv_scale):
                if deferred:
                    raise RuntimeError("forward_l1(defer_loss=True) must be followed by the step's backward on the "
                                       "same tensors: its loss statistics were left to that launch")
                prepared = None
        if prepared is not None:
            self.partials_offset = self._L.sdfr_render_partials_offset(self.R, self.B, self.W, self.H, 1)
            self._fixed_layout = 1
        else:
            self.partials_offset = self._L.sdfr_render_partials_offset(self.R, self.B, self.W, self.H, 0)
            self._fixed_layout = 0
        deferred = prepared is not None and prepared[2]
        args = (loss_grad.data_ptr() if loss_grad is not None else None, weight,
                None if deferred else self.loss_stats.data_ptr(), target.data_ptr(), self.depth.data_ptr(),
                sdf.data_ptr(), self.R, self.sdf_stride, pos.data_ptr(), quat.data_ptr(), inv_scale.data_ptr(), self.B,
                self.W, self.H, self.cx, self.cy, self.fx, self.fy, self.sdf_grad_mode,
                self.g_sdf.data_ptr(), self.sdf_stride, self.workspace.data_ptr(), self.workspace.numel(),
                pc_weight, points.data_ptr(), offsets.data_ptr() if offsets is not None else None, max_view_points,
                scale.data_ptr(), pc_workspace.data_ptr(), pc_workspace.numel())
        if prepared is not None:
            rc = self._L.sdfr_render_step_backward_l1_pc(
                *args, self.loss.data_ptr() if deferred else None, self.loss_stats.data_ptr() if deferred else None,
                self.device.index, _stream(self.device))
        else:
            rc = self._L.sdfr_render_backward_l1_pc(*args, self.device.index, _stream(self.device))
        _lib.check(rc, "sdfr_render_backward_l1_pc" if prepared is None else "sdfr_render_step_backward_l1_pc")
        return self.g_sdf
