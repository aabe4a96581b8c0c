"""Point-cloud loss sampler: host-side mirror of ``sdfest/estimation/losses.py::pc_loss``.

Same signature and return value as the reference (losses.py:32-135); the arithmetic runs in
``libsdfr_hip.so`` (sampler.hip) instead of ~40 torch kernels forward and as many backward.
"""
from typing import Optional

import torch

from . import _lib
from .differentiable_renderer import _check_input, _ptr, _stream, _workspace


def _forward_raw(points, offsets, max_pts, pos, quat, scale, sdf):
    B = pos.shape[0]
    R = sdf.shape[-1]
    stride = R * R * R if sdf.dim() == 4 else 0
    out = torch.empty(points.shape[0], dtype=torch.float32, device=points.device)
    rc = _lib.lib().sdfr_pc_loss_forward(_ptr(points), _ptr(offsets), B, max_pts, _ptr(pos),
                                         _ptr(quat), _ptr(scale), _ptr(sdf), R, stride, _ptr(out),
                                         points.device.index, _stream(points.device))
    _lib.check(rc, "sdfr_pc_loss_forward")
    return out


def _backward_raw(grad_out, points, offsets, max_pts, pos, quat, scale, sdf):
    B = pos.shape[0]
    R = sdf.shape[-1]
    stride = R * R * R if sdf.dim() == 4 else 0
    dev = points.device
    g_sdf = torch.empty_like(sdf)
    g_pos = torch.empty_like(pos)
    g_quat = torch.empty_like(quat)
    g_scale = torch.empty_like(scale)
    L = _lib.lib()
    ws = _workspace(dev, max(L.sdfr_pc_loss_backward_workspace_bytes(B, max_pts), 256))
    rc = L.sdfr_pc_loss_backward(_ptr(grad_out), _ptr(points), _ptr(offsets), B, max_pts, _ptr(pos),
                                 _ptr(quat), _ptr(scale), _ptr(sdf), R, stride, _ptr(g_sdf), stride,
                                 _ptr(g_pos), _ptr(g_quat), _ptr(g_scale), _ptr(ws), ws.numel(),
                                 dev.index, _stream(dev))
    _lib.check(rc, "sdfr_pc_loss_backward")
    return g_sdf, g_pos, g_quat, g_scale


class _PCLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, offsets, max_pts, position, orientation, scale, sdf):
        pts = points.detach().contiguous()
        pos = position.detach().contiguous()
        quat = orientation.detach().contiguous()
        sc = scale.detach().contiguous()
        vol = sdf.detach().contiguous()
        out = _forward_raw(pts, offsets, max_pts, pos, quat, sc, vol)
        ctx.save_for_backward(pts, pos, quat, sc, vol)
        ctx.offsets, ctx.max_pts = offsets, max_pts
        return out

    @staticmethod
    def backward(ctx, grad_out):
        pts, pos, quat, sc, vol = ctx.saved_tensors
        g_sdf, g_pos, g_quat, g_scale = _backward_raw(grad_out.contiguous(), pts, ctx.offsets,
                                                      ctx.max_pts, pos, quat, sc, vol)
        return None, None, None, g_pos, g_quat, g_scale, g_sdf


def pc_loss(points: torch.Tensor, position: torch.Tensor, orientation: torch.Tensor,
            scale: torch.Tensor, sdf: torch.Tensor) -> torch.Tensor:
    """Trilinearly interpolated SDF value at point positions (reference: losses.py:32-135).

    points (M,3) in the camera frame, position (3,), orientation (4,) quaternion (normalised
    inside, with gradient), scale () half-width of the SDF volume, sdf (res,res,res).
    Returns (M,): the distance in world units, 0 for points outside the volume.
    """
    for t, n in ((points, "points"), (position, "position"), (orientation, "orientation"),
                 (scale, "scale"), (sdf, "sdf")):
        _check_input(t, n)
    M = points.shape[0]
    out = _PCLoss.apply(points[:, :3] if points.shape[1] != 3 else points, None, M,
                        position.reshape(1, 3), orientation.reshape(1, 4), scale.reshape(1), sdf)
    return out


def pc_loss_batch(points: torch.Tensor, offsets: torch.Tensor, max_view_points: int,
                  positions: torch.Tensor, orientations: torch.Tensor, scales: torch.Tensor,
                  sdf: torch.Tensor) -> torch.Tensor:
    """All views of an optimisation step in one launch.

    points (N,3): the views' point clouds concatenated; offsets (B+1,) int32 on the device with
    view v owning points[offsets[v]:offsets[v+1]]; max_view_points: host int >= the longest
    segment; positions (B,3), orientations (B,4), scales (B,); sdf (R,R,R) or (B,R,R,R).
    """
    if offsets.dtype != torch.int32 or not offsets.is_cuda or not offsets.is_contiguous():
        raise RuntimeError("offsets must be a contiguous int32 CUDA tensor")
    return _PCLoss.apply(points, offsets, int(max_view_points), positions, orientations, scales, sdf)
