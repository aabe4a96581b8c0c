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


class _NNLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points_from, points_to):
        a = points_from.detach().contiguous()
        b = points_to.detach().contiguous()
        N, M = a.shape[0], b.shape[0]
        dist = torch.empty(N, dtype=torch.float32, device=a.device)
        nearest = torch.empty(N, dtype=torch.int32, device=a.device)
        rc = _lib.lib().sdfr_nn_loss_forward(_ptr(a), N, _ptr(b), M, _ptr(dist), _ptr(nearest), a.device.index,
                                             _stream(a.device))
        _lib.check(rc, "sdfr_nn_loss_forward")
        ctx.save_for_backward(a, b, dist, nearest)
        return dist

    @staticmethod
    def backward(ctx, grad_dist):
        a, b, dist, nearest = ctx.saved_tensors
        g_from, g_to = torch.empty_like(a), torch.empty_like(b)
        rc = _lib.lib().sdfr_nn_loss_backward(_ptr(grad_dist.contiguous()), _ptr(a), a.shape[0], _ptr(b), b.shape[0],
                                              _ptr(dist), _ptr(nearest), _ptr(g_from), _ptr(g_to), a.device.index,
                                              _stream(a.device))
        _lib.check(rc, "sdfr_nn_loss_backward")
        return g_from, g_to


def nn_loss(points_from: torch.Tensor, points_to: torch.Tensor) -> torch.Tensor:
    """Squared distance from every point of ``points_from`` (N,3) to its nearest neighbour in
    ``points_to`` (M,3), shape (N,) -- reference: losses.py:8-29, same expression
    (-2 a.b + |a|^2 + |b|^2, negatives clamped to 0) and the gradient autograd gives it.  The
    reference's loop never uses it (``loss_nn = 0``, simple_setup.py:147); D = 3 only."""
    for t, n in ((points_from, "points_from"), (points_to, "points_to")):
        _check_input(t, n)
        if t.dim() != 2 or t.shape[1] != 3:
            raise RuntimeError(f"{n} must have shape (N, 3)")
    return _NNLoss.apply(points_from, points_to)


class _PointConstraint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, orientation_q, source, target):
        q, s, t = (x.detach().contiguous() for x in (orientation_q, source, target))
        loss = torch.empty(1, dtype=torch.float32, device=q.device)
        g_q = torch.zeros(4, dtype=torch.float32, device=q.device)
        rc = _lib.lib().sdfr_point_constraint(_ptr(q), _ptr(s), _ptr(t), 1.0, _ptr(loss), _ptr(g_q), q.device.index,
                                              _stream(q.device))
        _lib.check(rc, "sdfr_point_constraint")
        ctx.save_for_backward(g_q)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad):
        (g_q,) = ctx.saved_tensors
        return grad * g_q, None, None


def point_constraint_loss(orientation_q: torch.Tensor, source: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """| quaternion_apply(orientation_q, source) - target |, a scalar (reference: losses.py:138-153).
    quaternion_apply is q (v,0) conj(q): an un-normalised q also scales by |q|^2, as in the reference.
    Differentiable w.r.t. orientation_q (4,); source, target (3,) are constants."""
    for t, n, k in ((orientation_q, "orientation_q", 4), (source, "source", 3), (target, "target", 3)):
        _check_input(t, n)
        if tuple(t.shape) != (k,):
            raise RuntimeError(f"{n} must have shape ({k},)")
    return _PointConstraint.apply(orientation_q, source, target)
