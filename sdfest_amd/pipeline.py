"""Render-and-compare optimisation loop: host-side mirror of the hot part of
``sdfest/estimation/simple_setup.py::SDFPipeline.__call__`` (:381-470) and its helpers
``_compute_view_losses`` (:115-162), ``quaternion_utils`` (:12-66) and
``pointset_utils.depth_to_pointcloud`` (:34-89).

What the reference does per iteration with a Python loop over views and ~200 small torch
kernels, this does with one decoder launch sequence, ONE batched render and ONE batched
point-cloud sample for all views (and their two backward launches); the per-view reductions and
Adam are a handful of torch ops on tiny tensors.  The initialisation network, logging,
visualisation and mesh export of the reference are out of scope.
"""
from typing import Dict, List, Optional, Sequence

import torch

from .differentiable_renderer import Camera, render_depth_batch
from .losses import pc_loss_batch


# ---- quaternion helpers (scalar-last), reference: initialization/quaternion_utils.py:12-66 ------

def quaternion_multiply(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    ax, ay, az, aw = torch.unbind(q1, -1)
    bx, by, bz, bw = torch.unbind(q2, -1)
    return torch.stack((aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw,
                        aw * bw - ax * bx - ay * by - az * bz), -1)


def quaternion_invert(q: torch.Tensor) -> torch.Tensor:
    return q * q.new_tensor([-1, -1, -1, 1])


def quaternion_apply(q: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    pq = torch.cat([points, points.new_zeros(points.shape[:-1] + (1,))], -1)
    return quaternion_multiply(quaternion_multiply(q, pq), quaternion_invert(q))[..., :3]


def depth_to_pointcloud(depth_image: torch.Tensor, camera: Camera) -> torch.Tensor:
    """(H,W) depth -> (N,3) OpenGL-frame points of the non-zero pixels, row-major order.

    Reference: pointset_utils.depth_to_pointcloud (:57-77, convention "opengl", no mask, no
    normalisation) -- note the pixel-centre-0 intrinsics."""
    fx, fy, cx, cy, _ = camera.get_pinhole_camera_parameters(0.0)
    rows, cols = torch.nonzero(depth_image, as_tuple=True)
    z = depth_image[rows, cols]
    return torch.stack(((cols.float() - cx) * z / fx, -(rows.float() - cy) * z / fy, -z), dim=1)


class RenderAndCompare:
    """Adam on (position, orientation, scale, latent) against V observed depth images.

    config keys (reference: estimation/configs/default.yaml): ``threshold``, ``max_iterations``,
    ``depth_weight``, ``pc_weight``; learning rates are the reference's (simple_setup.py:400-405).
    """

    def __init__(self, decoder, camera: Camera, config: Dict, device="cuda"):
        self.decoder = decoder
        self.cam = camera
        self.config = config
        self.device = torch.device(device)

    def prepare_views(self, depth_images: torch.Tensor):
        """Observed point clouds of all views, concatenated, with their segment offsets.  Done once
        per call (the reference recomputes it every iteration, simple_setup.py:134-136)."""
        clouds = [depth_to_pointcloud(d, self.cam) for d in depth_images]
        lens = [c.shape[0] for c in clouds]
        offsets = torch.tensor([0] + list(torch.tensor(lens).cumsum(0).tolist()), dtype=torch.int32,
                               device=self.device)
        points = torch.cat(clouds).contiguous() if sum(lens) else torch.zeros((0, 3), device=self.device)
        return points, offsets, lens

    def losses(self, depth_images, points, offsets, lens, cam_pos, cam_quat, position, orientation,
               scale, sdf):
        """Sum over views of the depth-L1 and point-cloud-L1 losses (simple_setup.py:411-446)."""
        V = depth_images.shape[0]
        norm_q = orientation / torch.sqrt(torch.sum(orientation ** 2))
        q_w2c = quaternion_invert(cam_quat)                                   # (V,4)
        pos_c = quaternion_apply(q_w2c, position - cam_pos)                   # (V,3)
        quat_c = quaternion_multiply(q_w2c, norm_q)                           # (V,4)
        inv_scale = (1.0 / scale).expand(V)
        est = render_depth_batch(sdf, pos_c.contiguous(), quat_c.contiguous(), inv_scale.contiguous(),
                                 self.config["threshold"], self.cam)          # (V,H,W)
        overlap = (depth_images > 0) & (est > 0)
        err = torch.abs(est - depth_images) * overlap
        loss_depth = (err.sum(dim=(1, 2)) / overlap.sum(dim=(1, 2))).sum()    # empty overlap -> NaN, as :131
        if points.shape[0]:
            val = pc_loss_batch(points, offsets, max(lens), pos_c.contiguous(), quat_c.contiguous(),
                                scale.expand(V).contiguous(), sdf)
            seg = torch.segment_reduce(val.abs(), "sum", offsets=offsets.long())
            loss_pc = (seg / torch.tensor(lens, device=self.device, dtype=torch.float32)).sum()
        else:
            loss_pc = est.new_tensor(float("nan"))
        return loss_depth, loss_pc, est

    def __call__(self, depth_images: torch.Tensor, position: torch.Tensor, orientation: torch.Tensor,
                 scale: torch.Tensor, latent: torch.Tensor,
                 camera_positions: Optional[torch.Tensor] = None,
                 camera_orientations: Optional[torch.Tensor] = None,
                 shape_optimization: bool = True, history: Optional[List] = None):
        """depth_images (V,H,W); position (1,3), orientation (1,4), scale (1,), latent (1,L): the
        initial estimate (the reference gets it from its init network, out of scope here).
        Returns the optimised (position, orientation, scale, latent)."""
        V = depth_images.shape[0]
        dev = self.device
        if camera_positions is None:
            camera_positions = torch.zeros((V, 3), device=dev)
        if camera_orientations is None:
            camera_orientations = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev).repeat(V, 1)
        position = position.detach().clone().requires_grad_()
        orientation = orientation.detach().clone().requires_grad_()
        scale = scale.detach().clone().requires_grad_()
        latent = latent.detach().clone().requires_grad_()
        optimizer = torch.optim.Adam([{"params": position, "lr": 1e-3},
                                      {"params": orientation, "lr": 1e-2},
                                      {"params": scale, "lr": 1e-3},
                                      {"params": latent, "lr": 1e-2}])
        points, offsets, lens = self.prepare_views(depth_images)
        for it in range(self.config["max_iterations"]):
            optimizer.zero_grad()
            with torch.set_grad_enabled(shape_optimization):
                sdf = self.decoder.decode(latent)[0, 0]
            loss_depth, loss_pc, est = self.losses(depth_images, points, offsets, lens,
                                                   camera_positions, camera_orientations, position,
                                                   orientation, scale, sdf)
            loss = self.config["depth_weight"] * loss_depth + self.config["pc_weight"] * loss_pc
            loss.backward()
            optimizer.step()
            with torch.no_grad():
                orientation /= torch.sqrt(torch.sum(orientation ** 2))
            if history is not None:
                history.append({"loss": loss.detach(), "loss_depth": loss_depth.detach(),
                                "loss_pc": loss_pc.detach(), "position": position.detach().clone(),
                                "orientation": orientation.detach().clone(),
                                "scale": scale.detach().clone(), "latent": latent.detach().clone()})
        return position.detach(), orientation.detach(), scale.detach(), latent.detach()
