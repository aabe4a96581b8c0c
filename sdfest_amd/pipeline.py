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
import ctypes
from typing import Dict, List, Optional, Sequence

import torch

from .differentiable_renderer import Camera, render_depth_batch
from .losses import pc_loss_batch, point_constraint_loss


def compute_inlier_ratio(depth_input: torch.Tensor, depth_estimate: torch.Tensor,
                         relative_inlier_threshold: float = 0.03) -> torch.Tensor:
    """Ratio of pixels with a small relative depth error (simple_setup.py:177-188), as the reference
    writes it: a pixel without input depth divides by zero (inf or NaN, never below the threshold)."""
    rel = torch.abs(depth_input - depth_estimate) / depth_input
    return torch.count_nonzero(rel < relative_inlier_threshold) / torch.count_nonzero(depth_input)


class _BestEstimate:
    """simple_setup.py:190-211 and :583-596.  The reference keeps REFERENCES to the parameter tensors,
    which Adam and the renormalisation then update in place -- so its ``best_inlier_ratio`` result is
    the LAST iterate whatever the ratios were.  That is reproduced (`result()`); the snapshot one
    would expect from the name is kept beside it (`snapshot`, with its 1-based `iteration`)."""

    def __init__(self):
        self.ratio, self.iteration, self.snapshot = None, None, None

    def update(self, ratio, iteration, params):
        if self.ratio is None or ratio > self.ratio:
            self.ratio, self.iteration = ratio, iteration
            self.snapshot = tuple(p.detach().clone() for p in params)


def _selection_strategy(config: Dict) -> str:
    strategy = config.get("result_selection_strategy", "last_iteration")
    if strategy not in ("last_iteration", "best_inlier_ratio"):
        raise ValueError(f"Result selection strategy {strategy} is not supported.")   # :592-596
    return strategy


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


def depth_to_pointcloud(depth_image: torch.Tensor, camera: Camera, tiled: bool = False) -> torch.Tensor:
    """(H,W) depth -> (N,3) OpenGL-frame points of the non-zero pixels, row-major order.

    Reference: pointset_utils.depth_to_pointcloud (:57-77, convention "opengl", no mask, no
    normalisation) -- note the pixel-centre-0 intrinsics.  tiled: the same points in 16 x 16-pixel patches
    (``generated_views.depth_to_pointsets``)."""
    if depth_image.is_cuda and depth_image.dtype == torch.float32:
        from .generated_views import depth_to_pointsets   # sdfr_depth_count / sdfr_depth_to_points
        return depth_to_pointsets(depth_image[None], camera, tiled=tiled)[0]
    fx, fy, cx, cy, _ = camera.get_pinhole_camera_parameters(0.0)
    rows, cols = torch.nonzero(depth_image, as_tuple=True)
    z = depth_image[rows, cols]
    return torch.stack(((cols.float() - cx) * z / fx, -(rows.float() - cy) * z / fy, -z), dim=1)


def preprocess_depth(depth_images: torch.Tensor, masks: Optional[torch.Tensor], far_field: Optional[float] = None,
                     copy_to=None) -> torch.Tensor:
    """``SDFPipeline._preprocess_depth`` (simple_setup.py:671-693), in place like the reference:
    ``depth_images[~masks] = 0`` and, with a far field, ``depth_images[depth_images > far_field] = 0`` -- one kernel
    (``sdfr_preprocess_depth``) instead of two boolean-index assignments (each a mask conversion, a nonzero with
    its host synchronisation, and an index_put).  depth_images (N,H,W) float32 CUDA contiguous; masks: bool (or
    uint8) of the same shape, None = keep every pixel.  copy_to = (tensor (n,H,W), begin, end): the preprocessed
    images [begin, end) are also written there in the same pass (whole batch only when begin, end span it)."""
    from . import _lib
    if not (isinstance(depth_images, torch.Tensor) and depth_images.is_cuda and depth_images.dtype is torch.float32
            and depth_images.is_contiguous() and depth_images.dim() == 3):
        raise RuntimeError("depth_images must be a contiguous float32 CUDA tensor of shape (N, H, W)")
    N, H, W = depth_images.shape
    dev = depth_images.device
    if masks is None:
        m8 = torch.ones((N, H, W), dtype=torch.uint8, device=dev)
    else:
        if tuple(masks.shape) != (N, H, W):
            raise RuntimeError(f"masks must have the shape of depth_images {(N, H, W)}, got {tuple(masks.shape)}")
        m8 = masks.to(device=dev)
        m8 = (m8 if m8.dtype in (torch.bool, torch.uint8) else m8 != 0).contiguous()
        m8 = m8.view(torch.uint8) if m8.dtype is torch.bool else m8
    out, same = None, False
    if copy_to is not None:
        dst, b, e = copy_to
        same = (b, e) == (0, N) and dst.is_contiguous() and dst.dtype is torch.float32 and dst.device == dev \
            and tuple(dst.shape) == (N, H, W)
        out = dst if same else None
    L = _lib.lib()
    _lib.check(L.sdfr_preprocess_depth(depth_images.data_ptr(), m8.data_ptr(), N, W, H,
                                       float(far_field) if far_field is not None else 0.0,
                                       int(far_field is not None), out.data_ptr() if out is not None else None,
                                       dev.index, torch.cuda.current_stream(dev).cuda_stream), "sdfr_preprocess_depth")
    if copy_to is not None and not same:
        copy_to[0].copy_(depth_images[copy_to[1]:copy_to[2]])
    return depth_images


class RenderAndCompare:
    """Adam on (position, orientation, scale, latent) against V observed depth images.

    config keys (reference: estimation/configs/default.yaml): ``threshold``, ``max_iterations``,
    ``depth_weight``, ``pc_weight``; learning rates are the reference's (simple_setup.py:400-405).
    """

    def __init__(self, decoder, camera: Camera, config: Dict, device="cuda", process_group=None):
        """process_group (None | "world" | a torch.distributed group): the V views of a call are sharded over the
        group's ranks (``parallel.shard_views``); every rank passes the SAME full list and the same initial estimate,
        renders its shard, and the four parameter gradients (3 + 4 + 1 + L floats: autograd has already run the
        decoder's VJP on the rank's own d/dSDF, and the VJP is linear) are summed by ONE all-reduce before the
        replicated Adam step.  No broadcast: every rank ends with the same estimate."""
        from .parallel import resolve_group
        from .differentiable_renderer import parse_sdf_grad_mode
        self.decoder = decoder
        self.cam = camera
        self.config = config
        self.device = torch.device(device)
        self.group, self.rank, self.world = resolve_group(process_group)
        # config["sdf_grad_mode"]: "exact" (default) | "cuda_compat" -- see SDFPipeline
        self.sdf_grad_mode = parse_sdf_grad_mode(config.get("sdf_grad_mode"))

    def prepare_views(self, depth_images: torch.Tensor, tiled: bool = False):
        """Observed point clouds of all views, concatenated, with their segment offsets.  Done once
        per call (the reference recomputes it every iteration, simple_setup.py:134-136).  tiled: every view's
        points in compact patches instead of image rows (the point-cloud loss is a mean over them: any order)."""
        clouds = [depth_to_pointcloud(d, self.cam, tiled=tiled) for d in depth_images]
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
                                 self.config["threshold"], self.cam, self.sdf_grad_mode)          # (V,H,W)
        overlap = (depth_images > 0) & (est > 0)
        err = torch.abs(est - depth_images) * overlap
        count = overlap.sum(dim=(1, 2))
        # torch.mean(depth_error[overlap_mask]) (:131): an empty overlap makes the VALUE NaN, but no pixel is
        # selected, so no gradient flows from it -- the parameters keep finite gradients (the kernels of the
        # fused loop behave the same: loss NaN, contributions 0)
        per_view = torch.where(count > 0, err.sum(dim=(1, 2)) / count.clamp(min=1), err.new_tensor(float("nan")))
        loss_depth = per_view.sum()
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
                 shape_optimization: bool = True, history: Optional[List] = None,
                 point_constraint: Optional[Sequence] = None):
        """depth_images (V,H,W); position (1,3), orientation (1,4), scale (1,), latent (1,L): the
        initial estimate (the reference gets it from its init network).  point_constraint:
        (source (3,), target (3,), weight) as in simple_setup.py:224, :164-175.
        Returns the optimised (position, orientation, scale, latent); with
        ``config["result_selection_strategy"] == "best_inlier_ratio"`` what the reference returns for
        it (see :class:`_BestEstimate`); ``self.best`` holds the bookkeeping of the run.
        ``nn_weight`` has no effect: the reference's ``loss_nn`` is the constant 0 (:147)."""
        _selection_strategy(self.config)
        rel_thr = self.config.get("relative_inlier_threshold", 0.03)
        self.best = _BestEstimate()
        V_all = depth_images.shape[0]
        dev = self.device
        if camera_positions is None:
            camera_positions = torch.zeros((V_all, 3), device=dev)
        if camera_orientations is None:
            camera_orientations = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev).repeat(V_all, 1)
        if self.group is not None:
            import torch.distributed as dist
            from .parallel import shard_views
            b, e = shard_views(V_all, self.rank, self.world)
            if e == b:
                raise ValueError(f"{V_all} view(s) cannot be sharded over {self.world} ranks: every rank needs one")
            owns_last = e == V_all
            depth_images, camera_positions, camera_orientations = (depth_images[b:e], camera_positions[b:e],
                                                                   camera_orientations[b:e])
        V = depth_images.shape[0]
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
            if point_constraint is not None:   # on the un-normalised parameter, :164-175
                src, tgt, wgt = point_constraint
                loss_con = wgt * point_constraint_loss(orientation[0], src.to(dev, torch.float32),
                                                       tgt.to(dev, torch.float32))
            else:
                loss_con = orientation.new_tensor(0.0)
            if self.group is not None:
                loss_con = loss_con / self.world   # the constraint is not a per-view term: once over the ranks
            loss = (self.config["depth_weight"] * loss_depth + self.config["pc_weight"] * loss_pc
                    + loss_con)
            loss.backward()
            if self.group is not None:
                # the iteration's ONE exchange: [d/d position | orientation | scale | latent | loss terms | inlier counts]
                with torch.no_grad():
                    last_in, last_est = depth_images[V - 1], est[V - 1].detach()
                    rel = torch.abs(last_in - last_est) / last_in
                    counts = torch.stack((torch.count_nonzero(rel < rel_thr), torch.count_nonzero(last_in))).float()
                    lat_g = latent.grad if latent.grad is not None else torch.zeros_like(latent)
                    flat = torch.cat([position.grad.reshape(-1), orientation.grad.reshape(-1), scale.grad.reshape(-1),
                                      lat_g.reshape(-1), torch.stack((loss_depth.detach(), loss_pc.detach(),
                                                                      loss_con.detach())),
                                      counts * float(owns_last)])
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                    n = 0
                    for prm in (position, orientation, scale, latent):
                        if prm.grad is not None:
                            prm.grad.copy_(flat[n:n + prm.numel()].view_as(prm))
                        n += prm.numel()
                    loss_depth, loss_pc, loss_con = flat[n], flat[n + 1], flat[n + 2]
                    loss = (self.config["depth_weight"] * loss_depth + self.config["pc_weight"] * loss_pc + loss_con)
                    counts = flat[n + 3:n + 5]
            optimizer.step()
            with torch.no_grad():
                orientation /= torch.sqrt(torch.sum(orientation ** 2))
                # the reference passes the loop variables that survive its `for` over the views: the
                # LAST view's input and estimate, rendered before this step (:463-470)
                if self.group is not None:
                    ratio = counts[0] / counts[1]
                else:
                    ratio = compute_inlier_ratio(depth_images[V - 1], est[V - 1].detach(), rel_thr)
                self.best.update(ratio, it + 1, (position, orientation, scale, latent))
            if history is not None:
                history.append({"loss": loss.detach(), "loss_depth": loss_depth.detach(),
                                "loss_pc": loss_pc.detach(), "loss_point_constraint": loss_con.detach(),
                                "inlier_ratio": ratio, "position": position.detach().clone(),
                                "orientation": orientation.detach().clone(),
                                "scale": scale.detach().clone(), "latent": latent.detach().clone()})
        # "best_inlier_ratio" returns the tensors the reference stored -- the live parameters (:583-596)
        return position.detach(), orientation.detach(), scale.detach(), latent.detach()


class FusedRenderAndCompare:
    """The same optimisation as :class:`RenderAndCompare`, without autograd and without Python in
    the loop: every iteration is one fixed sequence of launches of ``libsdfr_hip.so`` (decoder with
    tape, pose chain, batched render, losses, batched sampler, both backward passes, gradient
    chain, decoder VJP, Adam) captured once into a hipGraph and replayed.

    Parameters live in one device buffer ``[position 3 | orientation 4 | scale 1 | latent L]``.
    """

    def __init__(self, decoder, camera: Camera, config: Dict, depth_images: Optional[torch.Tensor] = None,
                 camera_positions: Optional[torch.Tensor] = None,
                 camera_orientations: Optional[torch.Tensor] = None,
                 shape_optimization: bool = True, device="cuda", fuse_depth_loss: bool = True,
                 point_constraint: Optional[Sequence] = None, track_inliers: Optional[bool] = None,
                 merge_launches: bool = True, graph_iterations: int = 5, process_group=None,
                 exchange: str = "sdf", sdf_grad_mode: int = 0, form: str = "auto", views: Optional[int] = None,
                 graph_collective: bool = False, defer_loss: Optional[bool] = None,
                 fused_render: Optional[bool] = None, fc_in_tail: Optional[bool] = None):
        """depth_images (V,H,W): the first observation (``rebind`` takes the next ones: the reference calls its
        pipeline once per detected object with fresh images, simple_setup.py:213-225, and so re-uses nothing; this
        object keeps every buffer and every captured graph across observations of the same V, W, H).  None with
        ``views=V``: buffers only, ``rebind`` before the first run.
        point_constraint: (source (3,), target (3,), weight), simple_setup.py:164-175.
        process_group (None | "world" | a torch.distributed group): the loop SHARDED over the group's ranks, one process
        per GPU.  Every rank is given the same full view list, cameras and initial estimate and keeps its contiguous
        shard of the views (``parallel.shard_views``).  An iteration then has exactly ONE exchange, an all-reduce of one
        bucket, between the two backward passes and the update (include/sdfr.h, "the loop sharded over ranks"):
          exchange="sdf"     [d loss / d SDF, R^3 words | one 20-float record per view]: every rank then
                             runs the identical decoder VJP, chain and Adam step on the summed volume;
          exchange="latent"  every rank runs the decoder VJP on ITS d/dSDF (the VJP is linear) and the bucket is
                             [view records | d loss / d latent]: 80 V + 4 L bytes instead of 1 MiB.  Same result up to
                             the rounding of the sum.
        Parameters and Adam state are replicated and stay identical on every rank; nothing is broadcast.
        form: how the per-view results reach the update.  "tail": ONE workgroup reduces every view's partial sums and
        runs the chain and Adam (``sdfr_loop_tail``: the fewest launches -- best for a handful of views, at most 64);
        "records": one wave per view writes the view's record (``sdfr_loop_view_records``), the tail works from the
        records (``sdfr_loop_tail_records``) -- one launch more, any number of views, and the only form a process
        group can use; "auto": records with a process group or from 8 views on (ms per iteration on one MI355X, 640x480
        views, tail / records: 1 view 0.135 / 0.139, 4: 0.152 / 0.152, 8: 0.165 / 0.161, 16: 0.199 / 0.175, 64: 0.304 /
        0.226), the tail otherwise.
        sdf_grad_mode: flag bits for the renderer's backward (differentiable_renderer.SDF_GRAD_*, BWD_*).  With
        ``SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES`` and exchange="sdf" the bucket is summed as integers and the
        trajectory is bitwise the same however the views are spread over ranks (and as a single process).
        defer_loss: the step's forward leaves the reduction of its depth-loss tile records to the backward's launch
        (include/sdfr.h, "DEFERRED LOSS": one dependent launch less per iteration, a count loop in every backward tile
        that holds a hit pixel); default: for fewer than 32 views (ms per iteration on the C5 scene, own launch / deferred:
        1 view 0.1223 / 0.1196, 8: 0.1531 / 0.1501, 16: 0.1784 / 0.1747, 64: 0.2115 / 0.2226 -- many views, many tiles
        that each repeat the count; profiles/r05_pc_rounds.md).  Same numbers either way.
        graph_collective (process group over RCCL only): try to capture the all-reduce INSIDE the graphs -- whole
        iterations (head | all-reduce | tail) x graph_iterations as ONE graph, instead of two graphs per iteration with
        the collective issued between them.  c10d's NCCL backend is capturable; whether RCCL's kernels replay on this
        platform is what ``self.graph_collective_error`` says afterwards (None: captured; a string: why not -- the loop
        then runs the two-graph form).  An experiment, off by default.
        fused_render: the render pair of an iteration as ONE launch (``sdfr_render_step_fused_l1_pc``: a tile runs the
        backward of its hit pixels right behind their march; include/sdfr.h) -- for the tail form (at most 7 views; every
        view's depth term has its own weight / count and so its own unscaled d/dSDF volume) of a grid up to 128^3,
        loss-fused, with plain ``sdf_grad_mode`` weights; default: where it measured faster (up to 4 views with shape
        optimisation, any tail-form loop without; C5: 0.110 -> 0.100 ms per iteration).  Depth
        images bit for bit the two launches'; gradients equal up to rounding (the view's weight / count multiplies sums
        instead of terms), so the trajectory is the two-launch form's to ~1e-6.
        fc_in_tail (with fused_render and shape optimisation, for decoders whose Linear stack has narrow leading layers:
        ``SDFDecoder.narrow_linear_stack``): the tail's launch also runs the decoder's Linear stack for the latent it has
        just updated (``sdfr_loop_tail_fused(decoder_tape=)``: one workgroup per 256 outputs of the wide layer, each
        repeating the latent's share of the tail), and an iteration's decode starts behind it
        (``sdfr_decoder_forward_stage``) -- one launch less per iteration, the same numbers; default: wherever it applies.
        graph_iterations: iterations per replayed hipGraph (a graph launch costs ~5-8 us between iterations; the
        remainder of max_iterations and runs with ``history`` replay the one-iteration graph).
        merge_launches: the per-view reductions of both backward passes run inside the gradient chain's launch
        (``sdfr_views_to_pose_grad_deferred``, up to 64 views); False: one launch each.  Same numbers.
        track_inliers: run the inlier-ratio bookkeeping of :177-211 every iteration (two small launches);
        default: only for ``result_selection_strategy == "best_inlier_ratio"``."""
        from . import _lib
        from .differentiable_renderer import BatchRenderPlan
        # True: the depth-L1 runs inside the render kernels (sdfr_render_forward_l1 / _backward_l1) and the
        # point-cloud L1 inside the sampler's backward (sdfr_pc_l1_backward);
        # False: every loss is a kernel of its own between a forward and a backward.  Same results.
        self.fuse_depth_loss = bool(fuse_depth_loss)
        self.graph_iterations = max(1, int(graph_iterations))
        self.graph_collective = bool(graph_collective)
        self._defer_loss_arg = defer_loss
        self.graph_collective_error = None
        self.graph_whole = self.graph_whole_one = None
        self.L = _lib.lib()
        self.check = _lib.check
        self.dec = decoder
        self.cam = camera
        self.cfg = config
        self.dev = torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.shape_opt = bool(shape_optimization)
        from .differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC, VIEW_RECORD_FLOATS
        from .parallel import resolve_group, shard_views
        if exchange not in ("sdf", "latent"):
            raise ValueError(f"exchange must be 'sdf' or 'latent', got {exchange!r}")
        self.group, self.rank, self.world = resolve_group(process_group)
        self.exchange = exchange
        self.sdf_grad_mode = int(sdf_grad_mode)
        self.det = bool(self.sdf_grad_mode & SDF_GRAD_DETERMINISTIC)
        # the records form of the iteration: head (decoder .. both backward passes .. view records), [exchange],
        # tail (decoder VJP, chain over ALL views' records, Adam, next poses).  A single process takes it too when
        # the pose sums are asked to be independent of the batch (BWD_SMALL_TILES): same arithmetic as the ranks'.
        if depth_images is None and views is None:
            raise ValueError("give the first observation (depth_images) or the number of views (views=)")
        self.V_all = int(depth_images.shape[0]) if depth_images is not None else int(views)
        if views is not None and int(views) != self.V_all:
            raise ValueError(f"views={views} but depth_images holds {self.V_all}")
        if form not in ("auto", "tail", "records"):
            raise ValueError(f"form must be 'auto', 'tail' or 'records', got {form!r}")
        must = self.group is not None or bool(self.sdf_grad_mode & BWD_SMALL_TILES)
        if form == "tail" and must:
            raise ValueError("a process group (and BWD_SMALL_TILES) needs the records form")
        self.records_form = must or form == "records" or (form == "auto" and self.V_all >= 8 and bool(fuse_depth_loss))
        self.view_begin, self.view_end = shard_views(self.V_all, self.rank, self.world)
        if self.view_end == self.view_begin:
            raise ValueError(f"{self.V_all} view(s) cannot be sharded over {self.world} ranks: every rank needs one")
        if self.records_form and not fuse_depth_loss:
            raise ValueError("the sharded loop runs the loss-fused kernels (fuse_depth_loss=True)")
        if self.det and not self.records_form:
            raise ValueError("SDF_GRAD_DETERMINISTIC in the loop goes with BWD_SMALL_TILES (or a process group)")
        f32 = dict(dtype=torch.float32, device=self.dev)
        # every rank keeps the WHOLE camera list (the records form's tail chains over all views); its own shard's is a
        # view of it.  Allocated once: ``rebind`` copies a new observation's cameras in place
        self.cam_pos_all = torch.zeros((self.V_all, 3), **f32)
        self.cam_quat_all = torch.zeros((self.V_all, 4), **f32)
        self.cam_pos = self.cam_pos_all[self.view_begin:self.view_end]
        self.cam_quat = self.cam_quat_all[self.view_begin:self.view_end]
        V, H, W = self.view_end - self.view_begin, int(camera.height), int(camera.width)
        if depth_images is not None and tuple(depth_images.shape[1:]) != (H, W):
            raise ValueError(f"depth_images must be (V, {H}, {W}) for this camera, got {tuple(depth_images.shape)}")
        self.V, self.H, self.W = V, H, W
        self.defer_loss = (V < 32) if self._defer_loss_arg is None else bool(self._defer_loss_arg)
        self.defer_pose = bool(merge_launches) and V <= 64
        # the iteration's tail -- gradient chain, point constraint, Adam and the next iteration's view poses -- in
        # one launch (sdfr_loop_tail), the render pair in its step form: 26 -> 22 launches per iteration
        self.merge_tail = bool(merge_launches) and V <= 64
        self.target = torch.zeros((V, H, W), **f32)
        # The observed point clouds at CAPACITY -- room for every pixel of every view, counts and offsets on the
        # device (sdfr_depth_to_points_resident) -- so that a new observation changes no address, no grid size and
        # needs no host read: the captured graphs stay valid.  The sampler's grids are sized for W * H points per
        # view; blocks beyond a view's count leave at once (sampler_device.hpp), the reducers take the real lengths
        # from `offsets`: the same sums in the same order as with exactly-sized buffers.
        # (compact patches: the sampler's backward pre-sums 256 consecutive points before its global atomics)
        self.max_pts = W * H
        self.points = torch.zeros((V * W * H, 3), **f32)
        self.offsets = torch.zeros(V + 1, dtype=torch.int32, device=self.dev)
        self.counts = torch.zeros(V, dtype=torch.int32, device=self.dev)
        self.ws_points = torch.empty(max(self.L.sdfr_depth_points_workspace_bytes(V, W, H), 256), dtype=torch.uint8,
                                     device=self.dev)
        self.Lz = decoder.latent_size
        n = 8 + self.Lz
        # the run's state in ONE buffer, [params | everything a run starts from zero]: one fill resets a run
        n4 = (n + 3) // 4 * 4
        self.max_history = int(config["max_iterations"])
        nh = (max(self.max_history, 1) + 3) // 4 * 4
        self._state = torch.zeros(4 * n4 + nh + 8, **f32)
        self.params = self._state[0:n]
        self.m = self._state[n4:n4 + n]
        self.v = self._state[2 * n4:2 * n4 + n]
        self.best_params = self._state[3 * n4:3 * n4 + n]
        self.inlier_history = self._state[4 * n4:4 * n4 + max(self.max_history, 1)]
        o = 4 * n4 + nh
        self.best_state = self._state[o:o + 3]        # best ratio, its 1-based iteration, has_best
        self.loss_con = self._state[o + 3:o + 4]      # the point constraint's loss value (0 without one)
        ints = self._state[o + 4:o + 8].view(torch.int32)
        self.step = ints[0:1]
        self.inlier_counts = ints[1:3]
        self.arrivals = ints[3:4]                     # sdfr_loop_tail_fused(decoder_tape=): zero at a run's start
        self._state_zero = self._state[n4:]           # (everything but the parameters)
        self.grads = torch.zeros(n, **f32)
        R = decoder._volume_size
        self.R = R
        n_rec = self.V_all * VIEW_RECORD_FLOATS
        self.big_exchange = self.records_form and self.shape_opt and exchange == "sdf"
        self.plan = BatchRenderPlan(R, V, camera, device=self.dev, sdf_grad_mode=self.sdf_grad_mode,
                                    grad_tail_words=n_rec if (self.big_exchange and not self.det) else 0)
        self.xbuf = self.records = None
        if self.records_form:
            if self.big_exchange and self.det:     # [int64 volume | records]
                self.xbuf = torch.zeros(R ** 3 + n_rec // 2, dtype=torch.int64, device=self.dev)
                self.records = self.xbuf[R ** 3:].view(torch.float32)
            elif self.big_exchange:                # the plan's bucket [float volume | records]
                self.xbuf = self.plan._bucket_ring[0]
                self.records = self.xbuf[R ** 3:R ** 3 + n_rec]
            else:                                  # [records | gradient vector] (the latter only travels for "latent")
                with_g = self.shape_opt and exchange == "latent"
                n_x = n_rec + (n + n % 2 if with_g else 0)
                self.xbuf = torch.zeros(n_x, **f32)
                self.records = self.xbuf[:n_rec]
                if with_g:
                    self.grads = self.xbuf[n_rec:n_rec + n]
            self.owns_last = self.view_end == self.V_all
        self.pos_c = torch.empty((V, 3), **f32)
        self.quat_c = torch.empty((V, 4), **f32)
        self.inv_scale = torch.empty((V,), **f32)
        self.scale_v = torch.empty((V,), **f32)
        # (the gradient image and the per-point arrays exist only when the losses run as kernels of their own)
        self.grad_est = None if self.fuse_depth_loss else torch.empty((V, H, W), **f32)
        self.loss_depth = torch.zeros((V,), **f32)
        self.loss_pc = torch.zeros((V,), **f32)
        N = self.points.shape[0]     # (capacity)
        self.vals = None if self.fuse_depth_loss else torch.empty((max(N, 1),), **f32)
        self.grad_vals = None if self.fuse_depth_loss else torch.empty((max(N, 1),), **f32)
        self.g_sdf_pc = torch.empty((R, R, R), **f32)
        self.g_pos_pc = torch.empty((V, 3), **f32)
        self.g_quat_pc = torch.empty((V, 4), **f32)
        self.g_scale_pc = torch.empty((V,), **f32)
        self.sdf = torch.empty((1, 1, R, R, R), **f32)
        u8 = dict(dtype=torch.uint8, device=self.dev)
        self.tape = torch.empty(max(self.L.sdfr_decoder_tape_bytes(decoder._h, 1), 256), **u8)
        self.ws_dec = torch.empty(max(self.L.sdfr_decoder_workspace_bytes(decoder._h, 1),
                                      self.L.sdfr_decoder_backward_workspace_bytes(decoder._h, 1), 256), **u8)
        self.ws_loss = torch.empty(max(self.L.sdfr_depth_l1_workspace_bytes(V, W, H), 256), **u8)
        self.ws_pc = torch.empty(max(self.L.sdfr_pc_loss_backward_workspace_bytes(V, self.max_pts), 256), **u8)
        can_fuse = (self.merge_tail and not self.records_form and self.fuse_depth_loss and R <= 128 and V <= 8
                    and self.max_pts > 0 and self.sdf_grad_mode in (0, 1) and 8 + self.Lz <= 256)
        if fused_render and not can_fuse:
            raise ValueError("fused_render needs the tail form over at most 8 views of a grid up to 128^3, the loss-fused "
                             "kernels and sdf_grad_mode 0 / 1")
        # (default: where it measured faster -- with shape optimisation up to 4 views: 1 view 0.111 -> 0.099 ms per
        # iteration, 2: 0.112 -> 0.102, 3: 0.114 -> 0.107, 4: 0.119 -> 0.116, but 7: 0.126 -> 0.132 -- every view adds a
        # megabyte to sum, to clear and to collide in; pose only, where no volume exists, wherever it applies)
        auto = can_fuse and (V <= 4 or not self.shape_opt)
        self.fused_render = auto if fused_render is None else bool(fused_render)
        can_fc = bool(self.fused_render and self.shape_opt and getattr(decoder, "narrow_linear_stack", lambda: False)())
        if fc_in_tail and not can_fc:
            raise ValueError("fc_in_tail needs fused_render, shape optimisation and a decoder with a narrow Linear stack")
        self.fc_in_tail = can_fc if fc_in_tail is None else bool(fc_in_tail)
        self._sums_open = False
        if self.fused_render:
            # what the one-launch step ADDS into must start from zero (its consumers clear what they have read)
            self._zero_fused_sums()
        self.strategy = _selection_strategy(config)
        self.track_inliers = (self.strategy == "best_inlier_ratio") if track_inliers is None else bool(track_inliers)
        self.rel_thr = float(config.get("relative_inlier_threshold", 0.03))
        # the point constraint's two points live in buffers of this object (``rebind`` copies new ones in place);
        # whether there is one, and its weight, are launch ARGUMENTS and so part of a captured graph: graphs are kept
        # per (constraint present, weight) -- a caller alternating between "none" and one weight captures twice, once
        self._con_points = torch.zeros(8, **f32)
        self.pc_source = None
        self.pc_weight = 0.0
        self._graphs = {}
        self.graph = self.graph_many = self.graph_tail = None
        self.bound = False
        if depth_images is not None:
            self.rebind(depth_images, camera_positions, camera_orientations, point_constraint)

    def rebind(self, depth_images: torch.Tensor, camera_positions: Optional[torch.Tensor] = None,
               camera_orientations: Optional[torch.Tensor] = None, point_constraint: Optional[Sequence] = None,
               masks: Optional[torch.Tensor] = None, far_field: Optional[float] = None) -> "FusedRenderAndCompare":
        """A NEW observation for the same views, image size and decoder: what the reference does by calling its
        pipeline again (simple_setup.py:213-225, :333-334, :420-446).  The images go into this object's target
        buffer, their point clouds are rebuilt on the device into the buffers the captured graphs already point at
        (no allocation, no host synchronisation, no capture), the cameras and the constraint's points are copied in
        place.  The next ``__call__`` starts a fresh run (Adam state, step count and inlier bookkeeping are reset
        there) and replays the EXISTING graphs.
        depth_images (V_all,H,W), every rank the full list, on any device (a host tensor costs its upload).
        masks / far_field: ``SDFPipeline._preprocess_depth`` (:671-693) on the way -- depth_images is then modified IN
        PLACE like the reference's argument (it must be a float32 CUDA tensor; masks a bool tensor of its shape)."""
        if tuple(depth_images.shape) != (self.V_all, self.H, self.W):
            raise ValueError(f"depth_images must have shape {(self.V_all, self.H, self.W)}, "
                             f"got {tuple(depth_images.shape)}")
        L, d, st = self.L, self.dev.index, self._stream()
        b, e = self.view_begin, self.view_end
        with torch.no_grad():
            if masks is not None or far_field is not None:
                preprocess_depth(depth_images, masks, far_field, copy_to=(self.target, b, e))
            else:
                self.target.copy_(depth_images[b:e])
            if camera_positions is None:
                self.cam_pos_all.zero_()
            else:
                self.cam_pos_all.copy_(camera_positions.reshape(self.V_all, 3))
            if camera_orientations is None:
                self.cam_quat_all.zero_()
                self.cam_quat_all[:, 3] = 1.0
            else:
                self.cam_quat_all.copy_(camera_orientations.reshape(self.V_all, 4))
            fx, fy, cx0, cy0, _ = self.cam.get_pinhole_camera_parameters(0.0)
            self.check(L.sdfr_depth_to_points_resident(
                self.target.data_ptr(), self.V, self.W, self.H, 1, 1.0 / fx, 1.0 / fy, cx0, cy0,
                self.counts.data_ptr(), self.offsets.data_ptr(), self.ws_points.data_ptr(), self.ws_points.numel(),
                self.points.data_ptr(), d, st), "sdfr_depth_to_points_resident")
            if point_constraint is not None:
                src, tgt, wgt = point_constraint
                self._con_points[0:3].copy_(torch.as_tensor(src).reshape(3))
                self._con_points[4:7].copy_(torch.as_tensor(tgt).reshape(3))
                self.pc_source, self.pc_target = self._con_points[0:3], self._con_points[4:7]
                self.pc_weight = float(wgt)
            else:
                self.pc_source, self.pc_weight = None, 0.0
        key = None if self.pc_source is None else self.pc_weight
        # (the whole-iteration graphs of `graph_collective` carry the constraint's launch arguments too: they belong
        # to the key like the others -- until round 6 a rebind to another constraint kept replaying the old ones)
        (self.graph, self.graph_many, self.graph_tail, self.graph_whole_one,
         self.graph_whole) = self._graphs.get(key, (None, None, None, None, None))
        self._graph_key = key
        self.bound = True
        return self

    def _zero_fused_sums(self):
        """the volumes and counts ``sdfr_render_step_fused_l1_pc`` adds into (normally left at zero by their consumers)"""
        if self.shape_opt:     # (a loop that does not optimise the shape has no d/dSDF volumes at all)
            self.plan._g_sdf_ring[0].zero_()
            self.plan.g_depth.zero_()
        self.plan.view_count.zero_()
        self._sums_open = False

    def _keep_graphs(self):
        # (at most 4 sets: a caller whose constraint weight changes from call to call re-captures instead of
        # collecting graphs without bound)
        if self._graph_key not in self._graphs and len(self._graphs) >= 4:
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[self._graph_key] = (self.graph, self.graph_many, self.graph_tail, self.graph_whole_one,
                                         self.graph_whole)

    def view_point_counts(self) -> torch.Tensor:
        """observed points per view of this rank's shard, (V,) int32 on the device (reading it synchronises)"""
        return self.counts

    # views of the parameter buffer
    @property
    def position(self):
        return self.params[0:3]

    @property
    def orientation(self):
        return self.params[3:7]

    @property
    def scale(self):
        return self.params[7:8]

    @property
    def latent(self):
        return self.params[8:]

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _decode(self, st, with_tape, stages=3):
        """stages (``sdfr_decoder_forward_stage``): 3 the whole decode; with ``fc_in_tail`` an iteration's decode is stage 2
        (the Linear stack's output is in the tape: the previous tail's launch left it there), and stage 1 runs once in
        front of the first iteration (``_prime``)"""
        if stages == 3:
            rc = self.L.sdfr_decoder_forward(self.dec._h, self.latent.data_ptr(), 1, 0, self.sdf.data_ptr(),
                                             self.tape.data_ptr() if with_tape else None, self.ws_dec.data_ptr(),
                                             self.ws_dec.numel(), st)
        else:
            rc = self.L.sdfr_decoder_forward_stage(self.dec._h, self.latent.data_ptr(), 1, 0, self.sdf.data_ptr(),
                                                   self.tape.data_ptr(), self.ws_dec.data_ptr(), self.ws_dec.numel(), st,
                                                   stages)
        self.check(rc, "sdfr_decoder_forward")

    def _prime(self):
        """what the FIRST iteration of a run finds where every later one finds the previous tail's results: the views'
        poses of the initial estimate and -- ``fc_in_tail`` -- its Linear-stack output"""
        st = self._stream()
        self._poses_to_views(st)
        if self.fc_in_tail:
            self._decode(st, True, stages=1)

    def iteration(self):
        """One iteration of simple_setup.py:408-462 as a launch sequence on the current stream."""
        L, d, st = self.L, self.dev.index, self._stream()
        p = self.params.data_ptr()
        pos, quat, scale = p, p + 12, p + 28
        g = self.grads.data_ptr()
        if self.shape_opt:
            self._decode(st, True, stages=2 if self.fc_in_tail else 3)
        sdf = self.sdf[0, 0]
        if self._tail_form():
            # [decoder] -> render pair as a step -> [decoder VJP] -> sdfr_loop_tail; the camera-frame poses of THIS
            # iteration were left by the previous tail (or by _poses_to_views before the first one)
            # (the depth loss values are reduced inside the backward's launch: no launch between the image kernels)
            if self.fused_render:
                return self._iteration_one_launch(L, d, st, p, g, sdf)
            self.plan.forward_l1(sdf, self.pos_c, self.quat_c, self.inv_scale, self.cfg["threshold"], self.target,
                                 prepare_backward=True, defer_loss=self.defer_loss)
            self.loss_depth = self.plan.loss
            g_sdf = self.plan.backward_l1_pc(self.target, sdf, self.pos_c, self.quat_c, self.inv_scale,
                                             self.scale_v, self.points, self.offsets, self.max_pts, self.ws_pc,
                                             weight=self.cfg["depth_weight"], pc_weight=self.cfg["pc_weight"])
            t_mid = None
            if self.shape_opt:
                # the VJP's last launch (one workgroup) runs inside the tail's launch
                t_mid = ctypes.c_void_p()
                self.check(L.sdfr_decoder_backward_latent_deferred(
                    self.dec._h, self.latent.data_ptr(), self.tape.data_ptr(), g_sdf.data_ptr(), self.ws_dec.data_ptr(),
                    self.ws_dec.numel(), st, ctypes.byref(t_mid)), "sdfr_decoder_backward_latent_deferred")
            con = self.pc_source is not None
            self.check(L.sdfr_loop_tail(
                p, g, self.m.data_ptr(), self.v.data_ptr(), self.step.data_ptr(), 8 + self.Lz, 1e-3, 1e-2, 1e-3, 1e-2,
                int(self.shape_opt), self.cam_pos.data_ptr(), self.cam_quat.data_ptr(), self.V,
                self.plan.workspace.data_ptr(), self.plan.partials_offset, self.W, self.H, self.ws_pc.data_ptr(),
                self.offsets.data_ptr(), self.max_pts, self.pos_c.data_ptr(), self.quat_c.data_ptr(),
                self.inv_scale.data_ptr(), self.scale_v.data_ptr(), self.loss_pc.data_ptr(),
                self.pc_source.data_ptr() if con else None, self.pc_target.data_ptr() if con else None,
                self.pc_weight if con else 0.0, self.loss_con.data_ptr() if con else None,
                self.dec._h if t_mid is not None else None, t_mid, d, st), "sdfr_loop_tail")
            self._inliers(L, p, d, st)
            return
        self._poses_to_views(st)
        # the renderer's and the sampler's per-view reductions run inside the gradient chain's launch
        # (sdfr_views_to_pose_grad_deferred: two launches less per iteration)
        defer = self.defer_pose
        have_pts = self.max_pts > 0
        summed = False   # the sampler's d/dSDF already added to the renderer's
        if self.fuse_depth_loss and defer and have_pts:
            # both backward passes side by side in one launch (they are independent and neither fills the chip)
            self.plan.forward_l1(sdf, self.pos_c, self.quat_c, self.inv_scale, self.cfg["threshold"],
                                 self.target)
            self.loss_depth = self.plan.loss
            g_sdf = self.plan.backward_l1_pc(self.target, sdf, self.pos_c, self.quat_c, self.inv_scale,
                                             self.scale_v, self.points, self.offsets, self.max_pts, self.ws_pc,
                                             weight=self.cfg["depth_weight"], pc_weight=self.cfg["pc_weight"])
            g_pos = g_quat = g_is = None
            summed = True
        elif self.fuse_depth_loss:
            self.plan.forward_l1(sdf, self.pos_c, self.quat_c, self.inv_scale, self.cfg["threshold"],
                                 self.target)
            self.loss_depth = self.plan.loss
            g_sdf, g_pos, g_quat, g_is = self.plan.backward_l1(self.target, sdf, self.pos_c, self.quat_c,
                                                               self.inv_scale,
                                                               weight=self.cfg["depth_weight"], defer_pose=defer)
        else:
            est = self.plan.forward(sdf, self.pos_c, self.quat_c, self.inv_scale, self.cfg["threshold"])
            self.check(L.sdfr_depth_l1_loss(est.data_ptr(), self.target.data_ptr(), self.V, self.W, self.H,
                                            self.cfg["depth_weight"], self.loss_depth.data_ptr(),
                                            self.grad_est.data_ptr(), self.ws_loss.data_ptr(),
                                            self.ws_loss.numel(), d, st), "sdfr_depth_l1_loss")
            g_sdf, g_pos, g_quat, g_is = self.plan.backward(self.grad_est, sdf, self.pos_c, self.quat_c,
                                                            self.inv_scale, defer_pose=defer)
        if summed:
            pass
        elif have_pts and self.fuse_depth_loss:
            # the point-cloud term in one pass: loss value and gradients, no sampler forward, no loss kernel;
            # its d/dSDF goes straight into the volume the renderer's backward has just written
            self.check(L.sdfr_pc_l1_backward_accumulate(
                self.cfg["pc_weight"], self.loss_pc.data_ptr(), self.points.data_ptr(), self.offsets.data_ptr(),
                self.V, self.max_pts, self.pos_c.data_ptr(), self.quat_c.data_ptr(), self.scale_v.data_ptr(),
                sdf.data_ptr(), self.R, 0, g_sdf.data_ptr(), 0, None if defer else self.g_pos_pc.data_ptr(),
                None if defer else self.g_quat_pc.data_ptr(), None if defer else self.g_scale_pc.data_ptr(),
                self.ws_pc.data_ptr(), self.ws_pc.numel(), d, st),
                "sdfr_pc_l1_backward_accumulate")
            summed = True
        elif have_pts:
            self.check(L.sdfr_pc_loss_forward(self.points.data_ptr(), self.offsets.data_ptr(), self.V,
                                              self.max_pts, self.pos_c.data_ptr(), self.quat_c.data_ptr(),
                                              self.scale_v.data_ptr(), sdf.data_ptr(), self.R, 0,
                                              self.vals.data_ptr(), d, st), "sdfr_pc_loss_forward")
            self.check(L.sdfr_pc_l1_loss(self.vals.data_ptr(), self.offsets.data_ptr(), self.V, self.max_pts,
                                         self.cfg["pc_weight"], self.loss_pc.data_ptr(),
                                         self.grad_vals.data_ptr(), d, st), "sdfr_pc_l1_loss")
            self.check(L.sdfr_pc_loss_backward(self.grad_vals.data_ptr(), self.points.data_ptr(),
                                               self.offsets.data_ptr(), self.V, self.max_pts,
                                               self.pos_c.data_ptr(), self.quat_c.data_ptr(),
                                               self.scale_v.data_ptr(), sdf.data_ptr(), self.R, 0,
                                               self.g_sdf_pc.data_ptr(), 0,
                                               None if defer else self.g_pos_pc.data_ptr(),
                                               None if defer else self.g_quat_pc.data_ptr(),
                                               None if defer else self.g_scale_pc.data_ptr(),
                                               self.ws_pc.data_ptr(), self.ws_pc.numel(), d, st),
                       "sdfr_pc_loss_backward")
        if defer:
            self.check(L.sdfr_views_to_pose_grad_deferred(
                quat, scale, self.cam_quat.data_ptr(), self.V, self.plan.workspace.data_ptr(), self.W, self.H,
                self.ws_pc.data_ptr() if have_pts else None, self.offsets.data_ptr() if have_pts else None,
                self.max_pts, self.quat_c.data_ptr(),
                self.loss_pc.data_ptr() if (have_pts and self.fuse_depth_loss) else None,
                g, g + 12, g + 28, d, st), "sdfr_views_to_pose_grad_deferred")
        else:
            self.check(L.sdfr_views_to_pose_grad(quat, scale, self.cam_quat.data_ptr(), self.V, g_pos.data_ptr(),
                                                 g_quat.data_ptr(), g_is.data_ptr(),
                                                 self.g_pos_pc.data_ptr() if have_pts else None,
                                                 self.g_quat_pc.data_ptr() if have_pts else None,
                                                 self.g_scale_pc.data_ptr() if have_pts else None,
                                                 g, g + 12, g + 28, d, st), "sdfr_views_to_pose_grad")
        if self.pc_source is not None:   # + the point constraint's gradient on the orientation parameter
            self.check(L.sdfr_point_constraint(quat, self.pc_source.data_ptr(), self.pc_target.data_ptr(),
                                               self.pc_weight, self.loss_con.data_ptr(), g + 12, d, st),
                       "sdfr_point_constraint")
        if self.shape_opt:
            if have_pts and not summed:
                self.check(L.sdfr_add_inplace(g_sdf.data_ptr(), self.g_sdf_pc.data_ptr(), g_sdf.numel(), d, st),
                           "sdfr_add_inplace")
            self.check(L.sdfr_decoder_backward_latent(self.dec._h, self.latent.data_ptr(), self.tape.data_ptr(),
                                                      g_sdf.data_ptr(), 1, g + 32, self.ws_dec.data_ptr(),
                                                      self.ws_dec.numel(), st), "sdfr_decoder_backward_latent")
        self.check(L.sdfr_adam_step(p, g, self.m.data_ptr(), self.v.data_ptr(), self.step.data_ptr(),
                                    8 + self.Lz, 1e-3, 1e-2, 1e-3, 1e-2, int(self.shape_opt), d, st),
                   "sdfr_adam_step")
        self._inliers(L, p, d, st)

    def _iteration_one_launch(self, L, d, st, p, g, sdf):
        """The tail form with the render pair as ONE launch (``fused_render``): [decoder] -> render forward + backward +
        the sampler's blocks (``sdfr_render_step_fused_l1_pc``; the depth term's sums come out unscaled, beside the view's
        overlap count) -> [decoder VJP on  point-cloud volume + k depth volume, which it clears afterwards] -> the tail
        (multiplies the pose sums by k, writes the depth loss, resets the count)."""
        plan = self.plan
        g_pc = plan._g_sdf_ring[0] if self.shape_opt else None
        self._sums_open = True     # (until the tail has been enqueued: __call__ re-zeroes after an interrupted iteration)
        plan.step_fused_l1_pc(sdf, self.pos_c, self.quat_c, self.inv_scale, self.scale_v, self.cfg["threshold"],
                              self.target, self.points, self.offsets, self.max_pts, self.ws_pc,
                              pc_weight=self.cfg["pc_weight"], g_sdf=g_pc)
        self.loss_depth = plan.loss
        ws = plan.workspace.data_ptr()
        cnt_off = L.sdfr_render_fused_view_count_offset(self.V, self.H)
        t_mid = None
        if self.shape_opt:
            t_mid = ctypes.c_void_p()
            self.check(L.sdfr_decoder_backward_latent_deferred_scaled(
                self.dec._h, self.latent.data_ptr(), self.tape.data_ptr(), g_pc.data_ptr(), plan.g_depth.data_ptr(),
                self.V, ws + cnt_off, self.cfg["depth_weight"], self.ws_dec.data_ptr(), self.ws_dec.numel(), st,
                ctypes.byref(t_mid)), "sdfr_decoder_backward_latent_deferred_scaled")
        con = self.pc_source is not None
        self.check(L.sdfr_loop_tail_fused(
            p, g, self.m.data_ptr(), self.v.data_ptr(), self.step.data_ptr(), 8 + self.Lz, 1e-3, 1e-2, 1e-3, 1e-2,
            int(self.shape_opt), self.cam_pos.data_ptr(), self.cam_quat.data_ptr(), self.V, ws, plan.partials_offset,
            cnt_off, L.sdfr_render_fused_tile_loss_offset(self.R, self.V, self.W, self.H), self.cfg["depth_weight"],
            plan.loss.data_ptr(), self.W, self.H, self.ws_pc.data_ptr(), self.offsets.data_ptr(), self.max_pts,
            self.pos_c.data_ptr(), self.quat_c.data_ptr(), self.inv_scale.data_ptr(), self.scale_v.data_ptr(),
            self.loss_pc.data_ptr(), self.pc_source.data_ptr() if con else None,
            self.pc_target.data_ptr() if con else None, self.pc_weight if con else 0.0,
            self.loss_con.data_ptr() if con else None, self.dec._h if t_mid is not None else None, t_mid,
            self.tape.data_ptr() if self.fc_in_tail else None, self.arrivals.data_ptr() if self.fc_in_tail else None,
            d, st), "sdfr_loop_tail_fused")
        self._sums_open = False
        self._inliers(L, p, d, st)

    # ---- the records form: head | exchange | tail (the loop sharded over ranks) -----------------------------------

    def _head(self):
        """This rank's share of simple_setup.py:408-446: the (replicated) decoder, the render / compare / sample of its
        own views with both backward passes, and its views' exchange records."""
        from .differentiable_renderer import VIEW_RECORD_FLOATS
        L, d, st = self.L, self.dev.index, self._stream()
        if self.shape_opt:
            self._decode(st, True)
        sdf = self.sdf[0, 0]
        self.plan.ring_reset()       # always the plan's first volume: the bucket's address is part of captured graphs
        # (views without an observed point: their sampler blocks leave at once, their point-cloud loss is the NaN of
        # torch.mean over nothing, simple_setup.py:144, and they contribute no gradient -- as in the reference)
        have_pts = True
        self.plan.forward_l1(sdf, self.pos_c, self.quat_c, self.inv_scale, self.cfg["threshold"], self.target,
                             prepare_backward=True, defer_loss=self.defer_loss)
        g_sdf = self.plan.backward_l1_pc(self.target, sdf, self.pos_c, self.quat_c, self.inv_scale, self.scale_v,
                                         self.points, self.offsets, self.max_pts, self.ws_pc,
                                         weight=self.cfg["depth_weight"], pc_weight=self.cfg["pc_weight"])
        if self.shape_opt and self.exchange == "latent":
            # the whole VJP on this rank's volume: d loss / d latent is what travels
            self.check(L.sdfr_decoder_backward_latent(self.dec._h, self.latent.data_ptr(), self.tape.data_ptr(),
                                                      g_sdf.data_ptr(), 1, self.grads.data_ptr() + 32,
                                                      self.ws_dec.data_ptr(), self.ws_dec.numel(), st),
                       "sdfr_decoder_backward_latent")
        elif self.big_exchange and self.det:
            self.xbuf[:self.R ** 3].copy_(self.plan.g_sdf_fixed().view(-1))
        self.check(L.sdfr_loop_view_records(
            self.plan.workspace.data_ptr(), self.plan.partials_offset, self.W, self.H, self.sdf_grad_mode,
            self.ws_pc.data_ptr() if have_pts else None, 1 if have_pts else 0,
            self.offsets.data_ptr() if have_pts else None, self.max_pts, self.quat_c.data_ptr(),
            self.plan.loss.data_ptr(), self.view_begin, self.V, self.V_all, self.records.data_ptr(), d, st),
            "sdfr_loop_view_records")
        if self.track_inliers and self.owns_last:
            # :463-470 -- the LAST view's input and the estimate rendered before this step
            self.check(L.sdfr_inlier_counts_record(
                self.target[self.V - 1].data_ptr(), self.plan.depth[self.V - 1].data_ptr(), self.W, self.H, self.rel_thr,
                self.inlier_counts.data_ptr(), self.records.data_ptr() + 4 * VIEW_RECORD_FLOATS * (self.V_all - 1), d, st),
                "sdfr_inlier_counts_record")

    def _exchange(self):
        """The iteration's one collective (RCCL over xGMI with backend "nccl"); nothing without a group."""
        from .parallel import allreduce_bucket
        if self.group is not None:
            # as integers only where that is a sum: the int64 volume, and records (non-zero on one rank each) -- not
            # the latent gradients of exchange="latent", which every rank contributes to
            with_g = self.shape_opt and self.exchange == "latent"
            allreduce_bucket(self.xbuf, self.group, integer=self.det and not with_g)

    def _tail(self):
        """simple_setup.py:448-462 on every rank alike: decoder VJP of the summed volume, the chain over all views'
        records, point constraint, Adam, renormalisation -- and this rank's view poses for the next iteration."""
        from .differentiable_renderer import VIEW_RECORD_FLOATS
        L, d, st = self.L, self.dev.index, self._stream()
        p, g = self.params.data_ptr(), self.grads.data_ptr()
        t_mid = None
        if self.big_exchange:
            g_sdf = self.plan.g_sdf
            if self.det:
                self.check(L.sdfr_fixed_to_float(self.xbuf.data_ptr(), self.R ** 3, g_sdf.data_ptr(), d, st),
                           "sdfr_fixed_to_float")
            t_mid = ctypes.c_void_p()
            self.check(L.sdfr_decoder_backward_latent_deferred(
                self.dec._h, self.latent.data_ptr(), self.tape.data_ptr(), g_sdf.data_ptr(), self.ws_dec.data_ptr(),
                self.ws_dec.numel(), st, ctypes.byref(t_mid)), "sdfr_decoder_backward_latent_deferred")
        con = self.pc_source is not None
        self.check(L.sdfr_loop_tail_records(
            p, g, self.m.data_ptr(), self.v.data_ptr(), self.step.data_ptr(), 8 + self.Lz, 1e-3, 1e-2, 1e-3, 1e-2,
            int(self.shape_opt), self.cam_pos_all.data_ptr(), self.cam_quat_all.data_ptr(), self.V_all,
            self.view_begin, self.V, self.records.data_ptr(), self.pos_c.data_ptr(), self.quat_c.data_ptr(),
            self.inv_scale.data_ptr(), self.scale_v.data_ptr(),
            self.pc_source.data_ptr() if con else None, self.pc_target.data_ptr() if con else None,
            self.pc_weight if con else 0.0, self.loss_con.data_ptr() if con else None,
            self.dec._h if t_mid is not None else None, t_mid, d, st), "sdfr_loop_tail_records")
        if self.track_inliers:
            self.check(L.sdfr_inlier_update_record(
                self.records.data_ptr() + 4 * VIEW_RECORD_FLOATS * (self.V_all - 1), self.step.data_ptr(),
                self.inlier_history.data_ptr(), self.max_history, self.best_state.data_ptr(), p, 8 + self.Lz,
                self.best_params.data_ptr(), d, st), "sdfr_inlier_update_record")

    def view_losses(self):
        """(depth loss, point-cloud loss) of every view of the last iteration, (V_all,) each -- on every rank after
        the exchange of the records form; this process's own views otherwise."""
        if self.records_form:
            from .differentiable_renderer import VIEW_RECORD_FLOATS
            rec = self.records.view(self.V_all, VIEW_RECORD_FLOATS)
            return rec[:, 16], rec[:, 17]
        return self.loss_depth, self.loss_pc

    def _run_records(self, n_iter, use_graph, history):
        state = (self._state,)
        if self.group is None:
            return self._run_records_single(n_iter, use_graph, history, state)
        if use_graph and self.graph is None:
            # warm up on a side stream (lazy module loads; every rank takes part in the exchange), restore, capture:
            # the collective stays outside the graphs -- head | all-reduce | tail+head | all-reduce | ... | tail
            saved = [t.clone() for t in state]
            side = torch.cuda.Stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._head()
                self._exchange()
                self._tail()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            for t, c in zip(state, saved):
                t.copy_(c)
            # (thread-local capture: the process group's watchdog thread may query events while this thread captures)
            mode = dict(capture_error_mode="thread_local") if self.group is not None else {}
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, **mode):
                self._head()
            self.graph_tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_tail, **mode):
                self._tail()
            self.graph_many = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_many, **mode):
                self._tail()
                self._head()
            for t, c in zip(state, saved):
                t.copy_(c)
            if self.graph_collective:
                self._capture_with_collective(mode)
                for t, c in zip(state, saved):
                    t.copy_(c)
            self._poses_to_views(self._stream())
            self._keep_graphs()
        if use_graph and self.graph_whole_one is not None:
            # the collective is a node of the graphs: whole iterations replay, nothing is issued in between
            done = 0
            if history is None and self.graph_whole is not None:
                for _ in range(n_iter // self.graph_iterations):
                    self.graph_whole.replay()
                done = n_iter - n_iter % self.graph_iterations
            for _ in range(n_iter - done):
                self.graph_whole_one.replay()
                if history is not None:
                    self._record_history(history)
            return
        fused = use_graph and history is None
        for it in range(n_iter):
            if not use_graph:
                self._head()
            elif not fused or it == 0:
                self.graph.replay()
            self._exchange()
            if not use_graph:
                self._tail()
            elif fused and it + 1 < n_iter:
                self.graph_many.replay()     # this iteration's tail and the next one's head
            else:
                self.graph_tail.replay()
            if history is not None:
                self._record_history(history)

    def _capture_with_collective(self, mode):
        """the experiment of ``graph_collective``: head | all-reduce | tail as ONE captured sequence"""
        import torch.distributed as dist
        try:
            if dist.get_backend(self.group) != "nccl":
                raise RuntimeError(f"backend {dist.get_backend(self.group)!r}: only RCCL's collectives are stream work")
            one = torch.cuda.CUDAGraph()
            with torch.cuda.graph(one, **mode):
                self._head()
                self._exchange()
                self._tail()
            many = None
            if self.graph_iterations > 1:
                many = torch.cuda.CUDAGraph()
                with torch.cuda.graph(many, **mode):
                    for _ in range(self.graph_iterations):
                        self._head()
                        self._exchange()
                        self._tail()
            self.graph_whole_one, self.graph_whole = one, many
        except Exception as e:      # capture refused: the two-graph form stays
            torch.cuda.synchronize(self.dev)
            self.graph_whole_one = self.graph_whole = None
            self.graph_collective_error = f"{type(e).__name__}: {e}"

    def _record_history(self, history):
        ld, lp = self.view_losses()
        history.append({"loss": (self.cfg["depth_weight"] * ld.sum() + self.cfg["pc_weight"] * lp.sum()
                                 + self.loss_con.sum()).clone(),
                        "loss_depth": ld.clone(), "loss_pc": lp.clone(),
                        "position": self.position.clone()[None], "orientation": self.orientation.clone()[None],
                        "scale": self.scale.clone(), "latent": self.latent.clone()[None]})

    def _run_records_single(self, n_iter, use_graph, history, state):
        """the records form without a process group: no collective, so whole iterations are captured -- one graph per
        iteration, and one of ``graph_iterations`` iterations, as in the tail form"""
        if use_graph and self.graph is None:
            saved = [t.clone() for t in state]
            side = torch.cuda.Stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._head()
                self._tail()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            for t, c in zip(state, saved):
                t.copy_(c)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._head()
                self._tail()
            if self.graph_iterations > 1:
                self.graph_many = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_many):
                    for _ in range(self.graph_iterations):
                        self._head()
                        self._tail()
            for t, c in zip(state, saved):
                t.copy_(c)
            self._poses_to_views(self._stream())
            self._keep_graphs()
        done = 0
        if use_graph and history is None and self.graph_many is not None:
            for _ in range(n_iter // self.graph_iterations):
                self.graph_many.replay()
            done = n_iter - n_iter % self.graph_iterations
        for _ in range(n_iter - done):
            if use_graph:
                self.graph.replay()
            else:
                self._head()
                self._tail()
            if history is not None:
                self._record_history(history)

    def _tail_form(self) -> bool:
        return (self.merge_tail and self.fuse_depth_loss and self.defer_pose and self.max_pts > 0
                and 8 + self.Lz <= 256)

    def _poses_to_views(self, st):
        p = self.params.data_ptr()
        self.check(self.L.sdfr_pose_to_views(p, p + 12, p + 28, self.cam_pos.data_ptr(), self.cam_quat.data_ptr(),
                                             self.V, self.pos_c.data_ptr(), self.quat_c.data_ptr(),
                                             self.inv_scale.data_ptr(), self.scale_v.data_ptr(), self.dev.index, st),
                   "sdfr_pose_to_views")

    def _inliers(self, L, p, d, st):
        if self.track_inliers:
            # :463-470 -- the LAST view's input and the estimate rendered before this step, against the
            # parameters after it
            est_last = self.plan.depth[self.V - 1]
            self.check(L.sdfr_inlier_ratio(self.target[self.V - 1].data_ptr(), est_last.data_ptr(), self.W, self.H,
                                           self.rel_thr, self.step.data_ptr(), self.inlier_counts.data_ptr(),
                                           self.inlier_history.data_ptr(), self.max_history,
                                           self.best_state.data_ptr(), p, 8 + self.Lz,
                                           self.best_params.data_ptr(), d, st), "sdfr_inlier_ratio")

    def __call__(self, position, orientation, scale, latent, use_graph: bool = True,
                 history: Optional[List] = None):
        """Run config['max_iterations'] iterations from the given initial estimate; returns
        (position (1,3), orientation (1,4), scale (1,), latent (1,L))."""
        if not self.bound:
            raise RuntimeError("no observation bound: call rebind(depth_images, ...) first")
        with torch.no_grad():
            self.params[0:3] = position.reshape(3)
            self.params[3:7] = orientation.reshape(4)
            self.params[7:8] = scale.reshape(1)
            self.params[8:] = latent.reshape(-1)
            self._state_zero.zero_()     # Adam's moments and step count, inlier history, counts and best-so-far state
            self.grads.zero_()
            if self.fused_render and self._sums_open:   # an iteration was abandoned between its render and its tail
                self._zero_fused_sums()
        if not self.shape_opt:
            self._decode(self._stream(), False)
        n_iter = self.cfg["max_iterations"]
        if self.records_form:
            self._poses_to_views(self._stream())
            self._run_records(n_iter, use_graph, history)
            return (self.position.clone()[None], self.orientation.clone()[None], self.scale.clone(),
                    self.latent.clone()[None])
        if self._tail_form():
            self._prime()   # every later iteration gets its view poses from the tail before it
        if use_graph and self.graph is None:
            # warm up on a side stream (lazy module loads), restore the state, then capture
            state = (self._state,)
            saved = [t.clone() for t in state]
            s = torch.cuda.Stream(self.dev)
            s.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(s):
                self.iteration()
            torch.cuda.current_stream(self.dev).wait_stream(s)
            for t, c in zip(state, saved):
                t.copy_(c)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.iteration()
            if self.graph_iterations > 1:
                self.graph_many = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_many):
                    for _ in range(self.graph_iterations):
                        self.iteration()
            for t, c in zip(state, saved):
                t.copy_(c)   # the capture itself does not execute, but keep the state explicit
            if self._tail_form():
                self._prime()   # the warm-up's tail left the poses of ITS updated parameters
            self._keep_graphs()
        done = 0
        if use_graph and history is None and self.graph_many is not None:
            for _ in range(n_iter // self.graph_iterations):
                self.graph_many.replay()
            done = n_iter - n_iter % self.graph_iterations
        for _ in range(n_iter - done):
            if use_graph:
                self.graph.replay()
            else:
                self.iteration()
            if history is not None:
                history.append({"loss": (self.cfg["depth_weight"] * self.loss_depth.sum()
                                         + self.cfg["pc_weight"] * self.loss_pc.sum()
                                         + self.loss_con.sum()).clone(),
                                "position": self.position.clone()[None],
                                "orientation": self.orientation.clone()[None],
                                "scale": self.scale.clone(), "latent": self.latent.clone()[None]})
        # "best_inlier_ratio": the reference returns the parameter tensors themselves, i.e. the last
        # iterate (_BestEstimate); the snapshot taken at the best ratio is in best_estimate()
        return (self.position.clone()[None], self.orientation.clone()[None], self.scale.clone(),
                self.latent.clone()[None])

    def best_estimate(self):
        """(ratio, 1-based iteration, (position, orientation, scale, latent)) at the best inlier ratio of the
        last run (strictly greater wins, the first counts: simple_setup.py:203-210); needs track_inliers."""
        ratio, it, has = self.best_state.tolist()
        if not has:
            return None
        bp = self.best_params
        return ratio, int(it), (bp[0:3].clone()[None], bp[3:7].clone()[None], bp[7:8].clone(), bp[8:].clone()[None])


class MultiObjectRenderAndCompare:
    """K estimates optimised SIDE BY SIDE: the K detected objects of one frame -- each with its own depth image (the
    frame masked by the object's instance mask), pose, scale, latent and Adam state -- go through ONE launch sequence
    per iteration.  The reference calls its pipeline once per object, one after the other (simple_setup.py:213-225), and
    a single estimate's iteration is a chain of ~13 dependent launches that leaves most of an MI355X idle (C5: 0.11 ms
    per iteration whatever the object); here the decoder runs on K latents at once, the renderer and the sampler on K
    views with one SDF each (``sdf_view_stride = R^3``), the decoder's VJP on the K gradient volumes, and the tail is K
    workgroups (``sdfr_loop_tail_objects``).  Same arithmetic per object as :class:`FusedRenderAndCompare` (one view per
    object, no point constraint, no inlier bookkeeping, "last_iteration" results); the objects share the camera.

    Everything is allocated here; ``rebind`` takes the next frame's K images (point clouds at capacity, counts on the
    device); ``__call__`` replays captured graphs."""

    def __init__(self, decoder, camera: Camera, config: Dict, objects: int, shape_optimization: bool = True,
                 device="cuda", graph_iterations: int = 5, sdf_grad_mode: int = 0):
        from . import _lib
        from .differentiable_renderer import BatchRenderPlan
        self.L, self.check = _lib.lib(), _lib.check
        self.dec, self.cam, self.cfg = decoder, camera, config
        self.dev = torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        K = self.K = int(objects)
        if K < 1:
            raise ValueError("objects must be >= 1")
        self.shape_opt = bool(shape_optimization)
        self.graph_iterations = max(1, int(graph_iterations))
        H, W = int(camera.height), int(camera.width)
        self.H, self.W = H, W
        f32 = dict(dtype=torch.float32, device=self.dev)
        i32 = dict(dtype=torch.int32, device=self.dev)
        u8 = dict(dtype=torch.uint8, device=self.dev)
        R = self.R = decoder._volume_size
        self.Lz = decoder.latent_size
        n = self.n = 8 + self.Lz
        if n > 256:
            raise ValueError("the tail handles parameter vectors of up to 256 entries")
        self.params = torch.zeros((K, n), **f32)
        self._zeroed = torch.zeros((3, K, n), **f32)          # Adam's moments and the gradients: one fill resets a run
        self.m, self.v, self.grads = self._zeroed[0], self._zeroed[1], self._zeroed[2]
        self.step = torch.zeros(K, **i32)
        self.z = torch.zeros((K, self.Lz), **f32)             # the latents, packed for the decoder
        self.g_z = torch.zeros((K, self.Lz), **f32)
        self.cam_pos = torch.zeros((1, 3), **f32)
        self.cam_quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]], **f32)
        self.target = torch.zeros((K, H, W), **f32)
        self.max_pts = W * H
        self.points = torch.zeros((K * W * H, 3), **f32)
        self.offsets = torch.zeros(K + 1, **i32)
        self.counts = torch.zeros(K, **i32)
        self.ws_points = torch.empty(max(self.L.sdfr_depth_points_workspace_bytes(K, W, H), 256), **u8)
        self.sdf_grad_mode = int(sdf_grad_mode)     # SDF_GRAD_EXACT / SDF_GRAD_CUDA_COMPAT
        self.plan = BatchRenderPlan(R, K, camera, device=self.dev, per_view_sdf=True, close_views=False,
                                    sdf_grad_mode=self.sdf_grad_mode)
        self.pos_c = torch.empty((K, 3), **f32)
        self.quat_c = torch.empty((K, 4), **f32)
        self.inv_scale = torch.empty((K,), **f32)
        self.scale_v = torch.empty((K,), **f32)
        self.loss_pc = torch.zeros((K,), **f32)
        self.sdf = torch.empty((K, 1, R, R, R), **f32)
        self.tape = torch.empty(max(self.L.sdfr_decoder_tape_bytes(decoder._h, K), 256), **u8)
        self.ws_dec = torch.empty(max(self.L.sdfr_decoder_workspace_bytes(decoder._h, K),
                                      self.L.sdfr_decoder_backward_workspace_bytes(decoder._h, K), 256), **u8)
        self.ws_pc = torch.empty(max(self.L.sdfr_pc_loss_backward_workspace_bytes(K, self.max_pts), 256), **u8)
        self.defer_loss = K < 32
        self.graph = self.graph_many = None
        self.bound = False

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def rebind(self, depth_images: torch.Tensor, camera_position: Optional[torch.Tensor] = None,
               camera_orientation: Optional[torch.Tensor] = None, masks: Optional[torch.Tensor] = None,
               far_field: Optional[float] = None) -> "MultiObjectRenderAndCompare":
        """depth_images (K,H,W): one image per object (``masks`` / ``far_field``: preprocessed in place on the way,
        simple_setup.py:671-693).  camera_position (3,) / camera_orientation (4,): the one camera all objects are seen
        from (default: the origin)."""
        if tuple(depth_images.shape) != (self.K, self.H, self.W):
            raise ValueError(f"depth_images must have shape {(self.K, self.H, self.W)}, got {tuple(depth_images.shape)}")
        with torch.no_grad():
            if masks is not None or far_field is not None:
                preprocess_depth(depth_images, masks, far_field, copy_to=(self.target, 0, self.K))
            else:
                self.target.copy_(depth_images)
            if camera_position is None:
                self.cam_pos.zero_()
            else:
                self.cam_pos.copy_(camera_position.reshape(1, 3))
            if camera_orientation is None:
                self.cam_quat.copy_(torch.tensor([[0.0, 0.0, 0.0, 1.0]]))
            else:
                self.cam_quat.copy_(camera_orientation.reshape(1, 4))
            fx, fy, cx0, cy0, _ = self.cam.get_pinhole_camera_parameters(0.0)
            self.check(self.L.sdfr_depth_to_points_resident(
                self.target.data_ptr(), self.K, self.W, self.H, 1, 1.0 / fx, 1.0 / fy, cx0, cy0, self.counts.data_ptr(),
                self.offsets.data_ptr(), self.ws_points.data_ptr(), self.ws_points.numel(), self.points.data_ptr(),
                self.dev.index, self._stream()), "sdfr_depth_to_points_resident")
        self.bound = True
        return self

    def _decode(self, st, with_tape):
        # (self.z holds the current latents, packed: sdfr_pose_to_views_objects at the start of a run, the tail after)
        self.check(self.L.sdfr_decoder_forward(self.dec._h, self.z.data_ptr(), self.K, 0, self.sdf.data_ptr(),
                                               self.tape.data_ptr() if with_tape else None, self.ws_dec.data_ptr(),
                                               self.ws_dec.numel(), st), "sdfr_decoder_forward")

    def _poses_to_views(self, st):
        """every object's pose in the camera's frame (simple_setup.py:411, :424-430), once per run: the tail leaves the
        NEXT iteration's"""
        self.check(self.L.sdfr_pose_to_views_objects(
            self.params.data_ptr(), self.n, self.K, self.cam_pos.data_ptr(), self.cam_quat.data_ptr(), 1,
            self.pos_c.data_ptr(), self.quat_c.data_ptr(), self.inv_scale.data_ptr(), self.scale_v.data_ptr(),
            self.z.data_ptr(), self.dev.index, st), "sdfr_pose_to_views_objects")

    def iteration(self):
        L, d, st = self.L, self.dev.index, self._stream()
        if self.shape_opt:
            self._decode(st, True)
        sdf = self.sdf[:, 0]
        self.plan.forward_l1(sdf, self.pos_c, self.quat_c, self.inv_scale, self.cfg["threshold"], self.target,
                             prepare_backward=True, defer_loss=self.defer_loss)
        g_sdf = self.plan.backward_l1_pc(self.target, sdf, self.pos_c, self.quat_c, self.inv_scale, self.scale_v,
                                         self.points, self.offsets, self.max_pts, self.ws_pc,
                                         weight=self.cfg["depth_weight"], pc_weight=self.cfg["pc_weight"])
        t_mid = ctypes.c_void_p()
        if self.shape_opt:
            # the VJP without its last stage: workgroup k of the tail finishes object k's (one launch and a copy less)
            self.check(L.sdfr_decoder_backward_latent_deferred_batch(
                self.dec._h, self.z.data_ptr(), self.tape.data_ptr(), g_sdf.data_ptr(), self.K, self.ws_dec.data_ptr(),
                self.ws_dec.numel(), st, ctypes.byref(t_mid)), "sdfr_decoder_backward_latent_deferred_batch")
        self.check(L.sdfr_loop_tail_objects(
            self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.step.data_ptr(),
            self.n, self.K, 1e-3, 1e-2, 1e-3, 1e-2, int(self.shape_opt), self.cam_pos.data_ptr(), self.cam_quat.data_ptr(),
            1, self.plan.workspace.data_ptr(), self.plan.partials_offset, self.W, self.H, self.ws_pc.data_ptr(),
            self.offsets.data_ptr(), self.max_pts, self.pos_c.data_ptr(), self.quat_c.data_ptr(),
            self.inv_scale.data_ptr(), self.scale_v.data_ptr(), self.loss_pc.data_ptr(),
            self.dec._h if self.shape_opt else None, t_mid if self.shape_opt else None, self.z.data_ptr(), d, st),
            "sdfr_loop_tail_objects")

    def view_losses(self):
        """(depth loss, point-cloud loss) of every object in the last iteration, (K,) each"""
        return self.plan.loss, self.loss_pc

    def __call__(self, position, orientation, scale, latent, use_graph: bool = True, history: Optional[List] = None):
        """position (K,3), orientation (K,4), scale (K,), latent (K,L): the initial estimates (or ``params=`` rows from
        ResidentInit).  Returns the optimised (position (K,3), orientation (K,4), scale (K,), latent (K,L))."""
        if not self.bound:
            raise RuntimeError("no observation bound: call rebind(depth_images, ...) first")
        K = self.K
        with torch.no_grad():
            self.params[:, 0:3] = position.reshape(K, 3)
            self.params[:, 3:7] = orientation.reshape(K, 4)
            self.params[:, 7] = scale.reshape(K)
            self.params[:, 8:] = latent.reshape(K, self.Lz)
            self._zeroed.zero_()
            self.step.zero_()
        st = self._stream()
        self._poses_to_views(st)        # (also packs the latents for the decoder)
        if not self.shape_opt:
            self._decode(st, False)
        n_iter = self.cfg["max_iterations"]
        if use_graph and self.graph is None:
            saved = [t.clone() for t in (self.params, self._zeroed, self.step)]
            side = torch.cuda.Stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self.iteration()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.iteration()
            if self.graph_iterations > 1:
                self.graph_many = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_many):
                    for _ in range(self.graph_iterations):
                        self.iteration()
            for t, c in zip((self.params, self._zeroed, self.step), saved):
                t.copy_(c)
            self._poses_to_views(self._stream())     # (the warm-up's tail left the poses of ITS updated parameters)
        done = 0
        if use_graph and history is None and self.graph_many is not None:
            for _ in range(n_iter // self.graph_iterations):
                self.graph_many.replay()
            done = n_iter - n_iter % self.graph_iterations
        for _ in range(n_iter - done):
            if use_graph:
                self.graph.replay()
            else:
                self.iteration()
            if history is not None:
                ld, lp = self.view_losses()
                history.append({"loss_depth": ld.clone(), "loss_pc": lp.clone(), "params": self.params.clone()})
        p = self.params
        return p[:, 0:3].clone(), p[:, 3:7].clone(), p[:, 7].clone(), p[:, 8:].clone()
