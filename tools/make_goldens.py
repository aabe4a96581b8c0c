#!/usr/bin/env python3
"""Capture golden vectors by IMPORTING the reference (dev container only).

Runs the reference's own importable code on seeded inputs and writes
inputs + expected outputs as small .npz data fixtures under tests/golden/:

  * sdfest/differentiable_renderer/simple_renderer.py  (numpy float64 twin of
    the CUDA renderer; loaded by file path because the package __init__
    imports open3d and JIT-builds CUDA)                       -> render_*.npz
  * sdfest/estimation/losses.py::pc_loss (torch, autograd)    -> pc_loss.npz
  * sdfest/vae/sdf_vae.py::SDFDecoder + tests/.../mug.pt       -> decoder_mug.npz,
                                                                 mug_decoder_weights.npz
  * sdfest/initialization/quaternion_utils.py                 -> quaternion.npz

Nothing of the reference's source travels: only numbers.  /root/reference does
not exist on the GPU box; tests read only the .npz files.

Usage:  python tools/make_goldens.py [--ref /root/reference] [--only NAME]
"""
import argparse
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import blobs_sdf, sphere_sdf  # noqa: E402  (input generators only)


def load_simple_renderer(ref):
    if not hasattr(np, "float"):
        np.float = float  # alias removed in numpy>=1.24; reference pinned 1.22
    path = os.path.join(ref, "sdfest/differentiable_renderer/simple_renderer.py")
    spec = importlib.util.spec_from_file_location("ref_simple_renderer", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_render(sr, sdf, W, H, fov, thr, p, q, inv_scale, g_images):
    """Run the numpy twin; returns depth, count image, 8 derivative images and
    the reduced gradients for every upstream image in g_images (the reduction
    is sdf_renderer.py:242-261 restated: sum(derivative * g_image))."""
    obj = sr.SDFObject(np.asarray(sdf, dtype=np.float64))
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        depth, der = sr.render_depth(obj, W, H, fov, "d", thr, np.array(p, dtype=np.float64),
                                     np.array(q, dtype=np.float64), float(inv_scale))
        count, _ = sr.render_depth(obj, W, H, fov, "c", thr, np.array(p, dtype=np.float64),
                                   np.array(q, dtype=np.float64), float(inv_scale))
    names = ["x", "y", "z", "qx", "qy", "qz", "qw", "s_inv"]
    zero = np.zeros_like(depth)
    dimg = np.stack([der[n] if n in der else zero for n in names], axis=-1)
    out = {"depth": depth, "count": count.astype(np.int32), "dimg": dimg}
    sdf_der = der["sdf"] if "sdf" in der else {}
    keys = sorted(sdf_der.keys())
    idx = np.array(keys, dtype=np.int32).reshape(-1, 3)
    out["gsdf_idx"] = idx
    for gi, g in enumerate(g_images):
        out[f"g{gi}_image"] = g
        out[f"g{gi}_pose"] = np.array([np.sum(dimg[..., k] * g) for k in range(8)])
        out[f"g{gi}_gsdf"] = np.array([np.sum(sdf_der[k] * g) for k in keys])
    return out


def fov_to_intrinsics(W, H, fov):
    f = W / np.tan(fov * np.pi / 180.0 / 2.0) / 2
    return f, f, W / 2, H / 2


def make_render(ref):
    sr = load_simple_renderer(ref)
    cases = {}
    rng = np.random.default_rng(7)

    def add(name, sdf_kind, W, H, fov, thr, p, q, isc):
        sdf = {"sphere": lambda: sphere_sdf(0.5), "blobs0": lambda: blobs_sdf(0)}[sdf_kind]()
        g = [np.ones((H, W)), rng.uniform(-1, 1, (H, W))]
        r = ref_render(sr, sdf, W, H, fov, thr, p, q, isc, g)
        fx, fy, cx, cy = fov_to_intrinsics(W, H, fov)
        r.update(sdf_kind=sdf_kind, W=W, H=H, fov=fov, thr=thr, p=np.array(p, float),
                 q=np.array(q, float), inv_scale=float(isc), fx=fx, fy=fy, cx=cx, cy=cy)
        cases[name] = r
        print(f"  {name}: hits={int((r['depth'] > 0).sum())} steps={int(r['count'].sum())} "
              f"voxels={len(r['gsdf_idx'])}")

    ident = (0, 0, 0, 1)
    # G1 analytic sphere
    add("g1_sphere_32x24", "sphere", 32, 24, 90, 0.01, (0, 0, -2), ident, 1.0)
    add("g1_sphere_64x48", "sphere", 64, 48, 90, 0.01, (0, 0, -2), ident, 1.0)
    # G2 = C1
    add("g2_blobs_c1_160x120", "blobs0", 160, 120, 90, 0.005, (0, 0, -1.5), ident, 2.0)
    # G3 random non-axis-aligned poses
    prng = np.random.default_rng(3)
    for i in range(4):
        q = prng.normal(size=4)
        q /= np.linalg.norm(q)
        p = (prng.uniform(-0.3, 0.3), prng.uniform(-0.2, 0.2), -prng.uniform(1.2, 2.0))
        s = prng.uniform(0.4, 0.6)
        add(f"g3_pose{i}_64x48", "blobs0", 64, 48, 90, 0.005, p, q, 1.0 / s)
    # G4 edge cases
    add("g4_parallel_slab_33x25", "sphere", 33, 25, 90, 0.01, (0, 0, -2), ident, 1.0)
    add("g4_offscreen_32x24", "sphere", 32, 24, 90, 0.01, (10, 0, -2), ident, 1.0)
    add("g4_behind_32x24", "sphere", 32, 24, 90, 0.01, (0, 0, 2), ident, 1.0)
    add("g4_camera_inside_32x24", "blobs0", 32, 24, 90, 0.01, (0.05, -0.1, -0.9), ident, 1.0)
    add("g4_partly_out_48x32", "blobs0", 48, 32, 60, 0.005, (0.9, 0.5, -1.3),
        tuple(np.array([1.0, 2.0, 3.0, 4.0]) / np.sqrt(30.0)), 1.8)
    for name, r in cases.items():
        np.savez_compressed(os.path.join(OUT, f"render_{name}.npz"), **r)


def make_pc_loss(ref):
    import torch
    sys.path.insert(0, ref)
    from sdfest.estimation import losses
    rng = np.random.default_rng(11)
    sdf = blobs_sdf(0)
    out = {}
    for i, (pos, quat, scale) in enumerate([
        ((0.02, -0.01, -0.5), (0.0, 0.0, 0.0, 1.0), 0.3),
        ((0.1, 0.05, -0.8), (0.3, -0.5, 0.2, 0.9), 0.25),   # un-normalised on purpose
        ((0.0, 0.0, -1.0), (-0.7, 0.1, 0.6, -0.3), 0.5),
    ]):
        M = 1000
        # points: mostly inside the volume, some well outside, a few near the faces
        pts = np.array(pos) + rng.uniform(-1.25, 1.25, (M, 3)) * scale
        g_out = rng.uniform(-1, 1, M)
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            t_pts = torch.tensor(pts, dtype=dt)
            t_pos = torch.tensor(pos, dtype=dt, requires_grad=True)
            t_q = torch.tensor(quat, dtype=dt, requires_grad=True)
            t_s = torch.tensor(scale, dtype=dt, requires_grad=True)
            t_sdf = torch.tensor(sdf, dtype=dt, requires_grad=True)
            val = losses.pc_loss(t_pts, t_pos, t_q, t_s, t_sdf)
            val.backward(torch.tensor(g_out, dtype=dt))
            gs = t_sdf.grad.numpy()
            nz = np.argwhere(gs != 0).astype(np.int32)
            out[f"c{i}_{tag}_value"] = val.detach().numpy()
            out[f"c{i}_{tag}_gpos"] = t_pos.grad.numpy()
            out[f"c{i}_{tag}_gquat"] = t_q.grad.numpy()
            out[f"c{i}_{tag}_gscale"] = t_s.grad.numpy()
            out[f"c{i}_{tag}_gsdf_idx"] = nz
            out[f"c{i}_{tag}_gsdf_val"] = gs[tuple(nz.T)]
        out[f"c{i}_points"] = pts
        out[f"c{i}_gout"] = g_out
        out[f"c{i}_pos"] = np.array(pos)
        out[f"c{i}_quat"] = np.array(quat)
        out[f"c{i}_scale"] = np.array(scale)
        n_in = int((out[f"c{i}_f64_value"] != 0).sum())
        print(f"  pc_loss case {i}: {n_in}/{M} points inside")
    out["n_cases"] = 3
    np.savez_compressed(os.path.join(OUT, "pc_loss.npz"), **out)


def decoder_probe_G(n=64):
    """Deterministic upstream gradient for the decoder VJP check (formula, so the
    test can rebuild it without storing 1 MiB)."""
    a = np.arange(n, dtype=np.float64)
    X, Y, Z = np.meshgrid(a, a, a, indexing="ij")
    return (np.sin(0.37 * X + 0.11) * np.cos(0.23 * Y - 0.4) + 0.5 * np.sin(0.05 * Z * X * 0.1 + 0.3 * Y)).astype(np.float32)


def make_decoder(ref):
    import torch
    import yaml
    sys.path.insert(0, ref)
    from sdfest.vae import sdf_vae
    cfg_path = os.path.join(ref, "tests/initilization/vae_model/mug.yaml")
    with open(cfg_path) as f:
        cfg = yaml.safe_load(f)
    vae = sdf_vae.SDFVAE(sdf_size=64, latent_size=cfg["latent_size"], encoder_dict=cfg["encoder"],
                         decoder_dict=cfg["decoder"], device="cpu")
    state = torch.load(os.path.join(ref, "tests/initilization/vae_model/mug.pt"),
                       map_location="cpu")
    vae.load_state_dict(state)
    vae.eval()
    # weights as plain numbers (decoder only: the encoder never runs at inference)
    weights = {k: v.numpy() for k, v in state.items() if k.startswith("decoder.")}
    np.savez_compressed(os.path.join(OUT, "mug_decoder_weights.npz"), **weights)
    print(f"  decoder tensors: {len(weights)}, params: {sum(v.size for v in weights.values())}")

    out = {"latent_size": cfg["latent_size"],
           "fc_out": np.array([l["out"] for l in cfg["decoder"]["fc_layers"]]),
           "conv_in_size": np.array([l["in_size"] for l in cfg["decoder"]["conv_layers"]]),
           "conv_cin": np.array([l["in_channels"] for l in cfg["decoder"]["conv_layers"]]),
           "conv_cout": np.array([l["out_channels"] for l in cfg["decoder"]["conv_layers"]]),
           "conv_k": np.array([l["kernel_size"] for l in cfg["decoder"]["conv_layers"]]),
           "conv_relu": np.array([int(l["relu"]) for l in cfg["decoder"]["conv_layers"]])}
    rng = np.random.default_rng(5)
    L = cfg["latent_size"]
    zs = [np.zeros(L)] + [np.eye(L)[i] for i in range(L)] + [rng.normal(size=L) for _ in range(3)]
    zs = np.array(zs, dtype=np.float32)
    out["z"] = zs
    G = decoder_probe_G()
    with torch.no_grad():
        full0 = vae.decode(torch.tensor(zs[:1]))[0, 0].numpy()
    out["z0_full"] = full0
    # intermediates for z=0 via forward hooks on the conv layers + fc layers
    inter = {}
    dec = vae.decoder
    hooks = []

    def fc_hook(i):
        def fn(module, args, output):
            inter[f"fc{i}_pre"] = output.detach().numpy().copy()
        return fn

    def conv_hook(i):
        def fn(module, args, output):
            inter[f"conv{i}_in"] = args[0].detach().numpy().copy()
            inter[f"conv{i}_pre"] = output.detach().numpy().copy()
        return fn

    for i, l in enumerate(dec._fc_layers):
        hooks.append(l.register_forward_hook(fc_hook(i)))
    for i, l in enumerate(dec._conv_layers):
        hooks.append(l.register_forward_hook(conv_hook(i)))
    with torch.no_grad():
        vae.decode(torch.tensor(zs[:1]))
    for h in hooks:
        h.remove()
    for k, v in inter.items():
        if v.size <= 120000:   # keep the fixture small: skip the 8x32^3 and 4x64^3 conv inputs
            out["z0_" + k] = v
    subs, stats, gz = [], [], []
    for i in range(len(zs)):
        zt = torch.tensor(zs[i:i + 1], requires_grad=True)
        o = vae.decode(zt)[0, 0]
        (o * torch.tensor(G)).sum().backward()
        on = o.detach().numpy()
        subs.append(on[::4, ::4, ::4].copy())
        stats.append([on.sum(dtype=np.float64), np.abs(on).sum(dtype=np.float64), on.min(), on.max()])
        gz.append(zt.grad.numpy()[0].copy())
    out["sub16"] = np.array(subs)
    out["stats"] = np.array(stats)
    out["grad_z"] = np.array(gz)
    # the same VJPs by the same module in float64 (vae.double()): what the fp32 figures are rounded versions of --
    # the comparand for the 1e-4 bound of the decoder's latent gradient (the fp32 ones differ from it by ~2e-5 themselves)
    import copy
    vae64 = copy.deepcopy(vae).double()
    gz64 = []
    for i in range(len(zs)):
        zt = torch.tensor(zs[i:i + 1], dtype=torch.float64, requires_grad=True)
        o = vae64.decode(zt)[0, 0]
        (o * torch.tensor(G, dtype=torch.float64)).sum().backward()
        gz64.append(zt.grad.numpy()[0].copy())
    out["grad_z_f64"] = np.array(gz64)
    out["G_sub16"] = G[::4, ::4, ::4]  # sanity check of the formula only
    np.savez_compressed(os.path.join(OUT, "decoder_mug.npz"), **out)
    print(f"  decoder z=0 range [{full0.min():.4f}, {full0.max():.4f}]")


def make_quaternion(ref):
    import torch
    sys.path.insert(0, ref)
    from sdfest.initialization import quaternion_utils as qu
    rng = np.random.default_rng(13)
    q1 = rng.normal(size=(16, 4)); q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
    q2 = rng.normal(size=(16, 4)); q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    pts = rng.normal(size=(16, 3))
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    np.savez_compressed(
        os.path.join(OUT, "quaternion.npz"), q1=q1, q2=q2, pts=pts,
        mul=qu.quaternion_multiply(t(q1), t(q2)).numpy(),
        apply=qu.quaternion_apply(t(q1), t(pts)).numpy(),
        inv=qu.quaternion_invert(t(q1)).numpy())


def make_loop_g7(ref, compat_only=False):
    """compat_only: write tests/golden/loop_g7_compat.npz instead (run D, see the end of this function).

    G7: a few iterations of the render-and-compare loop of
    estimation/simple_setup.py:381-470 assembled from IMPORTED reference pieces --
    the numpy twin (simple_renderer.render_depth, value "d") with the reduction of
    sdf_renderer.py:242-261, losses.pc_loss, losses.point_constraint_loss, quaternion_utils,
    SDFVAE.decode with the mug weights, torch.optim.Adam.  simple_setup.py and
    pointset_utils.py themselves cannot be imported here (open3d / the CUDA JIT build), so the
    loop statements, _compute_view_losses (:115-162), _compute_inlier_ratio (:177-188) and
    depth_to_pointcloud (pointset_utils.py:57-77) are the only lines restated.  Writes the
    parameter trajectory, the loss terms and the inlier ratios (last view's loop variables,
    :463-470) of two runs: A = 2 views with cameras, shape optimisation on;
    B = 1 view, point constraint, shape optimisation off."""
    import torch
    import yaml
    sys.path.insert(0, ref)
    from sdfest.estimation import losses
    from sdfest.initialization import quaternion_utils as qu
    from sdfest.vae import sdf_vae
    sr = load_simple_renderer(ref)
    with open(os.path.join(ref, "tests/initilization/vae_model/mug.yaml")) as f:
        cfg = yaml.safe_load(f)
    vae = sdf_vae.SDFVAE(sdf_size=64, latent_size=cfg["latent_size"], encoder_dict=cfg["encoder"],
                         decoder_dict=cfg["decoder"], device="cpu")
    vae.load_state_dict(torch.load(os.path.join(ref, "tests/initilization/vae_model/mug.pt"),
                                   map_location="cpu"))
    vae.eval()
    FOV, THR = 90.0, 0.005
    res = {"W": 96, "H": 72}                        # the image size of the run in progress (run() sets it)
    import oracle as margin_oracle                  # this repository's CPU restatement: ONLY for the decision margins

    def intrinsics():
        W, H = res["W"], res["H"]
        fx, fy, cx, cy = fov_to_intrinsics(W, H, FOV)   # pixel centre 0.5 (generate_rays)
        return W, H, fx, fy, cx, cy

    def decision_margin(sdf, position_c, orientation_c, inv_scale):
        """smallest |dist - threshold * t| over the samples of the pixel's ray (1e30: no sample), in float64 by the
        oracle, whose hit masks and step counts equal the twin's (tests/test_oracle_golden.py): how far each pixel is
        from changing between hit and miss.  Metadata for the tests' tolerances, not a reference value."""
        W, H, fx, fy, cx, cy = intrinsics()
        margin_oracle.set_margin_mode(True)     # the hit tests: the decisions that change a pixel's depth
        try:
            return margin_oracle.render_forward(np.asarray(sdf, np.float64), np.asarray(position_c, np.float64),
                                                np.asarray(orientation_c, np.float64), [float(inv_scale)], W, H, cx,
                                                cy, fx, fy, THR, dtype=np.float64, with_aux=True)[2][0]
        finally:
            margin_oracle.set_margin_mode(False)

    class TwinRender(torch.autograd.Function):  # = SDFRendererFunction, sdf_renderer.py:136-261
        @staticmethod
        def forward(ctx, sdf, position, orientation, inv_scale):
            ctx.save_for_backward(sdf, position, orientation, inv_scale)
            W, H = res["W"], res["H"]
            with contextlib.redirect_stdout(io.StringIO()):
                image, der = sr.render_depth(sr.SDFObject(sdf.detach().numpy()), W, H, FOV, "d", THR,
                                             position.detach().numpy(), orientation.detach().numpy(),
                                             inv_scale.detach().numpy())
            ctx.der = der
            ctx.image_np = np.asarray(image)
            return torch.from_numpy(image)

        @staticmethod
        def backward(ctx, inp):
            der = ctx.der
            sdf, pos, quat, inv_s = ctx.saved_tensors
            g = inp.numpy()
            g_sdf = torch.zeros_like(sdf)
            zero = np.zeros_like(g)
            s = lambda n: float(np.sum(der.get(n, zero) * g))
            g_p = torch.tensor([s("x"), s("y"), s("z")], dtype=pos.dtype)
            g_q = torch.tensor([s("qx"), s("qy"), s("qz"), s("qw")], dtype=quat.dtype)
            g_is = torch.tensor(s("s_inv"), dtype=inv_s.dtype)
            for k, v in der.get("sdf", {}).items():
                g_sdf[k] = float(np.sum(v * g))
            if res.get("compat"):
                # run D: d depth / d sdf with the weights the reference's GPU extension really adds
                # (sdf_renderer_cuda.cu:373-388) -- that extension cannot run here, so this one tensor comes
                # from the oracle's mode 1 (pinned by reading those lines), evaluated in float64 on the twin's own depth
                # image; every other number of the run is still the imported reference pieces'
                W, H, fx, fy, cx, cy = intrinsics()
                image, _ = ctx.image_np, None
                o = margin_oracle.render_backward(g.astype(np.float64)[None], image.astype(np.float64)[None],
                                                  sdf.detach().numpy().astype(np.float64),
                                                  pos.detach().numpy().astype(np.float64)[None],
                                                  quat.detach().numpy().astype(np.float64)[None],
                                                  [float(inv_s)], cx, cy, fx, fy, dtype=np.float64, sdf_grad_mode=1)
                exact = margin_oracle.render_backward(g.astype(np.float64)[None], image.astype(np.float64)[None],
                                                      sdf.detach().numpy().astype(np.float64),
                                                      pos.detach().numpy().astype(np.float64)[None],
                                                      quat.detach().numpy().astype(np.float64)[None],
                                                      [float(inv_s)], cx, cy, fx, fy, dtype=np.float64, sdf_grad_mode=0)
                # (the oracle's EXACT weights reproduce the twin's d/dsdf: what licenses taking its mode 1 here)
                twin = g_sdf.numpy().astype(np.float64)
                dev_ = np.max(np.abs(exact[0] - twin)) / max(np.max(np.abs(twin)), 1e-30)
                print(f"    (oracle exact mode vs twin d/dsdf: {dev_:.2e} of its maximum; depth dtype {image.dtype}, sdf {sdf.dtype})")
                assert dev_ <= 1e-5, "oracle != twin (exact mode)"
                g_sdf = torch.from_numpy(o[0].astype(np.float64)).to(sdf.dtype)
            return g_sdf, g_p, g_q, g_is

    def depth_to_pointcloud(depth):  # pointset_utils.py:57-77, "opengl", no mask
        _, _, fx, fy, cx, cy = intrinsics()
        cx0, cy0 = cx - 0.5, cy - 0.5                   # Camera.get_pinhole_camera_parameters(0.0)
        idx = torch.nonzero(depth, as_tuple=True)
        z = depth[idx]
        return torch.stack(((idx[1].float() - cx0) * z / fx, -(idx[0].float() - cy0) * z / fy, -z), 1)

    def inlier_ratio(depth_in, depth_est, thr=0.03):  # simple_setup.py:177-188
        rel = torch.abs(depth_in - depth_est) / depth_in
        return (torch.count_nonzero(rel < thr) / torch.count_nonzero(depth_in)).item()

    def first_gradient_f64(depth_images, cam_p, cam_q, position, orientation, scale, latent):
        """Iteration 1 of the same assembled loop with every piece in float64 (vae.double(), float64 parameters and
        cameras; the observation -- the float32 depth images and their float32 back-projection -- is the input and
        stays what it is): d loss / d (position, orientation, scale, latent).  The float32 pass's own gradients
        (``*_grads``) differ from these by its rounding; this is the comparand for a 1e-4 bound."""
        import copy
        f64 = torch.float64
        vae64 = copy.deepcopy(vae).double()
        p, o, s, z = (t.detach().to(f64).clone().requires_grad_() for t in (position, orientation, scale, latent))
        norm_o = o / torch.sqrt(torch.sum(o ** 2))
        sdf = vae64.decode(z)
        loss_depth = torch.tensor(0.0, dtype=f64, requires_grad=True)
        loss_pc = torch.tensor(0.0, dtype=f64, requires_grad=True)
        hits = []
        for depth_image, cp, cq in zip(depth_images, cam_p.to(f64), cam_q.to(f64)):
            q_w2c = qu.quaternion_invert(cq)
            position_c = qu.quaternion_apply(q_w2c, p - cp)
            orientation_c = qu.quaternion_multiply(q_w2c, norm_o)
            est = TwinRender.apply(sdf[0, 0], position_c[0], orientation_c[0], 1 / s[0])
            overlap = (depth_image > 0) & (est > 0)
            hits.append((est > 0).numpy())
            loss_depth = loss_depth + torch.mean(torch.abs(est - depth_image.to(f64))[overlap])
            pts = depth_to_pointcloud(depth_image).to(f64)
            loss_pc = loss_pc + torch.mean(torch.abs(losses.pc_loss(pts, position_c[0], orientation_c[0], s[0], sdf[0, 0])))
        (1.0 * loss_depth + 3.0 * loss_pc).backward()
        return np.concatenate([t.grad.numpy().ravel() for t in (p, o, s, z)]), hits

    def run(tag, z_true, p_true, q_true, s_true, cams, n_iter, shape_opt, point_constraint, out, size=(96, 72),
            min_margin=None, f64_first_gradient=False):
        """min_margin: give up (return False) as soon as a pixel of any view of any iteration lies closer than this
        to one of its decisions -- the search for a scene whose comparison needs no allowance for flipped pixels."""
        res["W"], res["H"] = size
        f32 = torch.float32
        cam_p = torch.tensor([c[0] for c in cams], dtype=f32)
        cam_q = torch.tensor([c[1] for c in cams], dtype=f32)
        cam_q = cam_q / cam_q.norm(dim=1, keepdim=True)
        with torch.no_grad():
            sdf_true = vae.decode(torch.tensor(z_true, dtype=f32)[None])
            targets = []
            for cp, cq in zip(cam_p, cam_q):
                qi = qu.quaternion_invert(cq)
                pc = qu.quaternion_apply(qi, torch.tensor(p_true, dtype=f32)[None] - cp)
                oc = qu.quaternion_multiply(qi, torch.tensor(q_true, dtype=f32)[None])
                targets.append(TwinRender.apply(sdf_true[0, 0].double(), pc[0].double(), oc[0].double(),
                                                torch.tensor(1.0 / s_true, dtype=torch.float64)).float())
        depth_images = torch.stack(targets)
        # initial estimate: the truth, perturbed (C5's recipe)
        position = (torch.tensor(p_true, dtype=f32) + 0.01)[None].clone().requires_grad_()
        ang = np.deg2rad(10.0) / 2
        dq = torch.tensor([np.sin(ang) * 0.6, np.sin(ang) * 0.0, np.sin(ang) * 0.8, np.cos(ang)], dtype=f32)
        orientation = qu.quaternion_multiply(torch.tensor(q_true, dtype=f32), dq)[None].clone().requires_grad_()
        scale = torch.tensor([s_true * 1.08], dtype=f32).requires_grad_()
        latent = torch.zeros((1, cfg["latent_size"]), dtype=f32).requires_grad_()
        out[f"{tag}_depth_images"] = depth_images.numpy()
        out[f"{tag}_cam_pos"] = cam_p.numpy(); out[f"{tag}_cam_quat"] = cam_q.numpy()
        out[f"{tag}_init"] = np.concatenate([position.detach().numpy().ravel(), orientation.detach().numpy().ravel(),
                                             scale.detach().numpy().ravel(), latent.detach().numpy().ravel()])
        opt = torch.optim.Adam([{"params": position, "lr": 1e-3}, {"params": orientation, "lr": 1e-2},
                                {"params": scale, "lr": 1e-3}, {"params": latent, "lr": 1e-2}])
        traj, terms, ratios, grads, margins = [], [], [], [], []
        initial = tuple(t.detach().clone() for t in (position, orientation, scale, latent))
        first_hits = []
        for it in range(n_iter):
            opt.zero_grad()
            norm_orientation = orientation / torch.sqrt(torch.sum(orientation ** 2))
            with torch.set_grad_enabled(shape_opt):
                sdf = vae.decode(latent)
            loss_depth = torch.tensor(0.0, requires_grad=True)
            loss_pc = torch.tensor(0.0, requires_grad=True)
            for depth_image, cp, cq in zip(depth_images, cam_p, cam_q):
                q_w2c = qu.quaternion_invert(cq)
                position_c = qu.quaternion_apply(q_w2c, position - cp)
                orientation_c = qu.quaternion_multiply(q_w2c, norm_orientation)
                margins.append(decision_margin(sdf[0, 0].detach().numpy(), position_c[0].detach().numpy(),
                                               orientation_c[0].detach().numpy(), 1.0 / float(scale[0])))
                if min_margin is not None and margins[-1].min() < min_margin:
                    print(f"  {tag} it{it}: a pixel at margin {margins[-1].min():.2e} < {min_margin:.0e}: scene dropped")
                    return False
                depth_estimate = TwinRender.apply(sdf[0, 0], position_c[0], orientation_c[0], 1 / scale[0]).float()
                if it == 0:
                    first_hits.append((depth_estimate > 0).numpy())
                overlap = (depth_image > 0) & (depth_estimate > 0)
                loss_depth = loss_depth + torch.mean(torch.abs(depth_estimate - depth_image)[overlap])
                pts = depth_to_pointcloud(depth_image)
                loss_pc = loss_pc + torch.mean(torch.abs(
                    losses.pc_loss(pts, position_c[0], orientation_c[0], scale[0], sdf[0, 0])))
            if point_constraint is not None:
                src, tgt, wgt = point_constraint
                loss_con = wgt * losses.point_constraint_loss(orientation[0], torch.tensor(src, dtype=f32),
                                                              torch.tensor(tgt, dtype=f32))
            else:
                loss_con = orientation.new_tensor(0.0)
            loss = 1.0 * loss_depth + 3.0 * loss_pc + loss_con   # default.yaml:14-16
            loss.backward()
            grads.append(np.concatenate([position.grad.numpy().ravel(), orientation.grad.numpy().ravel(),
                                         scale.grad.numpy().ravel(),
                                         latent.grad.numpy().ravel() if latent.grad is not None
                                         else np.zeros(cfg["latent_size"], np.float32)]))
            opt.step()
            with torch.no_grad():
                orientation /= torch.sqrt(torch.sum(orientation ** 2))
                ratios.append(inlier_ratio(depth_image, depth_estimate))   # LAST view's variables
            traj.append(np.concatenate([position.detach().numpy().ravel(), orientation.detach().numpy().ravel(),
                                        scale.detach().numpy().ravel(), latent.detach().numpy().ravel()]))
            terms.append([loss_depth.item(), loss_pc.item(), float(loss_con.detach()), loss.item()])
            print(f"  {tag} it{it}: depth {loss_depth.item():.6f} pc {loss_pc.item():.6f} con {float(loss_con):.6f} "
                  f"inlier {ratios[-1]:.4f} hits {[int((t > 0).sum()) for t in depth_images]}")
        out[f"{tag}_traj"] = np.array(traj); out[f"{tag}_terms"] = np.array(terms)
        out[f"{tag}_inlier"] = np.array(ratios); out[f"{tag}_grads"] = np.array(grads)
        # per (iteration, view): the smallest decision margin and how many pixels lie within 1e-6 / 1e-5 of flipping
        mg = np.array(margins).reshape(n_iter, len(cams), -1)
        out[f"{tag}_margin_min"] = mg.min(axis=2)
        out[f"{tag}_fragile_1e-6"] = (mg < 1e-6).sum(axis=2)
        out[f"{tag}_fragile_1e-5"] = (mg < 1e-5).sum(axis=2)
        out[f"{tag}_size"] = np.array(size)
        if f64_first_gradient:
            g64, hits64 = first_gradient_f64(depth_images, cam_p, cam_q, *initial)
            # (a clean scene: the float64 pass renders the very pixels the float32 pass rendered)
            assert all(np.array_equal(a, b) for a, b in zip(first_hits, hits64)), "hit masks differ between the passes"
            out[f"{tag}_grads_f64"] = g64
            rel = np.abs(g64 - grads[0]) / np.array([np.abs(g64[0:3]).max()] * 3 + [np.abs(g64[3:7]).max()] * 4
                                                     + [abs(g64[7])] + [np.abs(g64[8:]).max()] * (len(g64) - 8))
            print(f"  {tag}: float32 pass vs float64 pass, first gradient, per group scale: max {rel.max():.2e}")
        return True

    if compat_only:
        # Run D: scene C (its seed is in loop_g7.npz) with the reference GPU extension's d/dSDF weights, the gradient
        # the published system's latent trajectories came from -- 160x120, 2 views, shape optimisation on.
        seed = int(np.load(os.path.join(OUT, "loop_g7.npz"))["c_seed"])
        rc = np.random.default_rng(seed)
        qc = rc.normal(size=4); qc /= np.linalg.norm(qc)
        zc = (0.5 * rc.normal(size=cfg["latent_size"])).astype(np.float32)
        pc = (float(rc.uniform(-0.03, 0.03)), float(rc.uniform(-0.03, 0.03)), float(rc.uniform(-0.45, -0.36)))
        cams_c = [((0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0)),
                  ((float(rc.uniform(0.15, 0.3)), float(rc.uniform(-0.06, 0.06)), float(rc.uniform(-0.1, 0.0))),
                   (float(rc.uniform(-0.05, 0.05)), float(rc.uniform(0.2, 0.4)), float(rc.uniform(-0.05, 0.05)), 0.94))]
        trial = {}
        res["compat"] = True
        ok = run("d", zc, pc, qc, float(rc.uniform(0.10, 0.13)), cams_c, 4, True, None, trial, size=(160, 120))
        res["compat"] = False
        assert ok
        W, H, fx, fy, cx, cy = intrinsics()
        trial.update(d_W=W, d_H=H, d_fx=fx, d_fy=fy, d_cx=cx, d_cy=cy, thr=THR, d_seed=seed,
                     note="d/dsdf: oracle mode 1 (sdf_renderer_cuda.cu:373-388 by reading); everything else imported reference pieces")
        np.savez_compressed(os.path.join(OUT, "loop_g7_compat.npz"), **trial)
        return
    rng = np.random.default_rng(17)
    W, H, fx, fy, cx, cy = intrinsics()
    out = {"W": W, "H": H, "fov": FOV, "thr": THR, "fx": fx, "fy": fy, "cx": cx, "cy": cy}
    q_true = rng.normal(size=4); q_true /= np.linalg.norm(q_true)
    z_true = (0.5 * rng.normal(size=cfg["latent_size"])).astype(np.float32)
    cams_a = [((0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0)),
              ((0.25, 0.05, -0.08), (0.02, 0.33, -0.04, 0.94))]
    run("a", z_true, (0.02, -0.01, -0.40), q_true, 0.12, cams_a, 5, True, None, out)
    out["b_constraint_source"] = np.array([0.0, 1.0, 0.0], np.float32)
    out["b_constraint_target"] = np.array([0.1, 0.9, -0.2], np.float32)
    out["b_constraint_weight"] = 0.05
    run("b", z_true, (-0.03, 0.02, -0.35), q_true, 0.10, cams_a[:1], 4, False,
        (out["b_constraint_source"], out["b_constraint_target"], 0.05), out)
    out["z_true"] = z_true; out["q_true"] = q_true
    # Run C, the CLEAN scene: 160x120, 2 views with cameras, shape optimisation on -- the first
    # seeded scene in which no sample of any pixel of any view of any iteration lies within 2e-7 of its hit test
    # (dist against threshold * t ~ 2.5e-3; fp32 against float64 moves the comparison by ~1e-8), so that loss terms,
    # first gradients and the trajectory can be compared without an allowance for pixels that flip.  (A wider band is
    # not to be had: ~4000 hit pixels per image leave ~1 sample per 3 images within 2e-7 already.)  Its own
    # generator: runs A and B keep their draws.
    for seed in range(100, 400):
        rc = np.random.default_rng(seed)
        qc = rc.normal(size=4); qc /= np.linalg.norm(qc)
        zc = (0.5 * rc.normal(size=cfg["latent_size"])).astype(np.float32)
        pc = (float(rc.uniform(-0.03, 0.03)), float(rc.uniform(-0.03, 0.03)), float(rc.uniform(-0.45, -0.36)))
        cams_c = [((0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0)),
                  ((float(rc.uniform(0.15, 0.3)), float(rc.uniform(-0.06, 0.06)), float(rc.uniform(-0.1, 0.0))),
                   (float(rc.uniform(-0.05, 0.05)), float(rc.uniform(0.2, 0.4)), float(rc.uniform(-0.05, 0.05)), 0.94))]
        trial = {}
        print(f" run C: seed {seed}")
        if run("c", zc, pc, qc, float(rc.uniform(0.10, 0.13)), cams_c, 4, True, None, trial, size=(160, 120),
               min_margin=2e-7, f64_first_gradient=True):
            out.update(trial)
            out["c_seed"] = seed
            W, H, fx, fy, cx, cy = intrinsics()
            out.update(c_W=W, c_H=H, c_fx=fx, c_fy=fy, c_cx=cx, c_cy=cy)
            break
    else:
        raise RuntimeError("no clean scene among the seeds tried")
    res["W"], res["H"] = 96, 72
    # nn_loss (losses.py:8-29) and point_constraint_loss (:138-153) on seeded inputs
    a = torch.tensor(rng.normal(size=(200, 3)), dtype=torch.float32, requires_grad=True)
    b = torch.tensor(rng.normal(size=(333, 3)) * 1.3, dtype=torch.float32, requires_grad=True)
    d = losses.nn_loss(a, b)
    gw = torch.tensor(rng.uniform(-1, 1, 200), dtype=torch.float32)
    d.backward(gw)
    out.update(nn_from=a.detach().numpy(), nn_to=b.detach().numpy(), nn_d=d.detach().numpy(), nn_gout=gw.numpy(),
               nn_gfrom=a.grad.numpy(), nn_gto=b.grad.numpy())
    qs = rng.normal(size=(6, 4)); src = rng.normal(size=(6, 3)); tgt = rng.normal(size=(6, 3))
    pc_v, pc_g = [], []
    for i in range(6):
        q = torch.tensor(qs[i], dtype=torch.float64, requires_grad=True)
        v = losses.point_constraint_loss(q, torch.tensor(src[i]), torch.tensor(tgt[i]))
        v.backward()
        pc_v.append(v.item()); pc_g.append(q.grad.numpy().copy())
    out.update(pcl_q=qs, pcl_src=src, pcl_tgt=tgt, pcl_value=np.array(pc_v), pcl_gq=np.array(pc_g))
    np.savez_compressed(os.path.join(OUT, "loop_g7.npz"), **out)


def make_init_network(ref):
    """Forward of the initialisation network on seeded random weights
    (sdfest_amd.synthetic.init_network_state: the trained weights are not in the reference repository).
    Backbone: the IMPORTED sdfest/initialization/pointnet.py::VanillaPointNet in eval mode.  Head:
    sdf_pose_network.py cannot be imported (it imports so3grid -> healpy, absent), so SDFPoseHead.forward
    (:88-115) is restated with the torch modules it is made of (nn.Linear, nn.BatchNorm1d, relu); the
    arithmetic is torch's.  Writes the set features, the head outputs, softmax / adjusted posteriors."""
    import torch
    import torch.nn as nn
    sys.path.insert(0, ref)
    from sdfest.initialization import pointnet
    from sdfest_amd.synthetic import MUG_INIT_BACKBONE, MUG_INIT_HEAD, init_network_state
    out = {}
    rng = np.random.default_rng(23)
    for tag, bb, hd in (("mug", MUG_INIT_BACKBONE, MUG_INIT_HEAD),
                        ("plain", {"in_size": 3, "mlp_out_sizes": [64, 64, 200], "batchnorm": False, "dense": False,
                                   "residual": True},
                         {"in_size": 200, "mlp_out_sizes": [96], "batchnorm": False, "orientation_repr": "quaternion"})):
        state = init_network_state(7 if tag == "mug" else 8, bb, hd)
        net = pointnet.VanillaPointNet(bb["in_size"], bb["mlp_out_sizes"], bb["batchnorm"], residual=bb["residual"],
                                       dense=bb["dense"])
        sd = {k[len("_backbone."):]: torch.tensor(v) for k, v in state.items() if k.startswith("_backbone.")}
        missing = net.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys and all("num_batches_tracked" in k for k in missing.missing_keys)
        net.eval()
        hs = hd["mlp_out_sizes"]
        lins = [nn.Linear(hd["in_size"] if i == 0 else hs[i - 1], c) for i, c in enumerate(hs)]
        bns = [nn.BatchNorm1d(c) for c in hs] if hd["batchnorm"] else []
        final = nn.Linear(hs[-1], state["_head._final_layer.weight"].shape[0])
        with torch.no_grad():
            for i, l in enumerate(lins):
                l.weight.copy_(torch.tensor(state[f"_head._linear_layers.{i}.weight"]))
                l.bias.copy_(torch.tensor(state[f"_head._linear_layers.{i}.bias"]))
            for i, b in enumerate(bns):
                b.weight.copy_(torch.tensor(state[f"_head._bn_layers.{i}.weight"]))
                b.bias.copy_(torch.tensor(state[f"_head._bn_layers.{i}.bias"]))
                b.running_mean.copy_(torch.tensor(state[f"_head._bn_layers.{i}.running_mean"]))
                b.running_var.copy_(torch.tensor(state[f"_head._bn_layers.{i}.running_var"]))
                b.eval()
            final.weight.copy_(torch.tensor(state["_head._final_layer.weight"]))
            final.bias.copy_(torch.tensor(state["_head._final_layer.bias"]))
        for ci, M in enumerate((777, 2500) if tag == "mug" else (300,)):
            pts = (rng.normal(size=(M, 3)) * np.array([0.05, 0.04, 0.03])).astype(np.float32)
            pts -= pts.mean(0, keepdims=True)
            with torch.no_grad():
                feat = net(torch.tensor(pts)[None])                       # (1, C)
                o = feat
                for i, l in enumerate(lins):                              # sdf_pose_network.py:88-93
                    o = l(o)
                    if hd["batchnorm"]:
                        o = bns[i](o)
                    o = nn.functional.relu(o)
                o = final(o)
            out[f"{tag}{ci}_points"] = pts
            out[f"{tag}{ci}_feature"] = feat[0].numpy()
            out[f"{tag}{ci}_head"] = o[0].numpy()
            print(f"  {tag}{ci}: M={M} feature |max| {np.abs(feat.numpy()).max():.3f} head |max| {np.abs(o.numpy()).max():.3f}")
    # posterior: softmax and the adjustment of simple_setup.py:978-1009 (imported? no: open3d) -- torch ops
    logits = torch.tensor(out["mug0_head"][12:])
    post = torch.softmax(logits, -1)
    prior = torch.tensor(rng.uniform(0.1, 1.0, logits.numel()).astype(np.float32)); prior /= prior.sum()
    train = torch.tensor(rng.uniform(0.5, 1.0, logits.numel()).astype(np.float32)); train /= train.sum()
    adj = torch.nn.functional.normalize(post.clone() * prior / train, p=1, dim=-1)
    out.update(post=post.numpy(), prior=prior.numpy(), train_prior=train.numpy(), post_adjusted=adj.numpy())
    np.savez_compressed(os.path.join(OUT, "init_network.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    jobs = {"render": make_render, "pc_loss": make_pc_loss, "decoder": make_decoder,
            "quaternion": make_quaternion, "loop_g7": make_loop_g7,
            "loop_g7_compat": lambda ref: make_loop_g7(ref, compat_only=True), "init_network": make_init_network}
    for name, fn in jobs.items():
        if args.only and args.only != name:
            continue
        print(f"[{name}]")
        fn(args.ref)


if __name__ == "__main__":
    main()
