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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    jobs = {"render": make_render, "pc_loss": make_pc_loss, "decoder": make_decoder,
            "quaternion": make_quaternion}
    for name, fn in jobs.items():
        if args.only and args.only != name:
            continue
        print(f"[{name}]")
        fn(args.ref)


if __name__ == "__main__":
    main()
