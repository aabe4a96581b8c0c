"""Scenes of the sharded-loop tests (tests/test_loop_sharded_gpu.py and its worker): built the same way, from the
same numbers, in every process.
  "g7a"    run A of G7 (tests/golden/loop_g7.npz: 2 views with camera extrinsics, assembled from imported reference
           pieces by tools/make_goldens.py::make_loop_g7)
  "seven"  7 cameras on an arc around one mug-decoder shape at 160x120, the observed depth rendered at the true
           pose by the forward kernel (bitwise reproducible), a perturbed initial estimate
  "many"   80 cameras at 96x72 (3 iterations): a view list longer than one round of the tail's chain"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def mug_decoder():
    from sdfest_amd import SDFDecoder
    d = np.load(os.path.join(GOLDEN, "decoder_mug.npz"))
    w = np.load(os.path.join(GOLDEN, "mug_decoder_weights.npz"))
    cfg = {"latent_size": int(d["latent_size"]), "tsdf": False, "decoder": {
        "fc_layers": [{"out": int(o)} for o in d["fc_out"]],
        "conv_layers": [{"in_size": int(a), "in_channels": int(b), "out_channels": int(c), "kernel_size": int(k),
                         "relu": bool(r)}
                        for a, b, c, k, r in zip(d["conv_in_size"], d["conv_cin"], d["conv_cout"], d["conv_k"],
                                                 d["conv_relu"])]}}
    return SDFDecoder.from_config(cfg, {k: w[k] for k in w.files}), d


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _qrot(q, v):
    return _qmul(_qmul(q, np.append(v, 0.0)), q * np.array([-1, -1, -1, 1.0]))[:3]


def build(name, iterations=None, dev="cuda"):
    """-> dict(decoder, camera, config, depth (V,H,W), cam_pos (V,3), cam_quat (V,4), init (p0, q0, s0, z0))"""
    from sdfest_amd import Camera, render_depth_gpu
    dec, d = mug_decoder()
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    if name == "g7a":
        g7 = np.load(os.path.join(GOLDEN, "loop_g7.npz"))
        W, H = int(g7["W"]), int(g7["H"])
        cam = Camera(W, H, float(g7["fx"]), float(g7["fy"]), float(g7["cx"]), float(g7["cy"]), pixel_center=0.5)
        init = g7["a_init"]
        n_iter = iterations or g7["a_traj"].shape[0]
        cfg = {"threshold": float(g7["thr"]), "max_iterations": n_iter, "depth_weight": 1.0, "pc_weight": 3.0,
               "result_selection_strategy": "best_inlier_ratio"}
        return dict(decoder=dec, camera=cam, config=cfg, depth=t(g7["a_depth_images"]), cam_pos=t(g7["a_cam_pos"]),
                    cam_quat=t(g7["a_cam_quat"]),
                    init=(t(init[None, 0:3]), t(init[None, 3:7]), t(init[7:8]), t(init[None, 8:])))
    if name not in ("seven", "many"):
        raise ValueError(name)
    # "many": 80 small views -- more than the single-process tail's 64, several rounds of the tail's 256-thread chain
    V, W, H, f = (7, 160, 120, 160.0) if name == "seven" else (80, 96, 72, 96.0)
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    z_true = d["z"][9:10] * 0.5
    p_true = np.array([0.02, -0.01, -0.5])
    q_true = np.array([0.2, 0.6, -0.15, 0.75]); q_true /= np.linalg.norm(q_true)
    s_true = 0.055
    cam_pos, cam_quat = [], []
    for i in range(V):
        a = (i - 3) * np.deg2rad(2.5) if name == "seven" else (i % 9 - 4) * np.deg2rad(2.0)   # yaw about the world's y axis
        cam_pos.append([0.5 * np.sin(a) * 0.3, 0.01 * (i % 3 - 1), 0.02 * (i % 2)])
        cam_quat.append([0.01 * (i % 2), np.sin(0.15 * a), 0.0, np.cos(0.15 * a)])
    cam_pos = np.array(cam_pos)
    cam_quat = np.array(cam_quat); cam_quat /= np.linalg.norm(cam_quat, axis=1, keepdims=True)
    with torch.no_grad():
        sdf = dec.decode(t(z_true))[0, 0]
        depth = []
        for i in range(V):
            qc = cam_quat[i] * np.array([-1, -1, -1, 1.0])
            depth.append(render_depth_gpu(sdf, t(_qrot(qc, p_true - cam_pos[i])), t(_qmul(qc, q_true)),
                                          t(1.0 / s_true), None, None, None, 0.005, cam))
        depth = torch.stack(depth).contiguous()
    q0 = q_true + np.array([0.05, -0.04, 0.03, 0.0])
    cfg = {"threshold": 0.005, "max_iterations": iterations or (6 if name == "seven" else 3), "depth_weight": 1.0, "pc_weight": 3.0,
           "result_selection_strategy": "best_inlier_ratio"}
    return dict(decoder=dec, camera=cam, config=cfg, depth=depth, cam_pos=t(cam_pos), cam_quat=t(cam_quat),
                init=(t(p_true[None] + 0.008), t((q0 / np.linalg.norm(q0))[None]), t([0.06]),
                      torch.zeros(1, int(d["latent_size"]), device=dev)))


def history_array(hist):
    """(iterations, 8 + L) parameter trajectory of a history list"""
    return np.stack([np.concatenate([h[k].detach().cpu().numpy().ravel()
                                     for k in ("position", "orientation", "scale", "latent")]) for h in hist])
