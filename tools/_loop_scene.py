"""The C5 scene (BASELINE configs[4]) shared by the loop profiling scripts: mug decoder from the golden
weights, one 640x480 view of a decoded shape, a perturbed initial estimate."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def c5_scene(views: int = 1, max_iterations: int = 50):
    from sdfest_amd import Camera, SDFDecoder, render_depth_gpu
    g = os.path.join(ROOT, "tests", "golden")
    d = np.load(os.path.join(g, "decoder_mug.npz"))
    w = np.load(os.path.join(g, "mug_decoder_weights.npz"))
    cfg = {"latent_size": int(d["latent_size"]), "tsdf": False, "decoder": {
        "fc_layers": [{"out": int(o)} for o in d["fc_out"]],
        "conv_layers": [{"in_size": int(a), "in_channels": int(b), "out_channels": int(c), "kernel_size": int(k),
                         "relu": bool(r)}
                        for a, b, c, k, r in zip(d["conv_in_size"], d["conv_cin"], d["conv_cout"], d["conv_k"],
                                                 d["conv_relu"])]}}
    dec = SDFDecoder.from_config(cfg, {k: w[k] for k in w.files})
    cam = Camera(640, 480, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
    dev = "cuda"
    z_true = torch.tensor(d["z"][9:10], device=dev) * 0.5
    p_true = torch.tensor([[0.02, -0.01, -0.5]], device=dev)
    q_true = torch.tensor([[0.2, 0.6, -0.15, 0.75]], device=dev)
    q_true = q_true / q_true.norm()
    s_true = torch.tensor([0.055], device=dev)
    with torch.no_grad():
        target = render_depth_gpu(dec.decode(z_true)[0, 0], p_true[0], q_true[0], 1 / s_true[0], None, None, None,
                                  0.005, cam)
    targets = target[None].repeat(views, 1, 1).contiguous()
    config = {"threshold": 0.005, "max_iterations": max_iterations, "depth_weight": 1.0, "pc_weight": 3.0}
    q0 = q_true + torch.tensor([[0.06, -0.05, 0.04, 0.0]], device=dev)
    init = (p_true + 0.01, q0 / q0.norm(), torch.tensor([0.06], device=dev), torch.zeros(1, 8, device=dev))
    return {"decoder": dec, "camera": cam, "config": config, "targets": targets, "init": init, "p_true": p_true}
