#!/usr/bin/env python3
"""C5 (BASELINE configs[4]): full render-and-compare loop, decoder(z) -> 64^3 SDF -> render,
50 Adam steps on a synthetic depth image, mug config, 640x480, one GPU.  Prints ms/iteration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from sdfest_amd import Camera, SDFDecoder, render_depth_gpu
    from sdfest_amd.pipeline import RenderAndCompare
    g = os.path.join(ROOT, "tests", "golden")
    d = np.load(os.path.join(g, "decoder_mug.npz"))
    w = np.load(os.path.join(g, "mug_decoder_weights.npz"))
    cfg = {"latent_size": int(d["latent_size"]), "tsdf": False, "decoder": {
        "fc_layers": [{"out": int(o)} for o in d["fc_out"]],
        "conv_layers": [{"in_size": int(a), "in_channels": int(b), "out_channels": int(c),
                         "kernel_size": int(k), "relu": bool(r)}
                        for a, b, c, k, r in zip(d["conv_in_size"], d["conv_cin"], d["conv_cout"],
                                                 d["conv_k"], d["conv_relu"])]}}
    dec = SDFDecoder.from_config(cfg, {k: w[k] for k in w.files})
    views = int(os.environ.get("VIEWS", "1"))
    cam = Camera(640, 480, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
    dev = "cuda"
    z_true = torch.tensor(d["z"][9:10], device=dev) * 0.5
    p_true = torch.tensor([[0.02, -0.01, -0.5]], device=dev)
    q_true = torch.tensor([[0.2, 0.6, -0.15, 0.75]], device=dev)
    q_true = q_true / q_true.norm()
    s_true = torch.tensor([0.055], device=dev)
    with torch.no_grad():
        target = render_depth_gpu(dec.decode(z_true)[0, 0], p_true[0], q_true[0], 1 / s_true[0], None, None,
                                  None, 0.005, cam)
    targets = target[None].repeat(views, 1, 1).contiguous()
    loop = RenderAndCompare(dec, cam, {"threshold": 0.005, "max_iterations": 50, "depth_weight": 1.0,
                                       "pc_weight": 3.0})
    p0 = p_true + 0.01
    q0 = q_true + torch.tensor([[0.06, -0.05, 0.04, 0.0]], device=dev)
    args = (targets, p0, q0 / q0.norm(), torch.tensor([0.06], device=dev), torch.zeros(1, 8, device=dev))
    loop(*args)  # warm-up
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = loop(*args)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 50 * 1e3)
    # the same loop as a captured launch sequence (no autograd, no Python in the loop)
    from sdfest_amd.pipeline import FusedRenderAndCompare
    fused = FusedRenderAndCompare(dec, cam, loop.config, targets)
    fused(*args[1:])   # builds the graph
    torch.cuda.synchronize()
    ftimes = []
    for _ in range(5):
        t0 = time.perf_counter()
        fout = fused(*args[1:])
        torch.cuda.synchronize()
        ftimes.append((time.perf_counter() - t0) / 50 * 1e3)
    # ... with the depth-L1 as its own kernel between forward and backward (the round-1 v1 sequence)
    unf = FusedRenderAndCompare(dec, cam, loop.config, targets, fuse_depth_loss=False)
    unf(*args[1:])
    torch.cuda.synchronize()
    utimes = []
    for _ in range(5):
        t0 = time.perf_counter()
        unf(*args[1:])
        torch.cuda.synchronize()
        utimes.append((time.perf_counter() - t0) / 50 * 1e3)
    # decoder alone
    z = torch.zeros(1, 8, device=dev)
    with torch.no_grad():
        dec.decode(z); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            dec.decode(z)
        e1.record(); torch.cuda.synchronize()
    print(json.dumps({"workload": f"C5 loop, {views} view(s) 640x480, mug decoder, 50 Adam iterations",
                      "ms_per_iteration_autograd": round(float(np.median(times)), 4),
                      "ms_per_iteration_graph": round(float(np.median(ftimes)), 4),
                      "ms_per_iteration_graph_separate_l1_kernel": round(float(np.median(utimes)), 4),
                      "graph_final_position_error_mm": round((fout[0] - p_true).norm().item() * 1e3, 3),
                      "decoder_forward_us": round(e0.elapsed_time(e1) / 100 * 1e3, 2),
                      "final_position_error_mm": round((out[0] - p_true).norm().item() * 1e3, 3)}))


if __name__ == "__main__":
    main()
