#!/usr/bin/env python3
"""Batched decoder forward + VJP to the latents: the program to put after `rocprofv3 --kernel-trace --`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from sdfest_amd import SDFDecoder
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
    N = int(os.environ.get("N", "256"))
    z = torch.randn(N, 8, device="cuda", requires_grad=True)
    G = torch.randn(N, 1, 64, 64, 64, device="cuda")
    for _ in range(3):
        z.grad = None
        dec.decode(z).backward(G)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
