import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdfest_amd import SDFDecoder
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(ROOT, "tests", "golden")
d = np.load(os.path.join(g, "decoder_mug.npz")); w = np.load(os.path.join(g, "mug_decoder_weights.npz"))
cfg = {"latent_size": int(d["latent_size"]), "tsdf": False, "decoder": {
    "fc_layers": [{"out": int(o)} for o in d["fc_out"]],
    "conv_layers": [{"in_size": int(a), "in_channels": int(b), "out_channels": int(c), "kernel_size": int(k),
                     "relu": bool(r)} for a, b, c, k, r in zip(d["conv_in_size"], d["conv_cin"], d["conv_cout"],
                                                               d["conv_k"], d["conv_relu"])]}}
print(cfg)
dec = SDFDecoder.from_config(cfg, {k: w[k] for k in w.files})
N = int(os.environ.get("N", "256"))
z = torch.randn(N, 8, device="cuda")
with torch.no_grad():
    for _ in range(5):
        dec.decode(z)
torch.cuda.synchronize()
