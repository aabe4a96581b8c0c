"""Is the decoder's VJP reproducible run to run?  (A shape of the fused-pairs test was not: bisect it.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_decoder_gpu as T
from sdfest_amd import SDFDecoder

def run(case, N, fused=0, tiled=1):
    rng = np.random.default_rng(5)
    cfg = {"latent_size": case["latent"], "tsdf": False, "sdf_size": case["volume"],
           "decoder": {"fc_layers": case["fc"], "conv_layers": case["conv"]}}
    dec = SDFDecoder.from_config(cfg, T._random_state(rng, case), sdf_size=case["volume"])
    dec.set_option("fused_single", fused); dec.set_option("tiled_vjp", tiled)
    z_np = rng.normal(size=(N, case["latent"])).astype(np.float32)
    G = torch.tensor(rng.normal(size=(N, 1, case["volume"],) * 1 + (case["volume"],) * 2).astype(np.float32), device="cuda") if False else \
        torch.tensor(rng.normal(size=(N, 1, case["volume"], case["volume"], case["volume"])).astype(np.float32), device="cuda")
    gs = []
    for rep in range(4):
        junk = torch.full((1 << 22,), float(rep + 1) * 1e3, device="cuda"); del junk     # dirty the allocator's blocks
        z = torch.tensor(z_np, device="cuda", requires_grad=True)
        dec.decode(z).backward(G)
        torch.cuda.synchronize()
        gs.append(z.grad.clone())
    return all(torch.equal(gs[0], g) for g in gs[1:]), [round((g - gs[0]).abs().max().item(), 4) for g in gs[1:]]

base = dict(volume=16, latent=3, fc=[{"out": 7}, {"out": 3 * 6 ** 3}],
            conv=[dict(in_size=6, in_channels=3, out_channels=20, kernel_size=3, relu=True),
                  dict(in_size=12, in_channels=20, out_channels=2, kernel_size=3, relu=False),
                  dict(in_size=16, in_channels=2, out_channels=1, kernel_size=1, relu=True)])
import copy
def var(**kw):
    c = copy.deepcopy(base)
    for k, v in kw.items():
        l, key = k.split("_", 1)
        c["conv"][int(l[1:])][key] = v
    return c
tests = {"base N=2": (base, 2), "base N=1": (base, 1), "base untiled": (base, 2, 0, 0),
         "last relu False": (var(l2_relu=False), 2), "conv1 relu True": (var(l1_relu=True), 2),
         "16 channels": (var(l0_out_channels=16, l1_in_channels=16), 2),
         "17 channels": (var(l0_out_channels=17, l1_in_channels=17), 2),
         "conv1 4 out": (var(l1_out_channels=4, l2_in_channels=4), 2)}
for name, a in tests.items():
    print(f"{name:20s}", run(*a), flush=True)
