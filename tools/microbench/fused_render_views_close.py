"""The one-launch render step against the two launches for 2 / 4 / 7 views of a small (4 k pixels) and a larger (25 k)
object, shape optimised: where the default of FusedRenderAndCompare(fused_render=None) -- up to 4 views -- comes from
(run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from _loop_scene import c5_scene
from sdfest_amd import render_depth_gpu
from sdfest_amd.pipeline import FusedRenderAndCompare
s = c5_scene(views=1, max_iterations=50)
dec, cam = s["decoder"], s["camera"]
p0, q0, s0, z0 = s["init"]
for zc in (-0.5, -0.2):
    p_true = torch.tensor([[0.0, 0.0, zc]], device="cuda")
    with torch.no_grad():
        sdf = dec.decode(torch.zeros(1, 8, device="cuda"))[0, 0]
        tgt = render_depth_gpu(sdf, p_true[0], q0[0], 1 / s0[0], None, None, None, 0.005, cam)[None].contiguous()
    for V in (2, 4, 7):
        t = tgt.repeat(V, 1, 1).contiguous()
        row = {}
        for fused in (False, True):
            loop = FusedRenderAndCompare(dec, cam, s["config"], t, fused_render=fused, form="tail")
            init = (p_true + 0.004, q0, s0 * 1.03, z0)
            for _ in range(2): loop(*init)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); loop(*init); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 50)
            row[fused] = sorted(ts)[2]
        print(f"z {zc} pixels {int((tgt>0).sum())} views {V}: two launches {row[False]:.4f} one launch {row[True]:.4f}", flush=True)
