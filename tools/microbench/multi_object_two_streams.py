"""K objects as S side-by-side groups of K / S on S HIP streams (MultiObjectRenderAndCompare per group): objects per
second against one group of K on one stream (run on the GPU box)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _loop_scene import c5_scene  # noqa: E402
from sdfest_amd.pipeline import MultiObjectRenderAndCompare  # noqa: E402

s = c5_scene(views=1, max_iterations=50)
dev = s["targets"].device
p0, q0, s0, z0 = s["init"]
for K in (8, 16, 32, 64):
    for S in (1, 2, 4):
        G = K // S
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        frames = s["targets"].expand(G, -1, -1).contiguous()
        args = (p0.expand(G, 3).contiguous(), q0.expand(G, 4).contiguous(), s0.expand(G).contiguous(), z0.expand(G, 8).contiguous())
        loops = []
        for g in range(S):
            with torch.cuda.stream(streams[g]):
                m = MultiObjectRenderAndCompare(s["decoder"], s["camera"], s["config"], G)
                m.rebind(frames); m(*args)
            loops.append(m)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for g in range(S):
                with torch.cuda.stream(streams[g]):
                    loops[g].rebind(frames); out = loops[g](*args)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        err = (out[0] - s["p_true"]).norm(dim=1).max().item() * 1e3
        print(f"K={K:3d} as {S} group(s) of {G:2d}: {t * 1e3:7.2f} ms per frame, {K / t:8.1f} objects/s, worst error {err:.3f} mm", flush=True)
        del loops
