"""Objects per second: K estimates side by side (MultiObjectRenderAndCompare) against one at a time, on the C5 image
(run on the GPU box)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _loop_scene import c5_scene  # noqa: E402
from sdfest_amd.pipeline import FusedRenderAndCompare, MultiObjectRenderAndCompare  # noqa: E402

s = c5_scene(views=1, max_iterations=50)
single = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"])
single(*s["init"])
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); single.rebind(s["targets"]); single(*s["init"]); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
t1 = float(np.median(ts))
print(f"one object at a time: {t1 * 1e3:.3f} ms per object, {1 / t1:.1f} objects/s", flush=True)
p0, q0, s0, z0 = s["init"]
for K in [int(k) for k in os.environ.get("KS", "1,2,4,8,16,32,64").split(",")]:
    multi = MultiObjectRenderAndCompare(s["decoder"], s["camera"], s["config"], K)
    frames = s["targets"].expand(K, -1, -1).contiguous()
    args = (p0.expand(K, 3).contiguous(), q0.expand(K, 4).contiguous(), s0.expand(K).contiguous(), z0.expand(K, 8).contiguous())
    multi.rebind(frames); multi(*args)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); multi.rebind(frames); out = multi(*args); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    err = (out[0] - s["p_true"]).norm(dim=1).max().item() * 1e3
    print(f"K={K:3d} side by side: {t * 1e3:8.3f} ms per frame = {t / K * 1e3:6.3f} ms per object, {K / t:8.1f} objects/s "
          f"({K / t * t1:5.2f}x), {t / 50 * 1e3:.4f} ms per iteration, worst final position error {err:.3f} mm", flush=True)
    del multi
