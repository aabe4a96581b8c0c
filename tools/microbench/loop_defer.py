"""ms per iteration of the captured C5 loop with the depth-loss reduction deferred into the backward's launch or as
its own launch, for a few view counts (run on the GPU box)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _loop_scene import c5_scene  # noqa: E402
from sdfest_amd.pipeline import FusedRenderAndCompare  # noqa: E402

for views in [int(v) for v in os.environ.get("VS", "1,2,4,8,16,64").split(",")]:
    s = c5_scene(views=views, max_iterations=50)
    row = {}
    for defer in (False, True):
        loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], defer_loss=defer)
        loop(*s["init"])
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            loop(*s["init"])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 50 * 1e3)
        row[defer] = float(np.median(ts))
    print(f"views {views:3d}: own launch {row[False]:.4f} ms  deferred {row[True]:.4f} ms per iteration", flush=True)
