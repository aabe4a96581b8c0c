"""ms per iteration and per new observation of the captured C5 loop for several graph lengths (iterations per replayed
hipGraph), and what the longer captures cost the first call (run on the GPU box)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _loop_scene import c5_scene  # noqa: E402
from sdfest_amd.pipeline import FusedRenderAndCompare  # noqa: E402

s = c5_scene(views=1, max_iterations=50)
for gi in (5, 10, 25, 50, 5, 10, 25, 50):
    t0 = time.perf_counter()
    loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], graph_iterations=gi)
    loop(*s["init"])
    torch.cuda.synchronize()
    first = (time.perf_counter() - t0) * 1e3
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        loop.rebind(s["targets"])
        loop(*s["init"])
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"graph_iterations {gi:2d}: first call {first:6.2f} ms, new observation {np.median(ts):.3f} ms "
          f"({np.median(ts) / 50:.4f} ms per iteration)", flush=True)
