"""K independent render-and-compare loops (K detected objects of one frame) replaying their captured graphs on K HIP
streams at once: objects per second against one loop at a time (run on the GPU box).  The single loop is a chain of 17
dependent launches that leaves most of the chip idle; independent chains on separate streams can run side by side."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _loop_scene import c5_scene  # noqa: E402
from sdfest_amd.pipeline import FusedRenderAndCompare  # noqa: E402

s = c5_scene(views=1, max_iterations=50)
dev = s["targets"].device
for K in (1, 2, 4, 8, 16):
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    loops = []
    for k in range(K):
        with torch.cuda.stream(streams[k]):
            loop = FusedRenderAndCompare(s["decoder"] if k == 0 else s["decoder"], s["camera"], s["config"], s["targets"])
            loop(*s["init"])
        loops.append(loop)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                loops[k].rebind(s["targets"])
                loops[k](*s["init"])
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print(f"K={K:2d} objects on {K} streams: {t * 1e3:7.2f} ms per round = {t / K * 1e3:6.3f} ms per object, "
          f"{K / t:7.1f} objects/s", flush=True)
    del loops
