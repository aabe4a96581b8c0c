"""ms per iteration of the captured V-view loop of one object in its two forms (tail: the per-view reductions inside the
tail's launch; records: one wave per view writes the view's record first), for a few view counts (run on the GPU box;
SDFR_LIB picks the build, e.g. one with another SDFR_PACKED_MIN_VIEWS)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _loop_scene import c5_scene  # noqa: E402
from sdfest_amd.pipeline import FusedRenderAndCompare  # noqa: E402

for views in [int(v) for v in os.environ.get("VS", "6,7,8,10,12,16,24,32").split(",")]:
    s = c5_scene(views=views, max_iterations=50)
    row = {}
    for form in ("tail", "records"):
        loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], form=form)
        loop(*s["init"])
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            loop(*s["init"])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 50 * 1e3)
        row[form] = float(np.median(ts))
    print(f"views {views:3d}: tail form {row['tail']:.4f} ms  records form {row['records']:.4f} ms per iteration", flush=True)
