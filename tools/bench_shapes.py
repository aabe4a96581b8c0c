#!/usr/bin/env python3
"""Forward/backward timing over image sizes and grid resolutions (batch mode): looks for cliffs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    import oracle
    from sdfest_amd import BatchRenderPlan, Camera
    dev = torch.device("cuda", 0)
    for W, H, R, B in ((640, 480, 64, 256), (320, 240, 64, 256), (1280, 960, 64, 64), (160, 120, 64, 1024),
                       (640, 480, 32, 256), (640, 480, 128, 256), (640, 480, 100, 256), (640, 480, 200, 64)):
        sdf = torch.tensor(oracle.blobs_sdf(0, R=R), device=dev)
        p, q, i = (torch.tensor(a, device=dev) for a in oracle.random_poses(B, seed=1, width=W, height=H, f=W / 2.0))
        cam = Camera(W, H, W / 2.0, W / 2.0, W / 2.0, H / 2.0, pixel_center=0.5)
        plan = BatchRenderPlan(R, B, cam, device=dev)
        g = torch.rand(B, H, W, device=dev) * 2 - 1
        tf = timeit(lambda: plan.forward(sdf, p, q, i, 0.005))
        tb = timeit(lambda: plan.backward(g, sdf, p, q, i))
        hits = int((plan.depth > 0).sum())
        mpix = B * W * H / 1e6
        print(f"{W}x{H} R={R:3d} B={B:4d}: fwd {tf:8.1f} us  bwd {tb:8.1f} us  {hits / 1e6:6.2f} M hits / {mpix:6.1f} Mpx"
              f"  -> {B / (tf + tb) * 1e6:9.0f} renders/s, {mpix / (tf + tb) * 1e6 / 1e3:6.1f} Gpx/s")


if __name__ == "__main__":
    main()
