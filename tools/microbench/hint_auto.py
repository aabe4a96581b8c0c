"""Step time (forward + backward of the same views) of a BatchRenderPlan with the SDFR_BWD_HALF_GRID hint off, forced
on, and chosen from the device-side count of close views (close_views="auto"); MODE as time_variants.py."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sdfest_amd import BatchRenderPlan, Camera, synthetic as oracle

def main():
    B, W, H = int(os.environ.get("B", 256)), int(os.environ.get("W", 640)), int(os.environ.get("H", 480))
    f = W / 2.0
    dev = torch.device("cuda:0")
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    sdf = torch.tensor(oracle.blobs_sdf(0), device=dev)
    pos, quat, isc = (torch.tensor(a, device=dev) for a in oracle.random_poses(B, seed=1, width=W, height=H, f=f))
    if os.environ.get("MODE") == "mug":
        pos = (pos * torch.tensor([0.3, 0.3, 0.3], device=dev)).contiguous()
        isc = torch.full((B,), 1 / 0.055, device=dev)
    g = torch.rand((B, H, W), device=dev) * 2 - 1
    plans = {k: BatchRenderPlan(64, B, cam, close_views=v) for k, v in (("off", False), ("on", True), ("auto", "auto"))}
    res = {k: [] for k in plans}
    for r in range(8):
        for k, plan in plans.items():
            def step():
                plan.forward(sdf, pos, quat, isc, 0.005, prepare_backward=True)
                plan.backward(g, sdf, pos, quat, isc)
            step(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): step()
            e1.record(); torch.cuda.synchronize()
            if r: res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
    out = "  ".join(f"{k} {np.median(v):.1f} us" for k, v in res.items())
    a = plans["auto"]
    print(f"B={B} {W}x{H} MODE={os.environ.get('MODE', '')}: step {out}; auto: half-grid steps {a.half_grid_steps}, "
          f"close views seen {a.close_views_seen()}", flush=True)
main()
