"""Kernel-level comparison of the plain step and the loss-fused step at C3 (run under rocprofv3 --kernel-trace):
    tools/trace_full.sh <tag> tools/microbench/step_l1_kernels.py
60 steps of each form after a warm-up; the trace's per-kernel averages say where the fused form's extra time goes."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (pose generator only)
from sdfest_amd import BatchRenderPlan, Camera  # noqa: E402

B, W, H = 256, 640, 480
dev = torch.device("cuda:0")
cam = Camera(W, H, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
pos, quat, isc = (torch.tensor(a, device=dev) for a in oracle.random_poses(B, seed=1))
sdf = torch.tensor(oracle.blobs_sdf(0), device=dev)
plan = BatchRenderPlan(64, B, cam, device=dev)
g = torch.Generator(device=dev).manual_seed(5)
target = plan.forward(sdf, pos + 0.01 * torch.randn(pos.shape, device=dev, generator=g), quat, isc, 0.005).clone()
grad = torch.empty_like(target).uniform_(-1, 1)
which = sys.argv[1] if len(sys.argv) > 1 else "both"


def plain():
    plan.forward(sdf, pos, quat, isc, 0.005, prepare_backward=True)
    plan.backward(grad, sdf, pos, quat, isc)


def fused():
    plan.forward_l1(sdf, pos, quat, isc, 0.005, target, prepare_backward=True,
                    defer_loss=os.environ.get("SDFR_DEFER", "1") == "1")
    plan.backward_l1(target, sdf, pos, quat, isc)


for name, fn in (("plain", plain), ("fused", fused)):
    if which not in ("both", name):
        continue
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    print(name, "us per step", round((time.perf_counter() - t0) / 60 * 1e6, 1), flush=True)
