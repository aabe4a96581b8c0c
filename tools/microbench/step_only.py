import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sdfest_amd import BatchRenderPlan, Camera
from sdfest_amd.synthetic import blobs_sdf, random_poses
B, W, H = 256, 640, 480
dev = torch.device("cuda:0")
cam = Camera(W, H, W / 2.0, W / 2.0, W / 2.0, H / 2.0, pixel_center=0.5)
pos, quat, isc = (torch.tensor(a, device=dev) for a in random_poses(B, seed=1, width=W, height=H, f=W / 2.0))
sdf = torch.tensor(blobs_sdf(0), device=dev)
g = torch.rand((B, H, W), device=dev) * 2 - 1
plan = BatchRenderPlan(64, B, cam)
for _ in range(4):
    plan.forward(sdf, pos, quat, isc, 0.005, prepare_backward=True)
    plan.backward(g, sdf, pos, quat, isc)
torch.cuda.synchronize()
print("ok")
