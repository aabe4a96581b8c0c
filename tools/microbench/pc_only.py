"""The sampler's backward on coherent points (the back-projected depth images of 64 rendered views), a few launches:
the program for a trace / counter pass (tools/trace_cmd.sh, tools/pmc_cmd.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sdfest_amd import BatchRenderPlan, Camera
from sdfest_amd.generated_views import depth_to_pointsets
from sdfest_amd.losses import _backward_raw, _forward_raw
from sdfest_amd.synthetic import blobs_sdf, random_poses
dev = torch.device("cuda", 0)
sdf = torch.tensor(blobs_sdf(0), device=dev)
cam = Camera(640, 480, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
V = 64
pr, qr, ir = (torch.tensor(a, device=dev) for a in random_poses(V, seed=1))
depth = BatchRenderPlan(64, V, cam, device=dev).forward(sdf, pr, qr, ir, 0.005)
pts, counts = depth_to_pointsets(depth, cam, tiled=os.environ.get("TILED", "0") == "1")
offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), counts.cumsum(0)]).to(torch.int32)
M = int(counts.max())
go = torch.rand(pts.shape[0], device=dev) * 2 - 1
for _ in range(4):
    _forward_raw(pts, offs, M, pr, qr, 1.0 / ir, sdf)
    _backward_raw(go, pts, offs, M, pr, qr, 1.0 / ir, sdf)
torch.cuda.synchronize()
print("points", pts.shape[0])
