import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, oracle
from sdfest_amd import BatchRenderPlan, Camera
dev = torch.device("cuda", 0)
sdf = torch.tensor(oracle.blobs_sdf(0), device=dev)
cam = Camera(640, 480, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in [int(b) for b in os.environ.get("BS", "1,2,3,4,8,16,32,48,54,56,64,96,128,256,512").split(",")]:
    p, q, i = (torch.tensor(a, device=dev) for a in oracle.random_poses(B, seed=1))
    plan = BatchRenderPlan(64, B, cam, device=dev)
    g = torch.rand(B, 480, 640, device=dev) * 2 - 1
    tf = timeit(lambda: plan.forward(sdf, p, q, i, 0.005))
    tb = timeit(lambda: plan.backward(g, sdf, p, q, i))
    print(f"B={B:4d} fwd {tf:8.1f} us bwd {tb:8.1f} us  per view {((tf+tb)/B):7.2f} us  -> {B/(tf+tb)*1e6:10.0f} renders/s")
