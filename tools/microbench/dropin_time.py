import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from sdfest_amd import Camera, render_depth_gpu, synthetic
dev = torch.device("cuda:0")
sdf = torch.tensor(synthetic.blobs_sdf(0), device=dev, requires_grad=True)
p = torch.tensor([0.0, 0.0, -1.5], device=dev, requires_grad=True)
q = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev, requires_grad=True)
s = torch.tensor(2.0, device=dev, requires_grad=True)
cam = Camera(640, 480, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
g = torch.rand(480, 640, device=dev)
def step():
    for t in (sdf, p, q, s): t.grad = None
    d = render_depth_gpu(sdf, p, q, s, None, None, None, 0.005, cam)
    d.backward(g)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 300
for _ in range(n): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t = time.perf_counter() - t0
def fwd_only():
    with torch.no_grad():
        return render_depth_gpu(sdf, p, q, s, None, None, None, 0.005, cam)
for _ in range(20): fwd_only()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(n): fwd_only()
torch.cuda.synchronize(); tf = time.perf_counter() - t1
print(f"render_depth_gpu fwd+bwd through autograd: {t / n * 1e6:.1f} us per pair (host issue {t_issue / n * 1e6:.1f}); forward only, no_grad: {tf / n * 1e6:.1f} us")
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(300): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
