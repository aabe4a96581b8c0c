"""A C3 step (256 views of 640x480, forward + backward) as ONE plan on one stream against TWO plans of 128 views on two
streams, the second half's forward started when the first half's has finished -- so that a forward (bound by the CU's
vector-memory pipeline) runs beside a backward (VALU + LDS adds).  us per step, HIP events around STEPS steps."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sdfest_amd import BatchRenderPlan, Camera
from sdfest_amd.synthetic import blobs_sdf, random_poses
B, W, H = 256, 640, 480
STEPS = int(os.environ.get("STEPS", "50"))
dev = torch.device("cuda:0")
cam = Camera(W, H, W / 2.0, W / 2.0, W / 2.0, H / 2.0, pixel_center=0.5)
pos, quat, isc = (torch.tensor(a, device=dev) for a in random_poses(B, seed=1, width=W, height=H, f=W / 2.0))
sdf = torch.tensor(blobs_sdf(0), device=dev)
g = torch.rand((B, H, W), device=dev) * 2 - 1


def timed(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


one = BatchRenderPlan(64, B, cam)
def step_one():
    one.forward(sdf, pos, quat, isc, 0.005, prepare_backward=True)
    one.backward(g, sdf, pos, quat, isc)
print("one plan, one stream: %.1f us per step" % timed(step_one, STEPS))

h = B // 2
halves = [BatchRenderPlan(64, h, cam) for _ in range(2)]
parts = [(pos[k * h:(k + 1) * h].contiguous(), quat[k * h:(k + 1) * h].contiguous(), isc[k * h:(k + 1) * h].contiguous(),
          g[k * h:(k + 1) * h].contiguous()) for k in range(2)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
def step_two(stagger=True):
    main = torch.cuda.current_stream()
    start = torch.cuda.Event(); start.record(main)
    fwd_done = torch.cuda.Event()
    for k in range(2):
        with torch.cuda.stream(s[k]):
            s[k].wait_event(start)
            if k == 1 and stagger:
                s[k].wait_event(fwd_done)
            p, q, i, gg = parts[k]
            halves[k].forward(sdf, p, q, i, 0.005, prepare_backward=True)
            if k == 0:
                fwd_done.record(s[0])
            halves[k].backward(gg, sdf, p, q, i)
    for k in range(2):
        main.wait_stream(s[k])
print("two plans of 128 views, two streams, staggered: %.1f us per step" % timed(lambda: step_two(True), STEPS))
print("two plans of 128 views, two streams, both at once: %.1f us per step" % timed(lambda: step_two(False), STEPS))
def step_seq():
    for k in range(2):
        p, q, i, gg = parts[k]
        halves[k].forward(sdf, p, q, i, 0.005, prepare_backward=True)
        halves[k].backward(gg, sdf, p, q, i)
print("two plans of 128 views, one stream: %.1f us per step" % timed(step_seq, STEPS))
