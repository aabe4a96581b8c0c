"""What a node of a replayed hipGraph costs against the same launches issued eagerly, under the runtime's graph
switches (run on the GPU box; the switches are read when libamdhip64 initialises, so one process per setting:
tools/microbench/graph_env.sh).  Prints one JSON line: the C5 loop per iteration (graph, 5 iterations per replay),
C1 / C2 forward + backward pairs eager and replayed, and a chain of empty-ish launches eager and replayed."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from _loop_scene import c5_scene  # noqa: E402


def event_us(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def capture(step):
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(gr):
        step()
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    return gr


def pair(W, H):
    from sdfest_amd import BatchRenderPlan, Camera
    from sdfest_amd.synthetic import blobs_sdf
    dev = "cuda"
    f = W / 2.0
    cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
    sdf = torch.tensor(blobs_sdf(0), device=dev)
    pos = torch.tensor([[0.0, 0.0, -1.5]], device=dev)
    quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)
    isc = torch.tensor([2.0], device=dev)
    g = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (1, H, W)).astype(np.float32), device=dev)
    plan = BatchRenderPlan(64, 1, cam, device=dev)

    def step():
        plan.forward(sdf, pos, quat, isc, 0.005, prepare_backward=True)
        plan.backward(g, sdf, pos, quat, isc)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    out = {}
    for rep in range(3):
        out.setdefault("eager", []).append(round(event_us(step, 300), 2))
    gr = capture(step)
    for rep in range(3):
        out.setdefault("graph", []).append(round(event_us(gr.replay, 300), 2))
    # ten pairs per replay: the graph launch itself amortised
    def ten():
        for _ in range(10):
            step()
    gr10 = capture(ten)
    for rep in range(3):
        out.setdefault("graph_x10_per_pair", []).append(round(event_us(gr10.replay, 60) / 10, 2))
    return out


def tiny_chain(n=16):
    x = torch.zeros(64, device="cuda")

    def step():
        for _ in range(n):
            x.add_(1.0)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    eager = round(event_us(step, 100) / n, 2)
    gr = capture(step)
    graph = round(event_us(gr.replay, 200) / n, 2)
    return {"launches": n, "eager_us_per_launch": eager, "graph_us_per_launch": graph}


def main():
    from sdfest_amd.pipeline import FusedRenderAndCompare
    res = {"env": {k: os.environ[k] for k in os.environ if k.startswith(("DEBUG_", "HIP_", "GPU_", "AMD_", "HSA_"))}}
    res["tiny_chain"] = tiny_chain()
    res["C1"] = pair(160, 120)
    res["C2"] = pair(640, 480)
    s = c5_scene(views=1, max_iterations=50)
    for gi in (5, 50):
        loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], graph_iterations=gi)
        loop(*s["init"])
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            loop.rebind(s["targets"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(*s["init"])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3 / 50)
        res[f"C5_ms_per_iteration_gi{gi}"] = round(float(np.median(ts)), 4)
    # eager loop (host-bound): the same launches without a graph
    loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], graph_iterations=5)
    loop(*s["init"], use_graph=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        loop.rebind(s["targets"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(*s["init"], use_graph=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3 / 50)
    res["C5_ms_per_iteration_eager"] = round(float(np.median(ts)), 4)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
