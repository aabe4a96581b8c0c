#!/usr/bin/env python3
"""C1 (160x120) and C2 (640x480) single-view fwd+bwd, HIP vs the CPU port,
with the gradient errors against the oracle."""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import oracle
    from sdfest_amd import BatchRenderPlan, Camera
    here = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", here, "libsdfr_oracle_native.so"], stdout=subprocess.DEVNULL)
    nat = ctypes.CDLL(os.path.join(here, "libsdfr_oracle_native.so"))
    dev = torch.device("cuda", 0)
    sdf_np = oracle.blobs_sdf(0)
    sdf = torch.tensor(sdf_np, device=dev)
    out = {}
    for name, W, H in (("C1 160x120", 160, 120), ("C2 640x480", 640, 480)):
        f = W / 2.0
        cam = Camera(W, H, f, f, W / 2.0, H / 2.0, pixel_center=0.5)
        pos = torch.tensor([[0.0, 0.0, -1.5]], device=dev)
        quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)
        isc = torch.tensor([2.0], device=dev)
        g_np = np.random.default_rng(0).uniform(-1, 1, (1, H, W)).astype(np.float32)
        g = torch.tensor(g_np, device=dev)
        plan = BatchRenderPlan(64, 1, cam, device=dev)

        def pair():   # two stand-alone calls
            plan.forward(sdf, pos, quat, isc, 0.005)
            plan.backward(g, sdf, pos, quat, isc)

        def step():   # the step API (sdfr_render_step_forward / _backward): one launch less per call
            plan.forward(sdf, pos, quat, isc, 0.005, prepare_backward=True)
            plan.backward(g, sdf, pos, quat, isc)
        for _ in range(20):
            pair()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            pair()
        e1.record()
        torch.cuda.synchronize()
        pair_us = e0.elapsed_time(e1) / 300 * 1e3
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 300
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        hip_us = e0.elapsed_time(e1) / n * 1e3
        # graph replay of the same pair of calls
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
        e0.record()
        for _ in range(n):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        graph_us = e0.elapsed_time(e1) / n * 1e3
        # parity vs oracle
        d = plan.depth.cpu().numpy()
        do = oracle.render_forward(sdf_np, [0, 0, -1.5], [0, 0, 0, 1], [2.0], W, H, W / 2, H / 2, f, f, 0.005,
                                   dtype=np.float32)
        ob = oracle.render_backward(g_np, do, sdf_np, [0, 0, -1.5], [0, 0, 0, 1], [2.0], W / 2, H / 2, f, f,
                                    dtype=np.float32)
        gs = plan.g_sdf.cpu().numpy()
        both = (d[0] > 0) & (do[0] > 0)
        res = {"hip_us_fwd_bwd": round(hip_us, 2), "hip_graph_us_fwd_bwd": round(graph_us, 2),
               "hip_us_fwd_bwd_standalone_calls": round(pair_us, 2),
               "hip_renders_per_s": round(1e6 / graph_us, 1), "hit_pixels": int((d > 0).sum()),
               "depth_max_rel_err": float(np.max(np.abs(d[0][both] / do[0][both] - 1))),
               "g_sdf_max_abs_err": float(np.max(np.abs(gs - ob[0]))),
               "g_sdf_max_rel_err_of_max": float(np.max(np.abs(gs - ob[0])) / np.abs(ob[0]).max())}
        # CPU port: 1 thread and a small sweep
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        cd, ci = ctypes.c_double, ctypes.c_int
        p_, q_, i_ = (np.array(a, np.float32) for a in ([[0, 0, -1.5]], [[0, 0, 0, 1]], [2.0]))
        dep = np.empty((1, H, W), np.float32)
        gsb = np.empty((64, 64, 64), np.float32); gp = np.empty((1, 3), np.float32)
        gq = np.empty((1, 4), np.float32); gi = np.empty(1, np.float32)

        def cpu_once():
            nat.sdfo_render_forward_f32(P(sdf_np), ci(64), P(p_), P(q_), P(i_), ci(1), ci(W), ci(H), cd(W / 2),
                                        cd(H / 2), cd(f), cd(f), cd(0.005), P(dep), None, None, ci(0))
            nat.sdfo_render_backward_f32(P(g_np), P(dep), P(sdf_np), ci(64), P(p_), P(q_), P(i_), ci(1), ci(W),
                                         ci(H), cd(W / 2), cd(H / 2), cd(f), cd(f), ci(0), P(gsb), P(gp), P(gq), P(gi))
        cpu = {}
        ncpu = len(os.sched_getaffinity(0))
        for th in sorted({1, 8, 32, min(64, ncpu)}):
            nat.sdfo_set_threads(ci(th))
            cpu_once()
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < 1.5:
                cpu_once(); reps += 1
            cpu[th] = (time.perf_counter() - t0) / reps * 1e3
        res["cpu_ms_by_threads"] = {k: round(v, 3) for k, v in cpu.items()}
        best = min(cpu.values())
        res["speedup_vs_cpu_best_threads"] = round(best * 1e3 / graph_us, 1)
        res["speedup_vs_cpu_1_thread"] = round(cpu[1] * 1e3 / graph_us, 1)
        out[name] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
