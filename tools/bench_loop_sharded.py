#!/usr/bin/env python3
"""The render-and-compare loop in its sharded form (head | one all-reduce | tail, sdfest_amd.pipeline) under
torch.distributed.run: ms per iteration for the C5 scene seen from VIEWS cameras (the same image repeated), each
exchange, against the single-process loop of the same views.  One rank on a one-GPU box measures what the form itself
costs (two graphs and a collective per iteration instead of one graph per five iterations); N ranks on a node what the
exchange over xGMI costs.

    python -m torch.distributed.run --nproc-per-node N tools/bench_loop_sharded.py        (VIEWS=8 by default)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from _loop_scene import c5_scene
    from sdfest_amd.pipeline import FusedRenderAndCompare
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # (SDFR_BENCH_SHARE_GPU=1 SDFR_BENCH_BACKEND=gloo: a rehearsal with every rank on GPU 0, never a measurement)
    dev = torch.device("cuda", 0 if os.environ.get("SDFR_BENCH_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(os.environ.get("SDFR_BENCH_BACKEND", "nccl"), rank=rank, world_size=world,
                            **({"device_id": dev} if os.environ.get("SDFR_BENCH_BACKEND", "nccl") == "nccl" else {}))
    views = int(os.environ.get("VIEWS", "8"))
    s = c5_scene(views=views, max_iterations=50)
    out = {"workload": f"C5 scene, {views} views of 640x480, mug decoder, 50 Adam iterations, {world} rank(s), "
                       f"backend {dist.get_backend()}"}

    def time_loop(loop):
        loop(*s["init"])
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = loop(*s["init"])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 50 * 1e3)
        return round(float(np.median(ts)), 4), res

    # the single-process loop of the same views first, in both of its forms (every rank runs them: time_loop has barriers)
    for form in ("tail", "records"):
        if form == "tail" and views > 64:
            continue
        single = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], form=form)
        out[f"ms_per_iteration_single_process_{form}_form"], _ = time_loop(single)
        del single
    for exchange in ("sdf", "latent"):
        loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], process_group="world",
                                     exchange=exchange)
        ms, res = time_loop(loop)
        out[f"ms_per_iteration_sharded_exchange_{exchange}"] = ms
        out[f"final_position_error_mm_{exchange}"] = round((res[0] - s["p_true"]).norm().item() * 1e3, 3)
        del loop
        if os.environ.get("SDFR_BENCH_GRAPH_COLLECTIVE", "1") == "1" and dist.get_backend() == "nccl":
            # the experiment: the all-reduce captured INSIDE the graphs (whole iterations as one graph)
            loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], process_group="world",
                                         exchange=exchange, graph_collective=True)
            ms, res = time_loop(loop)
            out[f"ms_per_iteration_sharded_exchange_{exchange}_collective_in_graph"] = (
                ms if loop.graph_whole_one is not None else None)
            out[f"collective_in_graph_{exchange}"] = ("captured" if loop.graph_whole_one is not None
                                                      else f"refused: {loop.graph_collective_error}")
            out[f"final_position_error_mm_{exchange}_collective_in_graph"] = round(
                (res[0] - s["p_true"]).norm().item() * 1e3, 3)
            del loop
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
