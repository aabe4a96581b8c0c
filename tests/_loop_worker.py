"""Worker of tests/test_loop_sharded_gpu.py: one rank of the render-and-compare loop sharded over ranks
(sdfest_amd.pipeline, ``process_group``).  The ranks of the test share the box's one GPU, so the process group is gloo
on device tensors; on a multi-GPU node the same code runs over RCCL (backend "nccl").  Rank 0 writes the trajectory
for the test to compare with a single process."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _loop_scenes  # noqa: E402


def main():
    out_path, scene, flavour, exchange, graph, form = sys.argv[1:7]
    backend = sys.argv[7] if len(sys.argv) > 7 else "gloo"
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    from sdfest_amd.pipeline import FusedRenderAndCompare, RenderAndCompare
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    iterations = int(os.environ.get("SDFR_TEST_ITERATIONS", "0")) or None
    sc = _loop_scenes.build(scene, iterations=iterations)
    mode = (SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES) if flavour == "det" else 0
    # "graph_nohist": no history -> the multi-iteration replay (head | all-reduce | tail + next head | all-reduce | ...)
    hist = None if graph == "graph_nohist" else []
    if form in ("fused", "fused_pose_only"):
        # "fused_pose_only": shape optimisation off (the exchange is the view records alone) and a point constraint
        con = (torch.tensor([0.0, 1.0, 0.0]), torch.tensor([0.1, 0.9, -0.2]), 0.05) if form == "fused_pose_only" else None
        loop = FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["depth"], camera_positions=sc["cam_pos"],
                                     camera_orientations=sc["cam_quat"], shape_optimization=(form == "fused"),
                                     process_group="world", exchange=exchange, sdf_grad_mode=mode, track_inliers=True,
                                     point_constraint=con, graph_collective=(graph == "graph_one"))
        out = loop(*sc["init"], use_graph=graph.startswith("graph"), history=hist)
        if graph == "graph_one":     # what became of the experiment: for the test to report
            print(f"[graph_one] captured={loop.graph_whole_one is not None} error={loop.graph_collective_error}",
                  file=sys.stderr, flush=True)
            with open(out_path + ".graph_one.txt", "w") as f:
                f.write(f"{loop.graph_whole_one is not None}\n{loop.graph_collective_error}\n")
        inl = loop.inlier_history.cpu().numpy()
        shard = (loop.view_begin, loop.view_end)
    else:
        loop = RenderAndCompare(sc["decoder"], sc["camera"], sc["config"], process_group="world")
        out = loop(sc["depth"], *sc["init"], camera_positions=sc["cam_pos"], camera_orientations=sc["cam_quat"],
                   shape_optimization=True, history=hist)
        inl = np.array([float(h["inlier_ratio"]) for h in hist], dtype=np.float32)
        shard = (-1, -1)
    torch.cuda.synchronize()
    final = np.concatenate([o.detach().cpu().numpy().ravel() for o in out])
    traj = _loop_scenes.history_array(hist) if hist else final[None]
    loss = np.array([float(h["loss"]) for h in hist]) if hist else np.zeros(0)
    steps_taken = int(loop.step.item()) if hasattr(loop, "step") else -1
    parts = [None] * world
    dist.all_gather_object(parts, (rank, final, traj, loss, inl, shard))
    if rank == 0:
        parts.sort(key=lambda p: p[0])
        same = all(np.array_equal(p[1], parts[0][1]) and np.array_equal(p[2], parts[0][2], equal_nan=True)
                   and np.array_equal(p[3], parts[0][3], equal_nan=True) and np.array_equal(p[4], parts[0][4], equal_nan=True)
                   for p in parts)
        np.savez(out_path, final=final, traj=traj, loss=loss, inlier=inl, ranks_identical=same,
                 shards=np.array([p[5] for p in parts]), steps_taken=steps_taken)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
