"""Worker of tests/test_c4_sharded_gpu.py: one rank of BASELINE configs[3] in miniature -- the rank renders
its contiguous shard of the seeded C3/C4 pose list, runs the backward, and the d/dSDF volumes are summed over
the ranks with the ONE all-reduce of sdfest_amd.parallel.  Rank 0 writes the result for the test to compare
with a single-process run of all the views.  (Two ranks share the one GPU of the test box, so the process
group is gloo on device tensors; on a multi-GPU node bench.py takes the same code path over RCCL.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdfest_amd import BatchRenderPlan, Camera                       # noqa: E402
from sdfest_amd.parallel import allreduce_fixed_gradients, allreduce_shared_gradients, shard_views  # noqa: E402
from sdfest_amd.synthetic import blobs_sdf, random_poses              # noqa: E402


def main():
    out_path, n_views, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    det = len(sys.argv) > 5 and sys.argv[5] in ("det", "det_light")   # SDF_GRAD_DETERMINISTIC: integer exchange
    light = len(sys.argv) > 5 and sys.argv[5] == "det_light"           # large runs: per-view checksums, not images
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    b, e = shard_views(n_views, rank, world)
    pos, quat, isc = random_poses(n_views, seed=1, width=W, height=H, f=W / 2.0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    sdf = t(blobs_sdf(0))
    g_all = torch.rand((n_views, H, W), generator=torch.Generator().manual_seed(77)) * 2 - 1
    cam = Camera(W, H, W / 2.0, W / 2.0, W / 2.0, H / 2.0, pixel_center=0.5)
    plan = BatchRenderPlan(64, e - b, cam, device=dev, sdf_grad_mode=0x100 if det else 0)
    depth = plan.forward(sdf, t(pos[b:e]), t(quat[b:e]), t(isc[b:e]), 0.005).clone()
    g_sdf, g_pos, g_quat, g_is = plan.backward(g_all[b:e].to(dev).contiguous(), sdf, t(pos[b:e]), t(quat[b:e]), t(isc[b:e]))
    if det:
        allreduce_fixed_gradients(plan.g_sdf_fixed(), g_sdf)   # int64 sum over the ranks, then one conversion
    else:
        handle = allreduce_shared_gradients(g_sdf, async_op=True)
        handle.wait()
    torch.cuda.synchronize()
    # per-view outputs stay on their rank; gather them only for the comparison
    parts = [None] * world
    if light:   # (sum of the bit patterns and hit count per view: equal images give equal checksums)
        bits = depth.view(torch.int32).to(torch.int64).sum(dim=(1, 2))
        d_out = torch.stack((bits, (depth > 0).sum(dim=(1, 2)))).cpu().numpy()
    else:
        d_out = depth.cpu().numpy()
    dist.all_gather_object(parts, (b, e, d_out, g_pos.cpu().numpy(), g_quat.cpu().numpy(), g_is.cpu().numpy()))
    if rank == 0:
        parts.sort(key=lambda p: p[0])
        np.savez(out_path, g_sdf=g_sdf.cpu().numpy(), depth=np.concatenate([p[2] for p in parts], axis=1 if light else 0),
                 g_pos=np.concatenate([p[3] for p in parts]), g_quat=np.concatenate([p[4] for p in parts]),
                 g_is=np.concatenate([p[5] for p in parts]), spans=np.array([[p[0], p[1]] for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
