"""Where the time of SDFVAEViewGenerator.generate() goes (B = 256, 640x480, pointcloud + normalize_pose): the stages
timed one after the other with a synchronisation between them, against the un-synchronised whole."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from _loop_scene import c5_scene
from sdfest_amd.generated_views import SDFVAEViewGenerator, depth_to_pointsets, sample_poses

def main():
    dec = c5_scene()["decoder"]
    dev = torch.device("cuda", 0)
    B = int(os.environ.get("B", 256))
    gcfg = {"width": 640, "height": 480, "fov_deg": 90, "z_min": 0.3, "z_max": 0.7, "extent_mean": 0.15,
            "extent_std": 0.02, "render_threshold": 0.004, "pointcloud": True, "normalize_pose": True}
    gen = SDFVAEViewGenerator(gcfg, dec, batch_size=B, device=dev, seed=0)
    for _ in range(3): gen.generate()
    torch.cuda.synchronize()
    def t(fn, n=10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, r
    us_all, out = t(lambda: gen.generate())
    us_draw, drawn = t(lambda: (torch.randn((B, 8), generator=gen.gen),) + tuple(sample_poses(B, gen.camera, 0.3, 0.7, 0.15, 0.02, gen.gen)))
    z, p, q, s = drawn
    us_h2d, devd = t(lambda: tuple(x.to(dev) for x in (z, p, q, s)))
    zd, pd, qd, sd = devd
    us_render, depth = t(lambda: gen.render(zd, pd, qd, sd))
    depth = depth.clone()
    us_pts, (pts, counts) = t(lambda: depth_to_pointsets(depth, gen.camera))
    def tail():
        owner = torch.repeat_interleave(torch.arange(B, device=dev), counts)
        c = torch.zeros((B, 3), device=dev).index_add_(0, owner, pts) / counts.clamp(min=1)[:, None]
        return pts - c[owner]
    us_tail, _ = t(tail)
    us_split, _ = t(lambda: list(torch.split(pts, counts.tolist())))
    print(f"B={B}: generate() {us_all:.0f} us | draw (CPU) {us_draw:.0f}, H2D {us_h2d:.0f}, decode+render {us_render:.0f}, "
          f"depth_to_pointsets {us_pts:.0f}, normalise {us_tail:.0f}, split {us_split:.0f}; points {pts.shape[0]}", flush=True)
main()
