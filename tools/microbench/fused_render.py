"""The render pair of the captured loop as ONE launch (sdfr_render_step_fused_l1_pc) against the two launches it
replaces: ms per iteration of the C5 loop, and the depth images of one call side by side (run on the GPU box).
(FusedRenderAndCompare(fused_render=False / True))."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from _loop_scene import c5_scene  # noqa: E402
from sdfest_amd.pipeline import FusedRenderAndCompare  # noqa: E402


def run(views, fused, n=7, shape=True, fc=None):
    s = c5_scene(views=views, max_iterations=50)
    loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], form="tail", fused_render=fused,
                                 shape_optimization=shape, fc_in_tail=fc if fused else None)
    out = None
    for _ in range(3):
        out = loop(*s["init"], use_graph=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = loop(*s["init"], use_graph=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50)
    ts.sort()
    return ts[len(ts) // 2], [o.detach().cpu() for o in out]


def main():
    for views, shape in ((1, True), (1, False), (2, True), (3, True), (4, True), (7, True), (3, False)):
        res = {}
        for fused, fc in ((False, None), (True, False), (True, True)):
            if fc and (not shape or views > 1):
                continue
            res[fused, fc] = run(views, fused, shape=shape, fc=fc)
            print(f"views {views} shape optimisation {shape}: render pair as one launch {fused}, Linear stack in the tail's "
                  f"launch {bool(fc)}: {res[fused, fc][0]:.4f} ms per iteration", flush=True)
        for a, b, name in zip(res[False, None][1], res[True, False][1], ("position", "orientation", "scale", "latent")):
            print(f"   {name}: max |two launches - one| = {(a - b).abs().max().item():.3e}")


if __name__ == "__main__":
    main()
