"""The one-launch render step against the two launches when the object FILLS the image (many hit pixels: the float
atomics of the one-launch step collide more): the C5 scene at several distances (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from _loop_scene import c5_scene  # noqa: E402
from sdfest_amd import render_depth_gpu  # noqa: E402
from sdfest_amd.pipeline import FusedRenderAndCompare  # noqa: E402


def main():
    s = c5_scene(views=1, max_iterations=50)
    dec, cam = s["decoder"], s["camera"]
    p0, q0, s0, z0 = s["init"]
    for z in (-0.5, -0.3, -0.2, -0.15, -0.12):
        p_true = s["p_true"].clone()
        p_true[0, 2] = z
        p_true[0, 0] = p_true[0, 1] = 0.0
        with torch.no_grad():
            sdf = dec.decode(torch.zeros(1, 8, device="cuda"))[0, 0]
            tgt = render_depth_gpu(sdf, p_true[0], q0[0], 1 / s0[0], None, None, None, 0.005, cam)[None].contiguous()
        hits = int((tgt > 0).sum())
        row = {}
        for fused in (False, True):
            loop = FusedRenderAndCompare(dec, cam, s["config"], tgt, fused_render=fused)
            init = (p_true + 0.004, q0, s0 * 1.03, z0)
            for _ in range(2):
                loop(*init)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                loop(*init)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 50)
            row[fused] = sorted(ts)[2]
        print(f"distance {-z:.2f} m: {hits:6d} observed pixels: two launches {row[False]:.4f} ms, one launch {row[True]:.4f} ms "
              f"per iteration", flush=True)


if __name__ == "__main__":
    main()
