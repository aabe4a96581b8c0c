#!/usr/bin/env python3
"""A few captured runs of K objects side by side (MultiObjectRenderAndCompare): the program to put after
`rocprofv3 --kernel-trace --` (tools/trace_full.sh); K from the environment (default 8)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from _loop_scene import c5_scene  # noqa: E402


def main():
    from sdfest_amd.pipeline import MultiObjectRenderAndCompare
    K = int(os.environ.get("K", "8"))
    s = c5_scene(views=1, max_iterations=10)
    multi = MultiObjectRenderAndCompare(s["decoder"], s["camera"], s["config"], K)
    p0, q0, s0, z0 = s["init"]
    multi.rebind(s["targets"].expand(K, -1, -1).contiguous())
    args = (p0.expand(K, 3).contiguous(), q0.expand(K, 4).contiguous(), s0.expand(K).contiguous(), z0.expand(K, 8).contiguous())
    for _ in range(3):
        multi(*args)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
