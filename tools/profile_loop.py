#!/usr/bin/env python3
"""Run a few captured iterations of the C5 loop: the program to put after `rocprofv3 --kernel-trace --`
(per-kernel table: profiles/r01_c5_loop_kernels.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from _loop_scene import c5_scene  # noqa: E402


def main():
    from sdfest_amd.pipeline import FusedRenderAndCompare
    s = c5_scene(views=int(os.environ.get("VIEWS", "1")), max_iterations=10)
    if "FUSED_SINGLE" in os.environ:     # the decoder's fused layer pairs (SDFR_DECODER_OPT_FUSED_SINGLE bits)
        s["decoder"].set_option("fused_single", int(os.environ["FUSED_SINGLE"]))
    # FUSED_RENDER / FC_IN_TAIL = 0: the render pair as two launches / the Linear stack as a launch of its own
    flag = lambda k: None if k not in os.environ else bool(int(os.environ[k]))
    loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"],
                                 form=os.environ.get("FORM", "auto"), fused_render=flag("FUSED_RENDER"),
                                 fc_in_tail=flag("FC_IN_TAIL"))
    for _ in range(3):
        loop(*s["init"], use_graph=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
