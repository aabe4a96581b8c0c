#!/usr/bin/env python3
"""A few SDFPipeline.__call__ on the C5 image: the program to put after `rocprofv3 --kernel-trace --` to see what a
call launches around its 50 iterations (tools/front_door_sequence.sh prints the last call's launches up to the loop)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tools.bench_extra import c5_scene, front_door_times  # noqa: E402

if __name__ == "__main__":
    sc = c5_scene()
    print(front_door_times(sc, [sc["targets"], sc["targets"]]))
    torch.cuda.synchronize()
