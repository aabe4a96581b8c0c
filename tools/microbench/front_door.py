"""SDFPipeline.__call__ end to end on the C5 image(s): the initialisation network in its resident (captured, nothing
read back) and in its host-driven form (run on the GPU box)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import bench_extra  # noqa: E402
from tools.bench_extra import c5_scene, front_door_times  # noqa: E402
from sdfest_amd import render_depth_gpu  # noqa: E402
import sdfest_amd.simple_setup as ss  # noqa: E402

sc = c5_scene()
dev = sc["targets"].device
others = []
with torch.no_grad():
    for dp in ((0.03, 0.02, -0.02), (-0.04, 0.01, 0.05)):
        others.append(render_depth_gpu(sc["decoder"].decode(torch.zeros(1, 8, device=dev))[0, 0],
                                       (sc["p_true"] + torch.tensor([dp], device=dev))[0], sc["q_true"][0],
                                       1 / sc["s_true"][0], None, None, None, 0.005, sc["camera"])[None].contiguous())
orig = ss.SDFPipeline.__init__
for resident in (True, False):
    def patched(self, *a, _r=resident, **k):
        k["resident_init"] = _r
        return orig(self, *a, **k)
    ss.SDFPipeline.__init__ = patched
    print("resident_init", resident, json.dumps(front_door_times(sc, others)), flush=True)
ss.SDFPipeline.__init__ = orig
