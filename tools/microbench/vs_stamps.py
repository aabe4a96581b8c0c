"""The stages of the first workgroup of every fused VJP stage of the C5 loop (a build with -DSDFR_VS_STAMPS:
tools/microbench/build_variant.sh vs -DSDFR_VS_STAMPS; SDFR_LIB=build/variants/libsdfr_vs.so python tools/microbench/vs_stamps.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from _loop_scene import c5_scene
from sdfest_amd.pipeline import FusedRenderAndCompare

sc = c5_scene(1, max_iterations=3)
sc["decoder"].set_option("fused_single", int(os.environ.get("FUSED_SINGLE", "7")))
fused = FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["targets"])
fused(*sc["init"], use_graph=False)
torch.cuda.synchronize()
