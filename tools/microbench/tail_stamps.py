"""Where the time of loop_tail_kernel goes (C5 loop, 10 ns ticks between the stages of its last run).
Needs a build with -DSDFR_TAIL_STAMPS:  tools/microbench/build_variant.sh stamps -DSDFR_TAIL_STAMPS
                                        SDFR_LIB=build/variants/libsdfr_stamps.so python tools/microbench/tail_stamps.py
A build with -DSDFR_BT_STAMPS prints the stages of the first workgroup of every transposed-resize launch of the same
loop instead (device printf; the table at the end is then meaningless: the library has no tail stamps)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from sdfest_amd import _lib
from _loop_scene import c5_scene   # the C5 scene of tools/profile_loop.py


def main():
    from sdfest_amd.pipeline import FusedRenderAndCompare
    sc = c5_scene(int(os.environ.get("VIEWS", "1")))
    args = sc["init"]
    fused = FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["targets"])
    fused(*args)
    torch.cuda.synchronize()
    fused(*args)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    if not hasattr(_lib.lib(), "sdfr_debug_tail_stamps"):
        return
    fn = _lib.lib().sdfr_debug_tail_stamps
    fn.restype = ctypes.c_int
    assert fn(out) == 0
    t = np.array(list(out), dtype=np.int64)
    names = ["fc backward", "view reductions + chain", "constraint + barrier", "adam", "next poses"]
    for k, n in enumerate(names):
        print(f"{n:28s} {(t[k + 1] - t[k]) * 0.01:6.2f} us")
    print(f"{'total':28s} {(t[5] - t[0]) * 0.01:6.2f} us")


main()
