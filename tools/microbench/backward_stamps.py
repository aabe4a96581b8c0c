"""In-kernel stamps of the backward's hit tiles.  Build the library with -DSDFR_STAMPS
(tools/microbench/build_variant.sh stamps -DSDFR_STAMPS) and pass its path; prints where a hit tile's
time goes (median / mean s_memtime ticks per phase, wave 0 of each tile)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sdfest_amd import _lib
from sdfest_amd.synthetic import blobs_sdf, random_poses
L = ctypes.CDLL(sys.argv[1])
for name, (res, args) in _lib.SIGNATURES.items():
    if hasattr(L, name):
        fn = getattr(L, name); fn.restype = res; fn.argtypes = args
B, W, H = 256, 640, 480
dev = torch.device("cuda:0")
sdf = torch.tensor(blobs_sdf(0), device=dev)
pos, quat, isc = (torch.tensor(a, device=dev) for a in random_poses(B, seed=1))
g = torch.rand((B, H, W), device=dev) * 2 - 1
depth = torch.empty((B, H, W), device=dev)
gs = torch.empty((64, 64, 64), device=dev); gp = torch.empty((B, 3), device=dev); gq = torch.empty((B, 4), device=dev); gi = torch.empty((B,), device=dev)
nb = max(L.sdfr_render_forward_workspace_bytes(64, B, W, H), L.sdfr_render_backward_workspace_bytes(64, B, W, H))
ws = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert L.sdfr_render_forward(sdf.data_ptr(), 64, 0, pos.data_ptr(), quat.data_ptr(), isc.data_ptr(), B, W, H, 320.0, 240.0, 320.0, 320.0, 0.005, depth.data_ptr(), ws.data_ptr(), ws.numel(), 0, st) == 0
def bwd():
    assert L.sdfr_render_backward(g.data_ptr(), depth.data_ptr(), sdf.data_ptr(), 64, 0, pos.data_ptr(), quat.data_ptr(), isc.data_ptr(), B, W, H, 320.0, 240.0, 320.0, 320.0, 0, gs.data_ptr(), 0, gp.data_ptr(), gq.data_ptr(), gi.data_ptr(), ws.data_ptr(), ws.numel(), 0, st) == 0
bwd(); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * (153600 * 16))()
L.sdfr_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.sdfr_debug_stamps(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); bwd(); e1.record(); torch.cuda.synchronize()
L.sdfr_debug_stamps(out, 0)
a = np.ctypeslib.as_array(out).reshape(153600, 16).astype(np.float64)
hit = a[:, 5] > 0
a = a[hit]
print(f"backward (instrumented) {e0.elapsed_time(e1)*1e3:.1f} us; hit tiles {hit.sum()}")
# stamps: 0 start (after the depth loads), 1 after clear/max/barriers, 6 gathers of sub-tile 0 back, 7 derivative
# arithmetic of sub-tile 0 done, 8 table adds of sub-tile 0 done, 2 both sub-tiles done, 3 pose sums + barrier,
# 4 flush issued, 5 flush acknowledged
order = [(0, 1, "clear table, tile max, 2 barriers"), (1, 6, "sub-tile 0: ray, cell, grid gathers"),
         (6, 7, "sub-tile 0: derivative arithmetic"), (7, 8, "sub-tile 0: table look-ups + adds"),
         (8, 2, "sub-tile 1 (all of it)"), (2, 3, "pose sums, barrier (slowest wave)"), (3, 4, "flush issue"),
         (4, 5, "flush acknowledged")]
ok = (a[:, 6] > 0) & (a[:, 7] > 0) & (a[:, 8] > 0)
b_ = a[ok]
tot = np.median(b_[:, 5] - b_[:, 0])
print(f"tiles whose wave 0 has a hit in sub-tile 0: {ok.sum()}; median ticks start -> end: {tot:.0f}")
for i0, i1, name in order:
    dlt = b_[:, i1] - b_[:, i0]
    print(f"  {name:46s} median {np.median(dlt):8.0f}  mean {dlt.mean():8.0f}  ({100*dlt.mean()/(b_[:,5]-b_[:,0]).mean():5.1f} %)")
