"""Time forward/backward of several builds of the library (ablation / variant .so files)."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sdfest_amd import synthetic as oracle   # input generators only
from sdfest_amd import _lib

def load(path):
    h = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if not hasattr(h, name): continue   # an older build
        fn = getattr(h, name); fn.restype = res; fn.argtypes = args
    return h

def main():
    B, W, H = int(os.environ.get("B", 256)), int(os.environ.get("W", 640)), int(os.environ.get("H", 480))
    f, cxx, cyy = W / 2.0, W / 2.0, H / 2.0
    dev = torch.device("cuda:0")
    sdf = torch.tensor(oracle.blobs_sdf(0), device=dev)
    pos, quat, isc = (torch.tensor(a, device=dev) for a in oracle.random_poses(B, seed=1, width=W, height=H, f=f))
    g = torch.rand((B, H, W), device=dev) * 2 - 1
    depth = torch.empty((B, H, W), device=dev)
    gs = torch.empty((64, 64, 64), device=dev); gp = torch.empty((B, 3), device=dev)
    gq = torch.empty((B, 4), device=dev); gi = torch.empty((B,), device=dev)
    mode = os.environ.get("MODE", "")
    thr = float(os.environ.get("THR", "0.005"))
    if mode == "samepose":
        pos = torch.tensor([[0.0, 0.0, -1.5]], device=dev).repeat(B, 1).contiguous()
        quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev).repeat(B, 1).contiguous()
        isc = torch.full((B,), 2.0, device=dev)
    if mode == "mug":   # mug-sized objects at 0.4-0.6 m: ~1.1 pixels per voxel
        pos = pos * torch.tensor([0.3, 0.3, 0.3], device=dev)
        isc = torch.full((B,), 1 / 0.055, device=dev)
    if "ISC_MUL" in os.environ:   # smaller objects at the same places: fewer pixels per voxel
        isc = (isc * float(os.environ["ISC_MUL"])).contiguous()
    if mode == "nosurface":
        sdf = torch.ones_like(sdf)
    if mode == "offscreen":
        pos = pos.clone(); pos[:, 0] += 100.0
    libs = []
    for path in sys.argv[1:]:
        L = load(path)
        nb = max(L.sdfr_render_forward_workspace_bytes(64, B, W, H), L.sdfr_render_backward_workspace_bytes(64, B, W, H))
        if hasattr(L, "sdfr_render_step_workspace_bytes"): nb = max(nb, L.sdfr_render_step_workspace_bytes(64, B, W, H))
        libs.append((path, L, torch.empty(nb + 256, dtype=torch.uint8, device=dev)))
    st = torch.cuda.current_stream().cuda_stream
    def fwd(L, ws):
        rc = L.sdfr_render_forward(sdf.data_ptr(), 64, 0, pos.data_ptr(), quat.data_ptr(), isc.data_ptr(), B, W, H,
                              cxx, cyy, f, f, thr, depth.data_ptr(), ws.data_ptr(), ws.numel(), 0, st)
        assert rc == 0, L.sdfr_last_error()
    def bwd(L, ws):
        rc = L.sdfr_render_backward(g.data_ptr(), depth.data_ptr(), sdf.data_ptr(), 64, 0, pos.data_ptr(), quat.data_ptr(),
                               isc.data_ptr(), B, W, H, cxx, cyy, f, f, 0, gs.data_ptr(), 0, gp.data_ptr(),
                               gq.data_ptr(), gi.data_ptr(), ws.data_ptr(), ws.numel(), 0, st)
        assert rc == 0, L.sdfr_last_error()
    gs2 = torch.empty_like(gs)
    def sfwd(L, ws):
        rc = L.sdfr_render_step_forward(sdf.data_ptr(), 64, 0, pos.data_ptr(), quat.data_ptr(), isc.data_ptr(), B, W, H,
                                        cxx, cyy, f, f, thr, depth.data_ptr(), gs2.data_ptr(), 0, ws.data_ptr(), ws.numel(), 0, st)
        assert rc == 0, L.sdfr_last_error()
    def sbwd(L, ws):
        rc = L.sdfr_render_step_backward(g.data_ptr(), depth.data_ptr(), sdf.data_ptr(), 64, 0, B, W, H, cxx, cyy, f, f, 0,
                                         gs2.data_ptr(), 0, gp.data_ptr(), gq.data_ptr(), gi.data_ptr(), ws.data_ptr(),
                                         ws.numel(), 0, st)
        assert rc == 0, L.sdfr_last_error()
    def pair(L, ws): fwd(L, ws); bwd(L, ws)
    def step(L, ws): sfwd(L, ws); sbwd(L, ws)
    rounds = int(os.environ.get("ROUNDS", 7))
    res = {path: {"fwd": [], "bwd": [], "pair": [], "step": [], "sfwd": []} for path, _, _ in libs}
    with_step = os.environ.get("STEP", "1") == "1"
    for r in range(rounds + 1):
        for path, L, ws in libs:
            fns = [("fwd", fwd), ("bwd", bwd)]
            if with_step:
                fns.append(("pair", pair))
                if hasattr(L, "sdfr_render_step_forward"):
                    fns += [("sfwd", sfwd), ("step", step)]
            for name, fn in fns:
                fn(L, ws); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 10
                e0.record()
                for _ in range(n): fn(L, ws)
                e1.record(); torch.cuda.synchronize()
                if r > 0: res[path][name].append(e0.elapsed_time(e1) / n * 1e3)
    hits = int((depth > 0).sum())
    for path, _, _ in libs:
        f = float(np.median(res[path]["fwd"])); b = float(np.median(res[path]["bwd"]))
        print(f"{os.path.basename(path):40s} B={B} fwd {f:8.1f} us  bwd {b:8.1f} us  hits={hits} "
              f"-> {B/((f+b)*1e-6):,.0f} renders/s  (median of {rounds} interleaved rounds; "
              f"fwd min {min(res[path]['fwd']):.1f} bwd min {min(res[path]['bwd']):.1f})"
              + "".join(f"  {k} {float(np.median(v)):.1f} us" for k, v in res[path].items() if v and k not in ("fwd", "bwd")),
              flush=True)
main()
