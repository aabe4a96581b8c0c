"""The CPU port's backward by thread count, phases on stderr (oracle/libsdfr_oracle_native.so, sdfo_set_timing)."""
import ctypes, os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libsdfr_oracle_native.so"], stdout=subprocess.DEVNULL)
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsdfr_oracle_native.so"))
B, W, H, f = 256, 640, 480, 320.0
sdf = oracle.blobs_sdf(0)
pos, quat, isc = oracle.random_poses(B, seed=1, width=W, height=H, f=f)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
cd, ci = ctypes.c_double, ctypes.c_int
dep = np.empty((B, H, W), np.float32); g = np.ones((B, H, W), np.float32)
gs = np.empty((64, 64, 64), np.float32); gp = np.empty((B, 3), np.float32); gq = np.empty((B, 4), np.float32); gi = np.empty(B, np.float32)
lib.sdfo_set_timing(1)
for th in (8, 16, 32, 64, 128, 256):
    lib.sdfo_set_threads(ci(th))
    for rep in range(2):
        t0 = time.perf_counter()
        lib.sdfo_render_forward_f32(P(sdf), ci(64), P(pos), P(quat), P(isc), ci(B), ci(W), ci(H), cd(W / 2), cd(H / 2), cd(f), cd(f), cd(0.005), P(dep), None, None, ci(0))
        t1 = time.perf_counter()
        lib.sdfo_render_backward_f32(P(g), P(dep), P(sdf), ci(64), P(pos), P(quat), P(isc), ci(B), ci(W), ci(H), cd(W / 2), cd(H / 2), cd(f), cd(f), ci(0), P(gs), P(gp), P(gq), P(gi))
        t2 = time.perf_counter()
    print(f"{th} threads: forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f} ms -> {B / (t2 - t0):.0f} renders/s", flush=True)
