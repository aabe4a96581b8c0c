"""The single decode's fused layer pairs (sdfr_decoder_set_option, SDFR_DECODER_OPT_FUSED_SINGLE) against the launches
they replace: bitwise agreement per pair on the test shapes (a report, not an assertion), decode / decode + VJP times
per setting, and the C5 loop per iteration (run on the GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from _loop_scene import c5_scene  # noqa: E402
import test_decoder_gpu as T  # noqa: E402
from sdfest_amd import SDFDecoder  # noqa: E402
from sdfest_amd.pipeline import FusedRenderAndCompare  # noqa: E402


def event_us(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    d = np.load(os.path.join(T.GOLDEN, "decoder_mug.npz"))
    w = np.load(os.path.join(T.GOLDEN, "mug_decoder_weights.npz"))
    wts = {k: w[k] for k in w.files}
    rng = np.random.default_rng(23)
    only = os.environ.get("ONLY")
    for name, cfg, state, volume, latent, N in T._fused_single_cases(d, wts, rng):
        if only and only not in name:
            continue
        dec = SDFDecoder.from_config(cfg, state, sdf_size=volume)
        z_np = rng.normal(size=(N, latent)).astype(np.float32)
        G = torch.tensor(rng.normal(size=(N, 1, volume, volume, volume)).astype(np.float32), device="cuda")
        res = {}
        for bits in (0, 1, 2, 4, 12, 15, 16):     # (16: bits 0 again -- is the unfused form itself reproducible?)
            dec.set_option("fused_single", bits & 15)
            z = torch.tensor(z_np, device="cuda", requires_grad=True)
            try:
                o = dec.decode(z)
                o.backward(G)
                torch.cuda.synchronize()
                res[bits] = (o.detach().clone(), z.grad.clone())
            except RuntimeError as e:
                print(f"{name} N={N} bits={bits}: {e}", flush=True)
        for bits in (1, 2, 4, 12, 15, 16):
            if bits not in res or 0 not in res:
                continue
            o, g = res[bits]
            do = (o - res[0][0]).abs().max().item()
            dg = (g - res[0][1]).abs().max().item()
            bad_o = int((o != res[0][0]).sum().item())
            print(f"{name:55s} N={N:2d} bits={bits}: out {'==' if torch.equal(o, res[0][0]) else f'DIFF {do:.3e} ({bad_o} values of {o.numel()})':30s} "
                  f"grad {'==' if torch.equal(g, res[0][1]) else f'DIFF {dg:.3e} of max {res[0][1].abs().max().item():.3e}'}  (out max {res[0][0].abs().max().item():.3e})", flush=True)
    if os.environ.get("NO_TIMES"):
        return
    # times: the mug decoder, one latent
    cfg = T.mug_config(d)
    dec = SDFDecoder.from_config(cfg, wts)
    z0 = torch.zeros(1, 8, device="cuda")
    G = torch.ones(1, 1, 64, 64, 64, device="cuda")
    for bits in (0, 1, 5, 13, 15, 0, 5):
        dec.set_option("fused_single", bits)

        def fwd():
            with torch.no_grad():
                dec.decode(z0)

        def both():
            z = z0.clone().requires_grad_(True)
            dec.decode(z).backward(G)
        for _ in range(10):
            fwd(); both()
        torch.cuda.synchronize()
        print(f"bits {bits}: decode {event_us(fwd, 200):7.2f} us (host-driven), decode + VJP {event_us(both, 100):7.2f} us", flush=True)
    s = c5_scene(views=1, max_iterations=50)
    for bits in (0, 5, 1, 4, 13, 7, 15, 0, 5):
        s["decoder"].set_option("fused_single", bits)
        loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], graph_iterations=5)
        out = loop(*s["init"])
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            loop.rebind(s["targets"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = loop(*s["init"])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3 / 50)
        err = (out[0] - s["p_true"]).norm().item() * 1e3
        print(f"C5 loop, fused_single {bits}: {np.median(ts):.4f} ms per iteration (min {min(ts):.4f}), final position error {err:.3f} mm", flush=True)


if __name__ == "__main__":
    main()
