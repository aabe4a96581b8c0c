"""Shared helpers for the parity tests (test-side only)."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

POSE_NAMES = ["px", "py", "pz", "qx", "qy", "qz", "qw", "inv_scale"]


def render_cases():
    return sorted(os.path.basename(p)[len("render_"):-len(".npz")]
                  for p in glob.glob(os.path.join(GOLDEN, "render_*.npz")))


def load_render_case(name):
    d = np.load(os.path.join(GOLDEN, f"render_{name}.npz"))
    c = {k: d[k] for k in d.files}
    from oracle import blobs_sdf, sphere_sdf
    c["sdf"] = {"sphere": lambda: sphere_sdf(0.5), "blobs0": lambda: blobs_sdf(0)}[str(c["sdf_kind"])]()
    for k in ("W", "H"):
        c[k] = int(c[k])
    for k in ("fov", "thr", "inv_scale", "fx", "fy", "cx", "cy"):
        c[k] = float(c[k])
    return c


def dense_from_sparse(idx, val, R=64):
    g = np.zeros((R, R, R), dtype=np.float64)
    if len(idx):
        g[tuple(np.asarray(idx).T)] = val
    return g


def rel_err(a, b, floor=None):
    """max |a-b| / max(|b|, floor); floor defaults to max|b| (tensor-relative)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if floor is None:
        floor = np.max(np.abs(b)) if b.size else 1.0
    floor = max(float(floor), 1e-300)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def decoder_probe_G(n=64):
    """The deterministic upstream gradient used by tools/make_goldens.py for the decoder VJP."""
    a = np.arange(n, dtype=np.float64)
    X, Y, Z = np.meshgrid(a, a, a, indexing="ij")
    return (np.sin(0.37 * X + 0.11) * np.cos(0.23 * Y - 0.4) + 0.5 * np.sin(0.05 * Z * X * 0.1 + 0.3 * Y)).astype(np.float32)


def check_sdf_grad(hip, ref, mode, n_hits, rel=1e-4, name=""):
    """d/dSDF of the HIP backward against the oracle's.  mode 0 (the exact trilinear weights): within `rel` of the
    largest entry, everywhere -- the weights are continuous across cell faces, so a hit point within rounding of a face
    gives the same contributions whichever cell it is filed under.  mode 1 (the weights the reference's GPU extension
    adds, sdf_renderer_cuda.cu:373-388) is NOT continuous there: corner (0,0,0) of the cell gets (1-x)(1-y) z, not
    (1-x)(1-y)(1-z), so a hit point that float and double arithmetic file under neighbouring cells (a float hit point is
    good to ~3e-5 cells: a few pixels in ten thousand) puts one pixel's weight on other voxels -- up to 12 of them, the
    two cells' corners.  Those voxels are allowed for by COUNT; every other voxel is held to `rel`."""
    hip = np.asarray(hip, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    top = float(np.max(np.abs(ref))) if ref.size else 0.0
    if top == 0.0:
        assert not hip.any(), name
        return 0
    bad = np.abs(hip - ref) > rel * top
    if mode == 0:
        assert not bad.any(), f"{name}: d/dSDF off by {np.max(np.abs(hip - ref)) / top:.2e} of its maximum"
        return 0
    allowed = 12 * max(2, int(1e-3 * n_hits))
    assert bad.sum() <= allowed, f"{name}: {int(bad.sum())} voxels off (mode 1 allows for {allowed}: pixels on cell faces)"
    # ... and those differ by single pixels' weights, not by a share of the whole volume
    assert np.abs(hip - ref)[bad].sum() <= 0.02 * np.abs(ref).sum() + 1e-30, name
    return int(bad.sum())
