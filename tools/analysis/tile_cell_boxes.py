"""Backward: how big is the box of grid cells under the hit pixels of a 64x8 tile?  (CPU, oracle depth)
Decides whether a direct-indexed LDS brick could replace the run hash of the d/dSDF pre-summation."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle
from sdfest_amd.synthetic import blobs_sdf, random_poses

W, H, f, thr, R = 640, 480, 320.0, 0.005, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sdf = blobs_sdf(0)
pos, quat, isc = random_poses(256, seed=1)
pos, quat, isc = pos[:B], quat[:B], isc[:B]
oracle.set_threads(8)
depth = oracle.render_forward(sdf, pos, quat, isc, W, H, W / 2, H / 2, f, f, thr)

def rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

cols, rows = np.meshgrid(np.arange(W), np.arange(H))
dx = (cols + 0.5 - W / 2) / f; dy = -(rows + 0.5 - H / 2) / f
d = np.stack([dx, dy, -np.ones_like(dx)], -1); d /= np.linalg.norm(d, axis=-1, keepdims=True)
vols, dims, distinct = [], [], []
for (tw, th) in ((64, 8), (32, 8)):
    vols, distinct, dims = [], [], []
    for b in range(B):
        Rm = rot(quat[b].astype(np.float64))
        t = depth[b] / (-d[..., 2])
        o = (t[..., None] * d - pos[b]) @ Rm          # R^T (t d - p)
        g = (o * isc[b] + 1.0) * (R - 1) / 2
        base = np.clip(np.floor(g), 0, R - 2).astype(int)
        hit = depth[b] > 0
        for ty in range(H // th):
            for tx in range(W // tw):
                sl = (slice(ty * th, ty * th + th), slice(tx * tw, tx * tw + tw))
                m = hit[sl]
                if not m.any():
                    continue
                c = base[sl][m]
                ext = c.max(0) - c.min(0) + 2          # corners reach one further
                vols.append(int(np.prod(ext))); dims.append(ext)
                lin = (c[:, 0] * R + c[:, 1]) * R + c[:, 2]
                distinct.append(len(np.unique(lin)))
    vols = np.array(vols); distinct = np.array(distinct); dims = np.array(dims)
    print(f"tile {tw}x{th}: {len(vols)} hit tiles; box volume (voxels) median {np.median(vols):.0f} p90 {np.percentile(vols, 90):.0f} "
          f"p99 {np.percentile(vols, 99):.0f} max {vols.max()}; <=1024: {np.mean(vols <= 1024):.3f} <=2048: {np.mean(vols <= 2048):.3f} "
          f"<=4096: {np.mean(vols <= 4096):.3f}; distinct cells median {np.median(distinct):.0f} max {distinct.max()}; "
          f"mean box dims {dims.mean(0).round(1)}")
