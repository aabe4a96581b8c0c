"""Backward, round 3: the box of grid VOXELS (cells + 1) under the hit pixels of a backward tile, for the tile shape the
kernel really uses per view (32x32 pixels at >= 2 pixels per voxel, else 64x8; common.hpp, kBwdBigTile) -- sizes the
direct-indexed LDS box that replaces the run hash.  CPU only (oracle depth)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle
from sdfest_amd.synthetic import blobs_sdf, random_poses

W, H, f, thr, R = 640, 480, 320.0, 0.005, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else "c3"
sdf = blobs_sdf(0)
pos, quat, isc = random_poses(256, seed=1)
pos, quat, isc = pos[:B].copy(), quat[:B].copy(), isc[:B].copy()
if mode == "mug":
    pos *= 0.3
    isc[:] = 1 / 0.055
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
vols, dims, distinct, hits, big = [], [], [], [], []
for b in range(B):
    scale = 1.0 / isc[b]
    r = f * (scale / ((R - 1) / 2)) / np.linalg.norm(pos[b])
    tw, th = (32, 32) if r >= 2.0 else (64, 8)
    Rm = rot(quat[b].astype(np.float64))
    t = depth[b] / (-d[..., 2])
    o = (t[..., None] * d - pos[b]) @ Rm
    g = (o * isc[b] + 1.0) * (R - 1) / 2
    base = np.clip(np.floor(g), 0, R - 2).astype(int)
    hit = depth[b] > 0
    for ty in range((H + th - 1) // th):
        for tx in range((W + tw - 1) // tw):
            sl = (slice(ty * th, ty * th + th), slice(tx * tw, tx * tw + tw))
            m = hit[sl]
            if not m.any():
                continue
            c = base[sl][m]
            ext = c.max(0) - c.min(0) + 2
            vols.append(int(np.prod(ext))); dims.append(ext); hits.append(int(m.sum())); big.append(tw == 32)
            # z padded to even (pairs)
            lin = (c[:, 0] * R + c[:, 1]) * R + c[:, 2]
            distinct.append(len(np.unique(lin)))
vols = np.array(vols); dims = np.array(dims); hits = np.array(hits); big = np.array(big); distinct = np.array(distinct)
print(f"{mode}: {B} views, {big.mean():.2f} of hit tiles are 32x32; {len(vols)} hit tiles, {hits.sum()} hit pixels")
for cap in (1024, 2048, 3072, 4096, 4608, 5120, 6144, 8192):
    ok = vols <= cap
    print(f"  capacity {cap:5d} voxels: {ok.mean():.3f} of hit tiles fit, holding {hits[ok].sum() / hits.sum():.3f} of the hit pixels")
print(f"  box volume median {np.median(vols):.0f} p75 {np.percentile(vols,75):.0f} p90 {np.percentile(vols,90):.0f} p99 {np.percentile(vols,99):.0f} max {vols.max()}")
print(f"  mean dims {dims.mean(0).round(1)}; distinct cells per tile median {np.median(distinct):.0f} p90 {np.percentile(distinct,90):.0f}")

# ---- the box the kernel can know WITHOUT a per-pixel pre-pass: the 8 corners of the tile's frustum chunk between the
# smallest and the largest depth of its hit pixels, mapped to grid coordinates (an affine map: every hit point lies in
# their convex hull), floor/clamp like the cells, +1 for the far corners, z extent padded to a multiple of 4.
def frustum_fit(B):
    out = []
    for b in range(B):
        scale = 1.0 / isc[b]
        r = f * (scale / ((R - 1) / 2)) / np.linalg.norm(pos[b])
        tw, th = (32, 32) if r >= 2.0 else (64, 8)
        Rm = rot(quat[b].astype(np.float64))
        hit = depth[b] > 0
        for ty in range((H + th - 1) // th):
            for tx in range((W + tw - 1) // tw):
                sl = (slice(ty * th, ty * th + th), slice(tx * tw, tx * tw + tw))
                m = hit[sl]
                if not m.any():
                    continue
                z = depth[b][sl][m]
                zmin, zmax = z.min(), z.max()
                pts = []
                for (u, v) in ((tx * tw, ty * th), (tx * tw + tw - 1, ty * th), (tx * tw, ty * th + th - 1), (tx * tw + tw - 1, ty * th + th - 1)):
                    ddx = (u + 0.5 - W / 2) / f; ddy = -(v + 0.5 - H / 2) / f
                    for zz in (zmin, zmax):
                        p = np.array([ddx * zz, ddy * zz, -zz])   # depth = -z of the point
                        o = (p - pos[b]) @ Rm
                        pts.append((o * isc[b] + 1.0) * (R - 1) / 2)
                pts = np.array(pts)
                lo = np.clip(np.floor(pts.min(0) - 1e-3), 0, R - 2).astype(int)
                hi = np.clip(np.floor(pts.max(0) + 1e-3), 0, R - 2).astype(int) + 1
                ext = hi - lo + 1
                ext[2] = (ext[2] + 3) // 4 * 4
                out.append((int(np.prod(ext)), int(m.sum())))
    return np.array(out)
fr = frustum_fit(B)
print("  conservative frustum box (no per-pixel pre-pass), z padded to 4:")
for cap in (2048, 3072, 4096, 4608, 6144, 8192):
    ok = fr[:, 0] <= cap
    print(f"    capacity {cap:5d}: {ok.mean():.3f} of hit tiles fit, holding {fr[ok, 1].sum() / fr[:, 1].sum():.3f} of the hit pixels")
print(f"    volume median {np.median(fr[:,0]):.0f} p90 {np.percentile(fr[:,0],90):.0f}")
