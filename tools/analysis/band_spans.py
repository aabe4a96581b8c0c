"""Forward, round 3: how many 8x8 patches / 64x8 tiles of a view's may-hit RECTANGLE lie outside the projected may-hit
BOX itself?  Compares the bounding rectangle of the 8 projected corners (what the kernels cull against since round 2)
with per-band column spans of the projected box's outline (band = 8 image rows), both with a 2 pixel margin.  CPU."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from sdfest_amd.synthetic import blobs_sdf, random_poses

W, H, f, thr, R = 640, 480, 320.0, 0.005, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "c3"
sdf = blobs_sdf(0)
pos, quat, isc = random_poses(256, seed=1)
pos, quat, isc = pos[:B].astype(np.float64), quat[:B].astype(np.float64), isc[:B].astype(np.float64)
if mode == "mug":
    pos *= 0.3
    isc[:] = 1 / 0.055
pm = [sdf.min(axis=tuple(a for a in range(3) if a != ax)) for ax in range(3)]   # plane minima per axis

def rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

EDGES = [(a, a ^ (1 << k)) for a in range(8) for k in range(3) if not a & (1 << k)]
tot = dict(rect_p=0, span_p=0, exact_p=0, rect_t=0, span_t=0, exact_t=0)
cols, rows = np.meshgrid(np.arange(W), np.arange(H))
for b in range(B):
    scale = 1.0 / isc[b]; h = (R - 1) / 2
    Rm = rot(quat[b])
    vhit = thr * (np.linalg.norm(pos[b]) + 1.7321 * scale) * isc[b] * 1.0001
    lo, hi = np.empty(3), np.empty(3)
    for a in range(3):
        ok = np.minimum(pm[a][:-1], pm[a][1:]) < vhit
        first, last = np.argmax(ok), len(ok) - 1 - np.argmax(ok[::-1])
        cell = scale / h
        lo[a] = max(-scale, (first - 0.0625) * cell - scale); hi[a] = min(scale, (last + 1 + 0.0625) * cell - scale)
    corners = np.array([[hi[k] if c >> k & 1 else lo[k] for k in range(3)] for c in range(8)])
    P = pos[b] + corners @ Rm.T
    u = W / 2 + f * P[:, 0] / -P[:, 2]; v = H / 2 - f * P[:, 1] / -P[:, 2]     # pixel coordinates (centre-0.5 convention)
    m = 2.0 + 1e-5 * f
    x0 = int(np.clip(np.floor(u.min() - 0.5 - m), 0, W)); x1 = int(np.clip(np.ceil(u.max() - 0.5 + m) + 1, 0, W))
    y0 = int(np.clip(np.floor(v.min() - 0.5 - m), 0, H)); y1 = int(np.clip(np.ceil(v.max() - 0.5 + m) + 1, 0, H))
    # exact: rays crossing the box
    dx = (cols + 0.5 - W / 2) / f; dy = -(rows + 0.5 - H / 2) / f
    d = np.stack([dx, dy, -np.ones_like(dx)], -1)
    dobj = d @ Rm; e = Rm.T @ pos[b]
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (lo + e) / dobj; t2 = (hi + e) / dobj
    cross = (np.minimum(t1, t2).max(-1) <= np.maximum(t1, t2).min(-1)) & (np.maximum(t1, t2).min(-1) >= 0)
    # band spans: extent of the 12 projected box edges over rows [8k - m, 8k + 8 + m), +- m pixels
    nb = H // 8
    s0 = np.full(nb, W, int); s1 = np.zeros(nb, int)
    for k in range(nb):
        ya, yb = 8 * k - 0.5 - m, 8 * k + 8 - 0.5 + m     # pixel centres of rows 8k .. 8k+7 are at row + 0.5 in (u, v) terms
        ya += 0.5; yb += 0.5
        xs = []
        for (i, j) in EDGES:
            va, vb = v[i], v[j]
            for yy in (ya, yb):
                if (va - yy) * (vb - yy) <= 0 and va != vb:
                    tt = (yy - va) / (vb - va); xs.append(u[i] + tt * (u[j] - u[i]))
        xs += [u[i] for i in range(8) if ya <= v[i] <= yb]
        if xs:
            s0[k] = int(np.clip(np.floor(min(xs) - 0.5 - m), 0, W)); s1[k] = int(np.clip(np.ceil(max(xs) - 0.5 + m) + 1, 0, W))
    for (pw, key) in ((8, "p"), (64, "t")):
        for k in range(nb):
            for px in range(0, W, pw):
                in_rect = px < x1 and px + pw > x0 and 8 * k < y1 and 8 * k + 8 > y0
                in_span = in_rect and px < s1[k] and px + pw > s0[k]
                ex = cross[8 * k:8 * k + 8, px:px + pw].any()
                assert not (ex and not in_span), ("span culled a crossing ray", b, k, px)
                tot["rect_" + key] += in_rect; tot["span_" + key] += in_span; tot["exact_" + key] += ex
print(f"{mode}, {B} views of {W}x{H}")
for key, name in (("p", "8x8 patches"), ("t", "64x8 tiles")):
    r, s, e = tot["rect_" + key], tot["span_" + key], tot["exact_" + key]
    print(f"  {name}: in the rectangle {r / B:.0f} per view, in the band spans {s / B:.0f} ({s / r:.3f}), "
          f"holding a ray that crosses the box {e / B:.0f} ({e / r:.3f})")
