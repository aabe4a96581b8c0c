"""How much of the forward march would a tight may-hit bounding box prune?  (CPU, oracle step counts)

For each view the rays that pass the full-cube slab test are split into those that also cross the
AABB (object frame) of all cells having a corner value < v_max = threshold * t_far_max / scale, and
those that do not (certain misses: every sample of such a ray lies in a cell whose 8 corners are
>= v_max, so dist >= threshold * t everywhere).  Prints the share of march steps / rays pruned."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle
from sdfest_amd.synthetic import blobs_sdf, random_poses

W, H, f, thr = 640, 480, 320.0, 0.005
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sdf = blobs_sdf(0)
pos, quat, isc = random_poses(256, seed=1)
pos, quat, isc = pos[:B], quat[:B], isc[:B]
oracle.set_threads(8)
depth, steps, margin = oracle.render_forward(sdf, pos, quat, isc, W, H, W / 2, H / 2, f, f, thr, with_aux=True)

R = 64
cellmin = np.minimum.reduce([sdf[dx:R - 1 + dx, dy:R - 1 + dy, dz:R - 1 + dz] for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)])

def rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

tot_steps = tot_rays = kept_steps = kept_rays = 0
hits = 0
rect_full = rect_tight = 0
for b in range(B):
    scale = 1.0 / isc[b]
    Rm = rot(quat[b].astype(np.float64))
    tfar_max = np.linalg.norm(pos[b]) + np.sqrt(3) * scale
    vmax = thr * tfar_max / scale
    occ = np.argwhere(cellmin < vmax)
    lo = occ.min(0) - 1          # one cell of margin
    hi = occ.max(0) + 2
    lo = np.clip(lo, 0, R - 1); hi = np.clip(hi, 0, R - 1)
    # object-frame box (units of the object frame: grid coordinate g -> (g / ((R-1)/2) - 1) * scale)
    blo = (lo / ((R - 1) / 2) - 1) * scale
    bhi = (hi / ((R - 1) / 2) - 1) * scale
    # rays
    cols, rows = np.meshgrid(np.arange(W), np.arange(H))
    dx = (cols + 0.5 - W / 2) / f; dy = -(rows + 0.5 - H / 2) / f
    d = np.stack([dx, dy, -np.ones_like(dx)], -1); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    dobj = d @ Rm            # R^T d
    e = Rm.T @ pos[b].astype(np.float64)     # object centre seen from the origin, object frame
    # origin in object frame = -e ; box [blo, bhi]
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (blo + e) / dobj; t2 = (bhi + e) / dobj
    tn = np.minimum(t1, t2).max(-1); tf = np.maximum(t1, t2).min(-1)
    cross = (tn <= tf) & (tf >= 0)
    s = steps[b]
    inbox = s > 0
    tot_steps += s.sum(); tot_rays += inbox.sum()
    kept_steps += s[cross].sum(); kept_rays += (inbox & cross).sum()
    hits += (depth[b] > 0).sum()
    assert not np.any((depth[b] > 0) & ~cross), "tight box culled a hit!"
    ys, xs = np.nonzero(inbox); rect_full += (xs.max() - xs.min() + 1) * (ys.max() - ys.min() + 1)
    ys, xs = np.nonzero(inbox & cross); rect_tight += (xs.max() - xs.min() + 1) * (ys.max() - ys.min() + 1)
    if b < 4:
        print(f"view {b}: vmax {vmax:.4f} box lo {lo} hi {hi}")
print(f"{B} views: rays in cube {tot_rays/B:.0f}/view, steps {tot_steps/B:.0f}/view, hits {hits/B:.0f}/view")
print(f"tight box keeps {kept_rays/tot_rays:.3f} of the rays, {kept_steps/tot_steps:.3f} of the steps")
print(f"bounding rect area: full {rect_full/B:.0f} px, tight {rect_tight/B:.0f} px ({rect_tight/rect_full:.3f})")
