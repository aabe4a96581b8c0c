"""Forward, round 3: how often does a ray's march step stay in the cell of its previous step (the 8 corners could be
kept in registers and the step's two 16-byte gathers skipped for that lane)?  Re-marches a few benchmark views in
numpy (same sample sequence as the kernels: full cube's near plane, no may-hit box) and counts lane-steps, per wave
(8 x 8 patch) the share of gather instructions that would have NO lane left, and the lanes per issued gather.
CPU only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from sdfest_amd.synthetic import blobs_sdf, random_poses

W, H, f, thr, R = 640, 480, 320.0, 0.005, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sdf = blobs_sdf(0).astype(np.float64)
pos, quat, isc = random_poses(256, seed=1)

def rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

def trilerp(g):
    b = np.clip(np.floor(g), 0, R - 2).astype(int)
    o = g - b
    x, y, z = b[:, 0], b[:, 1], b[:, 2]
    c = lambda dx, dy, dz: sdf[x + dx, y + dy, z + dz]
    c00 = c(0, 0, 0) * (1 - o[:, 0]) + c(1, 0, 0) * o[:, 0]; c01 = c(0, 0, 1) * (1 - o[:, 0]) + c(1, 0, 1) * o[:, 0]
    c10 = c(0, 1, 0) * (1 - o[:, 0]) + c(1, 1, 0) * o[:, 0]; c11 = c(0, 1, 1) * (1 - o[:, 0]) + c(1, 1, 1) * o[:, 0]
    c0 = c00 * (1 - o[:, 1]) + c10 * o[:, 1]; c1 = c01 * (1 - o[:, 1]) + c11 * o[:, 1]
    return c0 * (1 - o[:, 2]) + c1 * o[:, 2], (x * R + y) * R + z

lane_steps = same_cell = 0
wave_steps = wave_steps_no_load = 0
lanes_per_issued = []
cols, rows = np.meshgrid(np.arange(W), np.arange(H))
for b in range(B):
    scale = 1.0 / isc[b]; h = (R - 1) / 2
    Rm = rot(quat[b].astype(np.float64)); e = Rm.T @ pos[b].astype(np.float64)
    dx = (cols + 0.5 - W / 2) / f; dy = -(rows + 0.5 - H / 2) / f
    d = np.stack([dx, dy, -np.ones_like(dx)], -1); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    dobj = (d @ Rm).reshape(-1, 3)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (e + scale) / dobj; t2 = (e - scale) / dobj
    tn = np.maximum(np.minimum(t1, t2).max(-1), 0); tf = np.maximum(t1, t2).min(-1)
    act = (tn < tf) & (tf >= 0)
    t = tn.copy()
    og = (-e * isc[b] + 1) * h; dg = dobj * isc[b] * h
    prev = np.full(W * H, -1, np.int64)
    it = 0
    while act.any() and it < 64:
        idx = np.nonzero(act)[0]
        v, cell = trilerp(og + t[idx, None] * dg[idx])
        same = cell == prev[idx]
        lane_steps += len(idx); same_cell += same.sum()
        need = np.zeros(W * H, bool); need[idx] = ~same
        live = np.zeros(W * H, bool); live[idx] = True
        pw = lambda a: a.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        nw, lw = pw(need).sum(1), pw(live).any(1)
        wave_steps += lw.sum(); wave_steps_no_load += (lw & (nw == 0)).sum()
        lanes_per_issued.append(nw[lw & (nw > 0)])
        prev[idx] = cell
        dist = v * scale
        hit = dist < thr * t[idx]
        tnew = t[idx] + dist
        stop = hit | ~(tnew < tf[idx])
        t[idx] = np.where(stop, t[idx], tnew)
        act[idx[stop]] = False
        it += 1
lp = np.concatenate(lanes_per_issued)
print(f"{B} views: lane-steps {lane_steps}, in the previous step's cell {same_cell} ({100 * same_cell / lane_steps:.1f} %)")
print(f"wave-steps {wave_steps}, with no lane needing a load {wave_steps_no_load} ({100 * wave_steps_no_load / wave_steps:.1f} %)")
print(f"lanes loading per issued gather: mean {lp.mean():.1f} of 64 (all lanes of live wave-steps: {lane_steps / wave_steps:.1f})")
