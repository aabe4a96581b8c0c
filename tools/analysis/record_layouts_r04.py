"""Forward, round 4: distinct 128-byte LINES of the face-record array per gather instruction of a wave (the TCP's tag
look-ups: DESIGN 3.2 measures 23.9 per instruction), both loads of a march step, for record layouts:
  shipped   [x | y>>1 | z>>1 | y&1 | z&1]            a line = 2 (y) x 4 (z) records of one x
  cube      [x>>1 | y>>1 | z>>1 | x&1 | y&1 | z&1]   a line = 2 x 2 x 2 records
  xz        [y | x>>1 | z>>2 | x&1 | z&3]            a line = 2 (x) x 4 (z) records of one y
  z8        [x | y | z]                              a line = 8 consecutive z
Re-marches a few benchmark views in numpy (the kernels' sample sequence).  CPU only."""
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
    return c0 * (1 - o[:, 2]) + c1 * o[:, 2], b

def line(layout, x, y, z):
    if layout == "shipped": return ((x * 32 + (y >> 1)) * 32 + (z >> 1)) >> 1
    if layout == "cube":    return ((x >> 1) * 32 + (y >> 1)) * 32 + (z >> 1)
    if layout == "xz":      return (y * 32 + (x >> 1)) * 16 + (z >> 2)
    if layout == "z8":      return (x * 64 + y) * 8 + (z >> 3)
layouts = ["shipped", "cube", "xz", "z8"]
tot = {k: [0, 0, 0] for k in layouts}    # lines of load a, lines of load b, instructions
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
    it = 0
    while act.any() and it < 60:
        idx = np.nonzero(act)[0]
        v, bb = trilerp(og + t[idx, None] * dg[idx])
        for name in layouts:
            for k, dxx in enumerate((0, 1)):
                ch = np.full(W * H, -1, np.int64); ch[idx] = line(name, bb[:, 0] + dxx, bb[:, 1], bb[:, 2])
                blk = ch.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
                live = (blk >= 0).any(axis=1)
                s = np.sort(blk[live], axis=1)
                distinct = ((s[:, 1:] != s[:, :-1]) & (s[:, 1:] >= 0)).sum(axis=1) + (s[:, 0] >= 0)
                tot[name][k] += distinct.sum()
                if k == 0: tot[name][2] += live.sum()
        dist = v * scale
        hit = dist < thr * t[idx]
        tnew = t[idx] + dist
        stop = hit | ~(tnew < tf[idx])
        t[idx] = np.where(stop, t[idx], tnew)
        act[idx[stop]] = False
        it += 1
print(f"{B} views; distinct 128-byte lines per gather instruction of an 8x8-pixel wave (load x / load x+1 / both):")
for name, (a, c, n) in tot.items():
    print(f"  {name:8s} {a / n:6.2f} {c / n:6.2f}   sum {(a + c) / n:6.2f}")
