"""Forward, round 3: how many distinct 64-byte chunks of the face-record array does one gather instruction of a wave
touch, per 16-lane group (a 128-bit load is processed a quarter wave at a time), for different lane -> pixel maps of
the 8 x 8 patch?  Re-marches a few benchmark views in numpy (same sample sequence as the kernels) and counts, per
wave and march iteration, the chunks of the first of the step's two loads (record of cell corner 000).  CPU only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from sdfest_amd.synthetic import blobs_sdf, random_poses

W, H, f, thr, R = 640, 480, 320.0, 0.005, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16      # lanes per coalescing group
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

def chunk_of(b):   # blocked record order: [x | y>>1 | z>>1 | y&1 | z&1] -> chunk = index >> 2
    return (b[:, 0] * 32 + (b[:, 1] >> 1)) * 32 + (b[:, 2] >> 1)

maps = {}
l = np.arange(64)
maps["row-major 8x8 (shipped)"] = (l % 8, l // 8)
maps["4x4 block per 16 lanes"] = ((l & 3) + 4 * ((l >> 4) & 1), ((l >> 2) & 3) + 4 * (l >> 5))
maps["8x2 strip per 16 lanes = shipped"] = maps["row-major 8x8 (shipped)"]
maps["2x8 strip per 16 lanes"] = ((l & 1) + 2 * (l >> 4), (l >> 1) & 7)
maps["2x2 pixel block per lane quad (Morton)"] = ((l & 1) + 2 * ((l >> 2) & 3), ((l >> 1) & 1) + 2 * (l >> 4))
maps["16x4 patch, 16x1 rows per 16 lanes"] = None   # handled separately

tot = {k: [0, 0] for k in maps}     # [chunks, instructions]
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
        ch = np.full(W * H, -1, np.int64); ch[idx] = chunk_of(bb)
        chim = ch.reshape(H, W)
        for name, m in maps.items():
            if m is None:   # 16x4 patches, lanes row-major: a 16-lane group = one row of 16 pixels
                blk = chim.reshape(H // 4, 4, W // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 64 // G, G)
            else:
                mx, my = m
                blk = chim.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3)[:, :, my, mx].reshape(-1, 64 // G, G)
            live = (blk >= 0).any(axis=(1, 2))
            blk = blk[live]
            s = np.sort(blk, axis=2)
            distinct = ((s[:, :, 1:] != s[:, :, :-1]) & (s[:, :, 1:] >= 0)).sum(axis=2) + (s[:, :, 0] >= 0)
            tot[name][0] += distinct.sum(); tot[name][1] += live.sum()
        dist = v * scale
        hit = dist < thr * t[idx]
        tnew = t[idx] + dist
        stop = hit | ~(tnew < tf[idx])
        t[idx] = np.where(stop, t[idx], tnew)
        act[idx[stop]] = False
        it += 1
print(f"{B} views; chunks per gather instruction (sum over its groups of {G} lanes), first load of a step:")
for name, (c, n) in tot.items():
    print(f"  {name:40s} {c / n:6.2f}   ({n / B:.0f} wave-iterations per view)")
