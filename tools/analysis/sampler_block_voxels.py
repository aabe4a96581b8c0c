"""How many global atomics the sampler's backward needs per point, by the shape of the patch of pixels a workgroup
pre-sums in LDS (unique voxels touched by the block's points / points): the oracle's depth images of C3 poses,
back-projected, every point's 8 voxels, grouped by patch.  CPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle

def main():
    V, W, H, f = 12, 640, 480, 320.0
    sdf = oracle.blobs_sdf(0)
    pos, quat, isc = oracle.random_poses(V, seed=1, width=W, height=H, f=f)
    depth = oracle.render_forward(sdf, pos, quat, isc, W, H, W / 2, H / 2, f, f, 0.005, dtype=np.float32)
    shapes = {"row-major 256": None, "16x16": (16, 16), "32x16 (512 pts)": (16, 32), "32x32 (1024 pts)": (32, 32),
              "64x16 (1024 pts)": (16, 64), "64x32 (2048 pts)": (32, 64)}
    tot = {k: [0, 0, 0] for k in shapes}
    for v in range(V):
        rows, cols = np.nonzero(depth[v])
        if len(rows) == 0: continue
        z = depth[v][rows, cols].astype(np.float64)
        P = np.stack(((cols - (W / 2 - 0.5)) * z / f, -(rows - (H / 2 - 0.5)) * z / f, -z), 1)
        # object frame (pc_loss: R(conj q)(P - p) / scale), grid cell
        x, y, zq, w = quat[v]
        R = np.array([[1 - 2 * (y * y + zq * zq), 2 * (x * y - w * zq), 2 * (x * zq + w * y)],
                      [2 * (x * y + w * zq), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - w * x)],
                      [2 * (x * zq - w * y), 2 * (y * zq + w * x), 1 - 2 * (x * x + y * y)]])
        o = (P - pos[v]) @ R            # R^T (P - p)
        g = (o * isc[v] + 1) * 31.5
        c = np.clip(np.floor(g), 0, 62).astype(np.int64)
        lin = (c[:, 0] * 64 + c[:, 1]) * 64 + c[:, 2]
        corners = (lin[:, None] + np.array([0, 1, 64, 65, 4096, 4097, 4160, 4161])[None]).ravel()
        for name, shp in shapes.items():
            if shp is None:
                blk = np.arange(len(rows)) // 256
            else:
                th, tw = shp
                key = (rows // th) * 1000 + cols // tw
                # consecutive points of the patch order: 256 x reps per block, patches are dense only inside the object
                order = np.argsort(key * 10**7 + (rows % th) * 10**3 + cols % tw, kind="stable")
                inv = np.empty_like(order); inv[order] = np.arange(len(order))
                blk = inv // (th * tw)
            pair = np.unique(np.repeat(blk, 8) * (1 << 20) + corners)
            # z-pair runs (one LDS slot per 4 consecutive z): table pressure
            runs = np.unique(np.repeat(blk, 8) * (1 << 20) + corners // 4)
            nb = blk.max() + 1
            tot[name][0] += len(pair); tot[name][1] += len(rows); tot[name][2] = max(tot[name][2], np.bincount(runs >> 20).max())
    for name, (a, n, r) in tot.items():
        print(f"{name:20s} {a / n:.3f} atomics per point, largest block touches {r} 4-voxel runs")
main()
