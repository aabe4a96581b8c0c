"""Backward work distribution on the benchmark scene (CPU, oracle depth): how many 64x8 tiles a view's
rectangle holds, how many of them contain a hit pixel, hit pixels per such tile, and how full the
16x4-pixel wave patches are.  Decides what a compaction of hit pixels could save."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle
from sdfest_amd.synthetic import blobs_sdf, random_poses

W, H, f, thr = 640, 480, 320.0, 0.005
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sdf = blobs_sdf(0)
pos, quat, isc = random_poses(256, seed=1)
pos, quat, isc = pos[:B], quat[:B], isc[:B]
oracle.set_threads(8)
depth, steps, margin = oracle.render_forward(sdf, pos, quat, isc, W, H, W / 2, H / 2, f, f, thr, with_aux=True)
hit = depth > 0
incube = steps > 0
for (tw, th) in ((64, 8), (32, 8), (64, 32), (128, 32)):
    ht = hit.reshape(B, H // th, th, W // tw, tw).sum(axis=(2, 4))
    ct = incube.reshape(B, H // th, th, W // tw, tw).sum(axis=(2, 4))
    ntiles = ht.size
    # rectangle tiles: bounding rectangle of in-cube pixels per view (what the cull keeps)
    rect_tiles = 0
    for b in range(B):
        ys, xs = np.nonzero(ct[b] > 0)
        if len(ys):
            rect_tiles += (ys.max() - ys.min() + 1) * (xs.max() - xs.min() + 1)
    live = ht > 0
    print(f"tile {tw}x{th}: {ntiles/B:.0f} tiles/view, in rect {rect_tiles/B:.0f}, with a hit {live.sum()/B:.1f}, "
          f"hits per live tile mean {ht[live].mean():.1f} median {np.median(ht[live]):.0f} p90 {np.percentile(ht[live],90):.0f} "
          f"max {ht.max()} (of {tw*th}); fill {ht[live].mean()/(tw*th):.3f}")
# wave patches 16x4 inside live 64x8 tiles
hp = hit.reshape(B, H // 4, 4, W // 16, 16).sum(axis=(2, 4))
print(f"16x4 patches with a hit: {(hp>0).sum()/B:.0f}/view, mean fill of those {hp[hp>0].mean()/64:.3f}")
print(f"hits/view {hit.sum()/B:.0f}; in-cube px/view {incube.sum()/B:.0f}")
