"""Lane utilisation of the batch backward on the benchmark poses (CPU, oracle depth): active 16x4 wave passes of
the shipped tiling (32x32-pixel tiles from 2 pixels per voxel, 64x8 below) against per-wave and per-workgroup
compaction of the hit pixels.  python tools/analysis/backward_utilisation.py [views]"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from sdfest_amd.synthetic import blobs_sdf, random_poses
W,H,f=640,480,320.0
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
sdf=blobs_sdf(0)
pos,quat,isc=random_poses(256,seed=1,width=W,height=H,f=f)
pos,quat,isc=pos[:B],quat[:B],isc[:B]
d=oracle.render_forward(sdf,pos,quat,isc,W,H,W/2,H/2,f,f,0.005,dtype=np.float32)
hit=(d>0)
tot_hits=hit.sum()
act=0; per_wave=0; per_wg=0; tiles_hit=0
for b in range(B):
    scale=1/isc[b]; dist=np.linalg.norm(pos[b]); r=f*(scale/31.5)/dist
    big = r>=2.0
    hb=hit[b]
    # 16x4 patches
    p=hb.reshape(H//4,4,W//16,16).sum(axis=(1,3))   # (120,40) patch hit counts
    act+= (p>0).sum()
    if big:
        th,tw=32,32   # tile: 4 subs of 32x8, wave w owns patches (ox=(w%2)*16, oy=(w//2)*4) in each sub
        t=p.reshape(H//32,8,W//32,2)      # rows of patches within tile: 8 patch rows (4 subs x 2), 2 patch cols
        # wave = (prow%2)*2 + pcol ; sub = prow//2
        tw_=t.reshape(H//32,4,2,W//32,2)  # (ty, sub, wrow, tx, wcol)
        wave_hits=tw_.sum(axis=1)         # (ty, wrow, tx, wcol)
        per_wave+=np.ceil(wave_hits/64).sum()
        tile_hits=wave_hits.sum(axis=(1,3))
        per_wg+= (np.ceil(tile_hits/256)*4).sum()
        tiles_hit+=(tile_hits>0).sum()
    else:
        # 64x8 tile: 2 subs (SX=2) of 32x8
        t=p.reshape(H//8,2,W//64,2,2)     # (ty, wrow, tx, sub, wcol)
        wave_hits=t.sum(axis=3)
        per_wave+=np.ceil(wave_hits/64).sum()
        tile_hits=wave_hits.sum(axis=(1,3))
        per_wg+=(np.ceil(tile_hits/256)*4).sum()
        tiles_hit+=(tile_hits>0).sum()
print(f"views {B} hits {tot_hits} hit tiles {tiles_hit}")
print(f"active wave passes now {act}  utilisation {tot_hits/(64*act):.3f}")
print(f"per-wave compaction   {int(per_wave)}  ({per_wave/act:.3f} of now)")
print(f"per-WG compaction     {int(per_wg)}  ({per_wg/act:.3f} of now)")
