"""Forward march: wave-iterations under different lane-assignment schemes (CPU, oracle step counts).

  static    one ray per lane of an 8x8 patch, a wave runs until its longest ray ends (what ships)
  refill    a workgroup's rays (TW x TH tile) go to one queue; a wave takes 64, and whenever >= K of its
            lanes are idle and the queue is not empty the idle lanes take new rays
  twophase  static for the first N iterations (or until < K lanes are left), survivors are pooled per
            workgroup and marched again in dense waves
Prints wave-iterations per view (the VALU cost of the march is proportional to it)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle
from sdfest_amd.synthetic import blobs_sdf, random_poses

W, H, f, thr = 640, 480, 320.0, 0.005
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sdf = blobs_sdf(0)
pos, quat, isc = random_poses(256, seed=1)
oracle.set_threads(8)
depth, steps, margin = oracle.render_forward(sdf, pos[:B], quat[:B], isc[:B], W, H, W / 2, H / 2, f, f, thr, with_aux=True)
steps = steps.astype(np.int64)
total = steps.sum()
print(f"{B} views: {total/B:.0f} lane-steps/view, ideal {total/B/64:.0f} wave-iterations/view")

# static 8x8
p = steps.reshape(B, H // 8, 8, W // 8, 8).transpose(0, 1, 3, 2, 4).reshape(B, -1, 64)
static = p.max(axis=2).sum()
print(f"static 8x8: {static/B:.0f} wave-iterations/view, utilisation {total/64/static:.3f}")


def tile_rays(TW, TH):
    """per workgroup tile: ray step counts in patch order (8x8 patches row-major inside the tile)"""
    t = steps.reshape(B, H // TH, TH // 8, 8, W // TW, TW // 8, 8).transpose(0, 1, 4, 2, 5, 3, 6)
    return t.reshape(-1, (TW * TH) // 64, 64)


def sim_refill(rays, K, nwaves=4):
    q = rays[rays > 0]
    if len(q) == 0:
        return 0
    head = 0
    lanes = [np.zeros(64, dtype=np.int64) for _ in range(nwaves)]
    iters = 0
    alive = [True] * nwaves
    while True:
        progressed = False
        for w in range(nwaves):
            l = lanes[w]
            idle = l == 0
            n_idle = int(idle.sum())
            if head < len(q) and (n_idle >= K):
                take = min(n_idle, len(q) - head)
                idx = np.nonzero(idle)[0][:take]
                l[idx] = q[head:head + take]
                head += take
            if l.any():
                iters += 1
                l[l > 0] -= 1
                progressed = True
        if not progressed:
            break
    return iters


def sim_twophase(tile, N, K, nwaves=4):
    """tile: (npatch, 64) step counts; phase 1 static up to N iterations or until < K lanes alive"""
    iters = 0
    surv = []
    for prow in tile:
        s = np.sort(prow[prow > 0])[::-1]
        if len(s) == 0:
            continue
        # iteration i (1-based) has #alive = count(s >= i); stop when i > N or alive < K
        mx = s[0]
        stop = mx
        for i in range(1, mx + 1):
            alive = int((s >= i).sum())
            if i > N or alive < K:
                stop = i - 1
                break
        iters += stop
        rest = s[s > stop] - stop
        surv.extend(rest.tolist())
    if surv:
        surv = np.array(surv)
        # dense waves of 64, static inside (sorted would be optimistic: keep arrival order)
        for i in range(0, len(surv), 64):
            iters += surv[i:i + 64].max()
    return iters


for (TW, TH) in ((64, 8), (64, 32)):
    tiles = tile_rays(TW, TH)
    for K in (8, 16, 32):
        it = sum(sim_refill(t.reshape(-1), K) for t in tiles)
        print(f"refill tile {TW}x{TH} K={K}: {it/B:.0f} wave-iterations/view, utilisation {total/64/it:.3f}")
    for (N, K) in ((6, 16), (8, 16), (8, 24), (12, 16)):
        it = sum(sim_twophase(t, N, K) for t in tiles)
        print(f"twophase tile {TW}x{TH} N={N} K={K}: {it/B:.0f} wave-iterations/view, utilisation {total/64/it:.3f}")
