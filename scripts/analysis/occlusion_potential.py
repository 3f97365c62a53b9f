"""How many (tile, Gaussian) pairs could a depth-slab emission skip?  CPU only (plain-C oracle).
For a bench view: per tile, list length L_t, consumed prefix n_t = max n_contrib over its pixels, and whether the tile
ended because every pixel terminated.  With K slabs of equal Gaussian count in depth order, a tile's pairs of the slabs
behind the one in which it saturates need never be emitted."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np
import util
from pcrender import camera, synth
from oracle.oracle import Oracle

W, H = 1920, 1080
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
views = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)
orc = Oracle()
for vi in [int(a) for a in sys.argv[1:]] or [1]:
    s = util.scene_from(g, views[vi], W, H, bg=(1, 1, 1))
    o = orc.forward(s, nthreads=8)
    R, P = o["R"], o["P"]
    ranges, ncon, fT = o["ranges"], o["n_contrib"], o["final_T"]
    keys, vals = o["keys"], o["vals"]
    gx, gy = o["gridx"], o["gridy"]
    # depth rank of every Gaussian = order of first appearance in the depth-sorted emission; use depth bits + id
    d = o["depths"]
    order = np.lexsort((np.arange(P), d.view(np.uint32)))
    rank = np.empty(P, np.int64); rank[order] = np.arange(P)
    vis = o["radii"] > 0
    nvis = int(vis.sum())
    # rank among visible only
    rv = np.cumsum(vis[order]) - 1
    rank_vis = np.empty(P, np.int64); rank_vis[order] = rv
    L = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
    # per-tile consumed prefix + saturation
    pad_h, pad_w = gy * 16, gx * 16
    nc = np.zeros((pad_h, pad_w), np.int64); nc[:H, :W] = ncon
    done = np.ones((pad_h, pad_w), bool); done[:H, :W] = fT < 1e-4   # pixels outside count as done
    # NB: final_T < 1e-4 never happens (the test is on test_T); a pixel is done when the NEXT test_T would be < 1e-4,
    # so use "n_contrib < L" as 'stopped before the list ended' instead
    nct = nc.reshape(gy, 16, gx, 16).max((1, 3)).reshape(-1)
    inside = np.zeros((pad_h, pad_w), bool); inside[:H, :W] = True
    stopped = (nc < np.repeat(np.repeat(L.reshape(gy, gx), 16, 0), 16, 1)) | ~inside
    sat = stopped.reshape(gy, 16, gx, 16).all((1, 3)).reshape(-1)
    print("view %d: P=%d visible=%d R=%d  non-empty tiles %d  consumed sum(n_t)=%d (%.1f%% of R)  saturated tiles %d holding %d pairs, unsaturated hold %d pairs"
          % (vi, P, nvis, R, int((L > 0).sum()), int(nct.sum()), 100.0 * nct.sum() / R, int((sat & (L > 0)).sum()), int(L[sat].sum()), int(L[~sat].sum())))
    for K in (1, 2, 4, 8, 16, 32, 64):
        emitted = 0
        slab_of = (rank_vis * K) // max(nvis, 1)
        for t in np.nonzero(L > 0)[0]:
            a, b = ranges[t]
            if not sat[t]:
                emitted += b - a
                continue
            sl = slab_of[vals[a:b]]
            last = sl[nct[t] - 1] if nct[t] > 0 else -1
            # the tile is found saturated at the end of slab `last` (its render consumed entry n_t-1 there and stopped)
            emitted += int(np.searchsorted(sl, last, side="right"))
        print("   K=%2d slabs: emitted %9d pairs = %.1f%% of R" % (K, emitted, 100.0 * emitted / R))
