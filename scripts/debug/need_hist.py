"""debug: distribution of the entries consumed per tile (tile_need) and list lengths on the headline workload"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import camera, synth
from diff_gaussian_rasterization import _native as N
import test_gpu_batch as TB
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
W, H = 1920, 1080
views = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)
P = g["means3D"].shape[0]
for v in (0, 1, 5):
    args = TB._batch_args(g, views[v:v + 1], W, H, dev)
    counts, color, radii, geom, binning, img = N.rasterize_gaussians_batch(*args, need_backward=True)
    need = N.query("TILE_NEED", P, W, H, counts[0], geom, binning, img).cpu().numpy().astype(np.int64)
    rng = N.query("RANGES", P, W, H, counts[0], geom, binning, img).cpu().numpy().astype(np.int64)
    total = rng[:, 1] - rng[:, 0]
    ne = total > 0
    print("view %d: %d non-empty tiles, sum need %d, sum total %d" % (v, ne.sum(), need.sum(), total.sum()))
    for thr in (256, 512, 1024, 1536, 2048, 3072, 4096, 5120):
        sel = need > thr
        print("   need > %4d: %4d tiles (%.1f%%), entries beyond: %d; their list lengths: median %d" % (
            thr, sel.sum(), 100.0 * sel.sum() / ne.sum(), (need[sel] - thr).sum(), np.median(total[sel]) if sel.any() else 0))
    full = (need >= total) & ne
    print("   tiles that walk their whole list: %d; need/total median over non-empty %.2f" % (full.sum(), np.median(need[ne] / total[ne])))
    o = np.argsort(-total)
    rank = np.empty_like(o); rank[o] = np.arange(len(o))
    deep = need > 2048
    print("   dispatch rank (by list length) of tiles with need > 2048: min %d median %d max %d of %d" % (
        rank[deep].min() if deep.any() else -1, np.median(rank[deep]) if deep.any() else -1, rank[deep].max() if deep.any() else -1, ne.sum()))
