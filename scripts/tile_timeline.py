"""Per-tile timeline of the render kernels on the headline workload (debug instrumentation: GSR_Q_TILE_CLOCK)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import synth, camera
from oracle.oracle import Scene
import util
from diff_gaussian_rasterization import _native as N
view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W, H = 1920, 1080
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
v = camera.circle_views(12, fov_deg=45., width_px=W, height_px=H)[view]
def t(a): return torch.empty(0) if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
args = (torch.ones(3, device=dev), t(g["means3D"]), t(None), t(g["opacities"]), t(g["scales"]), t(g["rotations"]), 1.0, t(None),
        v["viewmatrix"].to(dev), v["projmatrix"].to(dev), v["tanfovx"], v["tanfovy"], H, W, t(g["shs"]), 1, v["campos"].to(dev), False, False)
dL = t(np.random.default_rng(123).uniform(-1, 1, (3, H, W)).astype(np.float32))
for it in range(3):
    R, color, radii, geom, binning, img = N.rasterize_gaussians(*args, need_backward=True)
    N.rasterize_gaussians_backward(args[0], args[1], radii, args[2], args[4], args[5], 1.0, args[7], args[8], args[9],
                                   v["tanfovx"], v["tanfovy"], dL, args[14], 1, args[16], geom, R, binning, img, False)
torch.cuda.synchronize()
P = g["means3D"].shape[0]
clk = N.query("TILE_CLOCK", P, W, H, R, geom, binning, img).cpu().numpy().astype(np.int64)
need = N.query("TILE_NEED", P, W, H, R, geom, binning, img).cpu().numpy()
rng = N.query("RANGES", P, W, H, R, geom, binning, img).cpu().numpy()
ln = rng[:, 1] - rng[:, 0]
for name, a, b in (("forward", 0, 1), ("backward", 2, 3)):
    t0, t1 = clk[:, a], clk[:, b]
    start = t0.min()
    dur = (t1 - t0) * 10e-3      # us (100 MHz clock)
    end = (t1 - start) * 10e-3
    beg = (t0 - start) * 10e-3
    ne = ln > 0
    order = np.argsort(-end)
    print("== %s: kernel span %.1f us; non-empty tiles %d" % (name, end.max(), ne.sum()))
    print("   last 8 tiles to finish: " + ", ".join("tile%d need=%d len=%d start=%.0f dur=%.0f" % (i, need[i], ln[i], beg[i], dur[i]) for i in order[:8]))
    x = need[ne].astype(np.float64); y = dur[ne]
    A = np.stack([x, np.ones_like(x)], 1); coef = np.linalg.lstsq(A, y, rcond=None)[0]
    print("   dur ~= %.4f us/entry * need + %.1f us   (corr %.3f); ns/entry at the longest tiles: %.1f" % (coef[0], coef[1], np.corrcoef(x, y)[0, 1], 1e3 * (y[np.argsort(-x)[:20]] / x[np.argsort(-x)[:20]]).mean()))
    print("   start times: p50 %.0f us, p99 %.0f us, max %.0f us (non-empty)" % (np.percentile(beg[ne], 50), np.percentile(beg[ne], 99), beg[ne].max()))
    busy = dur[ne].sum()
    print("   sum of tile durations %.0f us over %d tiles -> avg concurrency %.1f tiles" % (busy, ne.sum(), busy / end.max()))
