"""What fraction of the emitted (tile, Gaussian) pairs can never reach alpha >= 1/255 at any pixel of the tile?
(The reference emits the bounding square of 3 sigma_max; anisotropic splats and low opacities leave dead tiles.)
CPU only: oracle forward of one bench view, exact per-pixel evaluation on a random sample of Gaussians; also split by
consumed / not consumed list entries."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np
import util
from pcrender import camera, synth
from oracle.oracle import Oracle

name = sys.argv[1] if len(sys.argv) > 1 else "synth-THuman-800K"
W, H = (3840, 2160) if "2M" in name else (1920, 1080)
cloud = synth.make_cloud(name, seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
views = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)
s = util.scene_from(g, views[1], W, H, bg=(1, 1, 1))
o = Oracle().forward(s, nthreads=8)
P, R = o["P"], o["R"]
m2, co, radii = o["means2D"], o["conic_opacity"], o["radii"]
gx, gy = o["gridx"], o["gridy"]
rng = np.random.default_rng(0)
vis = np.nonzero(radii > 0)[0]
sample = rng.choice(vis, size=min(20000, vis.size), replace=False)
tot = dead = 0
hist_area = {}
for i in sample:
    px, py, r = m2[i, 0], m2[i, 1], float(radii[i])
    x0 = int(min(gx, max(0, int((px - r) / 16)))); y0 = int(min(gy, max(0, int((py - r) / 16))))
    x1 = int(min(gx, max(0, int((px + r + 15) / 16)))); y1 = int(min(gy, max(0, int((py + r + 15) / 16))))
    if x1 <= x0 or y1 <= y0:
        continue
    A, B, C, op = co[i]
    xs = np.arange(x0 * 16, x1 * 16, dtype=np.float32); ys = np.arange(y0 * 16, y1 * 16, dtype=np.float32)
    dx = px - xs[None, :]; dy = py - ys[:, None]
    power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
    alpha = np.minimum(0.99, op * np.exp(power))
    hit = (power <= 0) & (alpha >= 1.0 / 255.0)
    ht = hit.reshape(y1 - y0, 16, x1 - x0, 16).any((1, 3))
    n = ht.size
    tot += n; dead += n - int(ht.sum())
print("%s view 1: P=%d R=%d (%.1f tiles per Gaussian); sample of %d Gaussians: %d pairs, %d dead at tile level = %.1f%%"
      % (name, P, R, R / max(1, vis.size), sample.size, tot, dead, 100.0 * dead / tot))
