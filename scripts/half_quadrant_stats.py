"""How much would a finer cull granularity save?  For the consumed (quadrant, entry) evaluations of the benchmark view: the
fraction whose alpha >= 1/255 region misses the top / bottom 8x4 half (or the left / right 4x8 half) of the 8x8 quadrant, and
the lock-step step count max(s_half_a, s_half_b) relative to the 8x8 survivor count.  numpy on the arrays of one forward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
import util
from pcrender import camera, synth
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
v = camera.circle_views(12, fov_deg=45.0, width_px=1920, height_px=1080)[1]
s = util.scene_from(g, v, 1920, 1080, bg=(1, 1, 1))
p, _ = util.run_product(s, dev)
W, H = 1920, 1080
gx = (W + 15) // 16
vals, ranges, ncon = p["vals"], p["ranges"].reshape(-1, 2), p["n_contrib"]
m2, co = p["means2D"], p["conic_opacity"]
rng = np.random.default_rng(0)
tiles = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
tiles = rng.choice(tiles, size=min(600, tiles.size), replace=False)
tot = dict(q=0, top=0, bot=0, lef=0, rig=0, steps_h=0, steps_v=0, lanes=0)
for t in tiles:
    ty, tx = divmod(int(t), gx)
    for q in range(4):
        x0, y0 = tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8
        ys, xs = np.mgrid[y0:y0 + 8, x0:x0 + 8]
        inside = (xs < W) & (ys < H)
        if not inside.any():
            continue
        nc = ncon[np.minimum(ys, H - 1), np.minimum(xs, W - 1)] * inside
        depth = int(nc.max())
        if depth == 0:
            continue
        ids = vals[ranges[t, 0]:ranges[t, 0] + depth]
        mx, my = m2[ids, 0][:, None, None], m2[ids, 1][:, None, None]
        A, B, C, o = (co[ids, k][:, None, None] for k in range(4))
        dx, dy = mx - xs[None].astype(np.float32), my - ys[None].astype(np.float32)
        power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
        alpha = np.minimum(0.99, o * np.exp(power))
        pos = np.arange(depth)[:, None, None]
        hit = (power <= 0) & (alpha >= 1.0 / 255.0) & (pos < nc[None]) & inside[None]
        anyq = hit.any((1, 2))
        top, bot = hit[:, :4].any((1, 2)), hit[:, 4:].any((1, 2))
        lef, rig = hit[:, :, :4].any((1, 2)), hit[:, :, 4:].any((1, 2))
        tot["q"] += int(anyq.sum()); tot["top"] += int(top.sum()); tot["bot"] += int(bot.sum())
        tot["lef"] += int(lef.sum()); tot["rig"] += int(rig.sum())
        tot["steps_h"] += max(int(top.sum()), int(bot.sum())); tot["steps_v"] += max(int(lef.sum()), int(rig.sum()))
        tot["lanes"] += int(hit.sum())
print(tot)
print("hit lanes per hitting (quadrant, entry): %.1f of 64" % (tot["lanes"] / tot["q"]))
print("8x4 halves: (top + bottom) / (2 x quadrant) = %.3f; lock-step steps max(top, bottom) / quadrant = %.3f" % ((tot["top"] + tot["bot"]) / (2 * tot["q"]), tot["steps_h"] / tot["q"]))
print("4x8 halves: (left + right) / (2 x quadrant) = %.3f; lock-step steps max(left, right) / quadrant = %.3f" % ((tot["lef"] + tot["rig"]) / (2 * tot["q"]), tot["steps_v"] / tot["q"]))
