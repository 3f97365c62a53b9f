"""How many pair-evaluation steps would the forward render take if the halves / quarters of an 8x8 quadrant culled and walked their
own survivor lists in lock-step per round of 64 entries (one wave, lanes of different sub-blocks on different entries)?
For the consumed (quadrant, entry) evaluations of a benchmark view: survivors = entries with a real hit (alpha >= 1/255 at a pixel
that is still before its last contributor) in the (sub-)block -- the exact-safe rectangle test keeps a few more.
steps(8x8) = sum over rounds of ceil(n / 2); steps(split) = sum over rounds of max over sub-blocks of ceil(n_sub / 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
import util
from pcrender import camera, synth
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
v = camera.circle_views(12, fov_deg=45.0, width_px=1920, height_px=1080)[1]
s = util.scene_from(g, v, 1920, 1080, bg=(1, 1, 1))
p, _ = util.run_product(s, dev, reference_lists=False)
W, H = 1920, 1080
gx = (W + 15) // 16
vals, ranges, ncon = p["vals"], p["ranges"].reshape(-1, 2), p["n_contrib"]
m2, co = p["means2D"], p["conic_opacity"]
rng = np.random.default_rng(0)
tiles = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
tiles = rng.choice(tiles, size=min(500, tiles.size), replace=False)
splits = {
    "8x8 (now)": [(slice(0, 8), slice(0, 8))],
    "two 8x4 (top / bottom)": [(slice(0, 4), slice(0, 8)), (slice(4, 8), slice(0, 8))],
    "two 4x8 (left / right)": [(slice(0, 8), slice(0, 4)), (slice(0, 8), slice(4, 8))],
    "four 4x4": [(slice(a, a + 4), slice(b, b + 4)) for a in (0, 4) for b in (0, 4)],
    "four 8x2 rows-pairs": [(slice(a, a + 2), slice(0, 8)) for a in (0, 2, 4, 6)],
}
steps = {k: 0 for k in splits}
evals = {k: 0 for k in splits}      # sum over sub-blocks of their own pair counts (what independent waves would evaluate)
lanes = 0
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
        lanes += int(hit.sum())
        nr = (depth + 63) // 64
        pad = nr * 64 - depth
        for name, blocks in splits.items():
            per = []
            for (sy, sx) in blocks:
                a = hit[:, sy, sx].any((1, 2))
                a = np.concatenate([a, np.zeros(pad, bool)]).reshape(nr, 64).sum(1)      # survivors per round
                per.append((a + 1) // 2)
            per = np.stack(per, 0)
            steps[name] += int(per.max(0).sum())
            evals[name] += int(per.sum())
base = steps["8x8 (now)"]
print("tiles sampled %d; hit (pixel, entry) pairs %d; 8x8 pair steps %d (%.1f hit lanes per entry evaluated)" % (len(tiles), lanes, base, lanes / (2.0 * base)))
for name in splits:
    print("  %-26s lock-step pair steps %8d = %.3f of now      (sum of the sub-blocks' own steps %.3f)" % (name, steps[name], steps[name] / base, evals[name] / base))
