"""VERDICT r05 item 4, priced before building it: would a PER-BATCH switch to the sub-quadrant moments (render_bwd.hip, SUBQ)
stay off on the headline workload?  The float32 rounding the switch cures grows with m = (b / sigma)^2, b = distance of the splat
centre from the quadrant centre, sigma its size along b: m = A bx^2 + 2 B bx by + C by^2 with the conic (A, B, C) = twice the
|power| at the quadrant centre, one compare per staged entry.  A batch (eight surviving entries of one quadrant) takes the
expensive path when ANY of its entries has m above the threshold.  CPU only (the oracle renders the view; numpy does the rest).
usage: python scripts/analysis/subq_flag_hist.py [points=800000]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import util
from oracle.oracle import Oracle
from pcrender import camera, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 800_000
W, H = 1920, 1080
cloud = synth.make_cloud("synth-THuman-800K", seed=0, P=P)
g = synth.make_gaussians(cloud, profile="training", seed=1)
view = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)[0]
s = util.scene_from(g, view, W, H, bg=(1, 1, 1))
o = Oracle().forward(s, nthreads=8)
m2, co, ranges, vals, ncon = o["means2D"], o["conic_opacity"], o["ranges"], o["vals"], o["n_contrib"]
gx = o["gridx"]
TH = (4.0, 9.0, 16.0, 20.0, 25.0, 36.0)
hist_edges = np.array([0, 1, 2, 4, 6, 9, 12, 16, 20, 25, 36, 1e9])
hist = np.zeros(len(hist_edges) - 1)
batches = 0
flagged = np.zeros(len(TH))
entries = 0
lx, ly = np.meshgrid(np.arange(8.0), np.arange(8.0))
for t in np.nonzero(ranges[:, 1] > ranges[:, 0])[0]:
    tx, ty = int(t % gx), int(t // gx)
    for qy in range(2):
        for qx in range(2):
            x0, y0 = tx * 16 + qx * 8, ty * 16 + qy * 8
            if x0 >= W or y0 >= H:
                continue
            nc = ncon[y0:y0 + 8, x0:x0 + 8]
            need = int(nc.max())
            if need == 0:
                continue
            ids = vals[ranges[t, 0]:ranges[t, 0] + need]
            X, Y = m2[ids, 0], m2[ids, 1]
            A, B, C, O = co[ids, 0], co[ids, 1], co[ids, 2], co[ids, 3]
            # entries that reach alpha >= 1/255 at some pixel of the quadrant that is still walking (what the kernel stages and evaluates)
            px = (x0 + lx.ravel())[None, :]
            py = (y0 + ly.ravel())[None, :]
            dx, dy = X[:, None] - px, Y[:, None] - py
            power = -0.5 * (A[:, None] * dx * dx + C[:, None] * dy * dy) - B[:, None] * dx * dy
            alpha = np.minimum(0.99, O[:, None] * np.exp(power))
            pos = np.arange(need)[:, None]
            hit = (power <= 0) & (alpha >= 1.0 / 255.0) & (pos < nc.ravel()[None, :])
            keep = hit.any(axis=1)
            if not keep.any():
                continue
            bx, by = X[keep] - (x0 + 3.5), Y[keep] - (y0 + 3.5)
            m = A[keep] * bx * bx + 2 * B[keep] * bx * by + C[keep] * by * by
            hist += np.histogram(m, hist_edges)[0]
            entries += m.size
            nb = (m.size + 7) // 8
            batches += nb
            mm = np.full(nb * 8, 0.0)
            mm[:m.size] = m[::-1]            # the walk is back to front
            mb = mm.reshape(nb, 8).max(axis=1)
            for i, th in enumerate(TH):
                flagged[i] += (mb > th).sum()
print("synth-THuman-800K view 0, %d Gaussians, 1920x1080: %d evaluated (quadrant, entry) pairs in %d batches of eight" % (P, entries, batches))
print("m = (b / sigma)^2 of the evaluated entries:")
for a, b, h in zip(hist_edges[:-1], hist_edges[1:], hist):
    print("   %5.0f .. %-6s %6.2f %%" % (a, "inf" if b > 1e8 else "%.0f" % b, 100.0 * h / entries))
for th, f in zip(TH, flagged):
    print("threshold m > %4.0f: %5.1f %% of the batches would take the sub-quadrant path" % (th, 100.0 * f / batches))
