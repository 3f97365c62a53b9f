"""Diagnostic: product vs oracle vs reference build on every test scene (prints a table)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
import util
from oracle.oracle import Oracle, Reference
dev = torch.device("cuda:0")
orc = Oracle()
ref = Reference("strict") if Reference.available("strict") else None
names = sys.argv[1:] or util.SCENES
for n in names:
    s = util.build_scene(n)
    dL = util.seeded_dL(s)
    o, go = orc.forward_backward(s, dL)
    p, gp = util.run_product(s, dev, dL_dpix=dL)
    line = "%-18s P=%6d R=%7d/%7d" % (n, s.P, p["R"], o["R"])
    if s.P:
        for k in ("radii", "tiles_touched", "vals", "keys", "ranges", "n_contrib"):
            a, b = p[k], o[k]
            line += " %s:%s" % (k[:5], "ok" if a.shape == b.shape and (a == b).all() else "DIFF(%d)" % ((a != b).sum() if a.shape == b.shape else -1))
        vis = o["radii"] > 0
        for k in ("means2D", "depths", "conic_opacity", "rgb"):
            line += " %s:%d" % (k[:5], util.ulp_diff(p[k][vis], o[k][vis]).max(initial=0))
        err = np.abs(p["out_color"] - o["out_color"]).max(axis=0)
        line += " rgb_maxerr=%.3g n>1e-4=%d" % (err.max(), (err > 1e-4).sum())
        for k in gp:
            a, b = gp[k].astype(np.float64).ravel(), go[k].astype(np.float64).ravel()
            if b.size:
                line += " %s=%.1e" % (k[4:8], np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    print(line, flush=True)
    if ref is not None:
        r, gr = ref.forward_backward(s, dL)
        line = "   vs ref_strict:      R=%7d" % r["R"]
        if s.P:
            for k in ("radii", "tiles_touched", "vals", "keys", "ranges", "n_contrib"):
                a, b = p[k], r[k]
                line += " %s:%s" % (k[:5], "ok" if a.shape == b.shape and (a == b).all() else "DIFF(%d)" % ((a != b).sum() if a.shape == b.shape else -1))
            vis = r["radii"] > 0
            for k in ("means2D", "depths", "conic_opacity", "rgb"):
                line += " %s:%d" % (k[:5], util.ulp_diff(p[k][vis], r[k][vis]).max(initial=0))
            err = np.abs(p["out_color"] - r["out_color"]).max(axis=0)
            line += " rgb_maxerr=%.3g n>0=%d" % (err.max(), (err > 0).sum())
            for k in gp:
                a, b = gp[k].astype(np.float64).ravel(), gr[k].astype(np.float64).ravel()
                if b.size:
                    line += " %s=%.1e" % (k[4:8], np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
        print(line, flush=True)
