"""Is a fuzz case's result the same when it runs after other cases (reused arena memory) as when it runs first?
usage: python scripts/diag_fuzz_state.py <case> [n_before]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
import util, test_gpu_fuzz as F
dev = torch.device("cuda:0")
i = int(sys.argv[1]); nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
s, mode = F._case(i)
dL = util.seeded_dL(s, seed=77 + i)
print("case", i, "P", s.P, "W,H", s.W, s.H, "mode", mode, "M", None if s.shs is None else s.shs.shape[1], "D", s.sh_degree, "cov3d", s.cov3D_precomp is not None)
p0, g0 = util.run_product(s, dev, dL_dpix=dL)
for j in range(i - nb, i):
    sj, mj = F._case(j)
    util.run_product(sj, dev, dL_dpix=util.seeded_dL(sj, seed=77 + j))
for rep in range(3):
    p1, g1 = util.run_product(s, dev, dL_dpix=dL)
    for k in g0:
        if g0[k].size and not np.array_equal(g0[k], g1[k]):
            d = np.abs(g0[k].astype(np.float64) - g1[k]).reshape(g0[k].shape[0], -1).max(1)
            rows = np.nonzero(d > 0)[0]
            print(" rep", rep, k, "differs in", rows.size, "rows; max", d.max(), "of max|g|", np.abs(g0[k]).max(), "rows", rows[:8],
                  "radii", p0["radii"][rows[:8]], "tiles", p0["tiles_touched"][rows[:8]])
    for k in ("out_color", "n_contrib", "final_T", "vals"):
        if not np.array_equal(p0[k], p1[k]):
            print(" rep", rep, "forward", k, "differs")
print("done")
