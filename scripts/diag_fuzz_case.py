"""Per-tensor gradient differences of one fuzz case: library vs reference build vs plain-C oracle (double sums).
usage: python scripts/diag_fuzz_case.py <case>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
import util, test_gpu_fuzz as F
from oracle.oracle import Oracle, Reference

i = int(sys.argv[1])
s, mode = F._case(i)
print("case", i, "P", s.P, "W,H", s.W, s.H, "mode", mode, "M", s.M, "D", s.sh_degree, "mod", s.scale_modifier)
dL = util.seeded_dL(s, seed=77 + i)
r, gr = Reference("strict").forward_backward(s, dL)
o, go = Oracle().forward_backward(s, dL)
p, gp = util.run_product(s, torch.device("cuda:0"), dL_dpix=dL)
print("R", p["R"], "visible", int((p["radii"] > 0).sum()), "max n_contrib", int(p["n_contrib"].max()))
for k in gp:
    a, b, c = gp[k].astype(np.float64), gr[k].astype(np.float64), go[k].astype(np.float64)
    if a.size == 0:
        continue
    m = np.abs(c).max() + 1e-300
    j = np.unravel_index(np.abs(a - c).argmax(), a.shape)
    print("%-12s max|g| %9.3g  lib-oracle %8.2e  ref-oracle %8.2e  lib-ref %8.2e   (rel to max)   worst idx %s: lib %.7g ref %.7g oracle %.7g" % (
        k, m, np.abs(a - c).max() / m, np.abs(b - c).max() / m, np.abs(a - b).max() / m, j, a[j], b[j], c[j]))
if len(sys.argv) > 2:   # details of the Gaussian with the worst dL_dmean3D row
    a, c = gp["dL_dmean3D"].astype(np.float64), go["dL_dmean3D"].astype(np.float64)
    g = int(np.linalg.norm(a - c, axis=1).argmax())
    print("gaussian", g, "radius", p["radii"][g], "mean", s.means3D[g], "scale", None if s.scales is None else s.scales[g],
          "rot", None if s.rotations is None else s.rotations[g], "opacity", s.opacities[g])
    print(" conic_opacity", p["conic_opacity"][g], "means2D", p["means2D"][g], "depth", p["depths"][g], "tiles", p["tiles_touched"][g])
    for k in ("dL_dmean2D", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot", "dL_dopacity"):
        if gp[k].size:
            print(" %-11s lib %s\n %11s ref %s\n %11s orc %s" % (k, gp[k][g], "", gr[k][g].reshape(gp[k][g].shape), "", go[k][g].reshape(gp[k][g].shape)))
    print(" dL_dconic   ref %s  orc %s" % (gr["dL_dconic"][g], go["dL_dconic"][g]))
