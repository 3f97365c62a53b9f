"""CPU pricing of float32 FORMULATIONS of the render backward's per-(pixel, entry) weights (no GPU): the oracle's model function
(orc_render_backward_model: float32 terms, double sums, so only the formulation's own rounding shows) against the float64 render
backward (orc_render_backward_fp64), over the fuzz cases of tests/test_gpu_fuzz.py.

  mode 0  the reference's back-to-front walk: T rebuilt by IEEE division, accum_rec recurrence
  mode 1  front-to-back: T by the forward's own multiply chain, dL_dalpha = T d - (S_tot - S_<=i) / (1 - alpha)   

usage: python scripts/analysis/bwd_formulations.py [--range a b]      (default: the 503 cases of profiles/r04_bwd_accuracy.txt)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np  # noqa: E402

import util  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
from oracle.oracle import Oracle, _f32, _fp  # noqa: E402


def sums(orc, scene, dL, mode):
    s = scene.as_struct()
    stp = orc._call_forward(s)
    dpix = _f32(dL).reshape(3, scene.H, scene.W)
    x9 = np.zeros((scene.P, 9), np.float64)
    if mode == "fp64":
        orc.lib.orc_render_backward_fp64(C.byref(s), stp, dpix.ctypes.data_as(_fp), x9.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(1))
    else:
        orc.lib.orc_render_backward_model(C.byref(s), stp, dpix.ctypes.data_as(_fp), x9.ctypes.data_as(C.POINTER(C.c_double)),
                                          C.c_int(mode), C.c_int(1))
    orc.lib.orc_free(stp)
    return dict(mean2D=x9[:, 0:2], conic=x9[:, 2:5], opacity=x9[:, 5:6], colour=x9[:, 6:9])


def main():
    a = sys.argv[1:]
    cases = [149, 14139, 14397] + list(range(0, 400)) + list(range(14100, 14200))
    if a[:1] == ["--range"]:
        cases = list(range(int(a[1]), int(a[2])))
    orc = Oracle()
    names = ("mean2D", "conic", "colour", "opacity")
    err = {m: {n: [] for n in names} for m in (0, 1)}
    worst = {n: (0.0, None) for n in names}
    for c in cases:
        s, _ = F._case(c)
        if s.P == 0:
            continue
        dL = util.seeded_dL(s, seed=77 + c)
        want = sums(orc, s, dL, "fp64")
        got = {m: sums(orc, s, dL, m) for m in (0, 1)}
        for n in names:
            mx = np.abs(want[n]).max()
            if mx == 0:
                continue
            e = [np.abs(got[m][n] - want[n]).max() / mx for m in (0, 1)]
            err[0][n].append(e[0]); err[1][n].append(e[1])
            if e[1] / (e[0] + 1e-30) > worst[n][0]:
                worst[n] = (e[1] / (e[0] + 1e-30), (c, e[0], e[1], float(s.opacities.max())))
    print("%d cases; per-term rounding of each formulation against the float64 render backward, max over elements in units of max|g|" % len(cases))
    print("%-8s %-36s %-36s %s" % ("tensor", "mode 0 reference walk: median/p90/max", "mode 1 front-to-back: median/p90/max", "geomean 1/0   cases 1 worse"))
    for n in names:
        x0, x1 = np.array(err[0][n]), np.array(err[1][n])
        q = lambda v: "%.2e / %.2e / %.2e" % (np.median(v), np.quantile(v, 0.9), v.max())  # noqa: E731
        print("%-8s %-36s %-36s %.2f   %d of %d   worst ratio: case %s" % (
            n, q(x0), q(x1), np.exp(np.mean(np.log((x1 + 1e-30) / (x0 + 1e-30)))), int((x1 > x0).sum()), len(x0), worst[n][1]))


if __name__ == "__main__":
    main()
