"""Accuracy of the RENDER-level backward sums (what k_render_backward leaves in the per-Gaussian gradient records:
dL/d{mean2D, conic, colour, opacity}) of the HIP library and of the reference build (oracle/_ref), each measured against
the oracle's float64 render backward (orc_render_backward_fp64: every per-(pixel, entry) term in double; --f32terms: against
the oracle's double SUMS of the reference's float32 terms instead, which flatters the reference build: its own per-term
rounding is inside that yardstick), over a set of fuzz cases (tests/test_gpu_fuzz.py::_case).

usage: python scripts/bwd_accuracy.py [case ...] [--range a b] [--repeat n]
Prints, per tensor: the error of each implementation in units of max|g| (max over elements), its median / 90 % / max over
the cases, and in how many cases the library is further from the double sums than the reference build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
import util, test_gpu_fuzz as F
from oracle.oracle import Oracle, Reference
from diff_gaussian_rasterization import _native as N


def grad_records(scene, dev, dL):
    """forward + backward of the product; returns the [P, 16] gradient records of view 0 (render-level sums)"""
    def t(a):
        return torch.empty(0) if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    s = scene
    args = (t(s.bg), t(s.means3D), t(s.colors_precomp), t(s.opacities), t(s.scales), t(s.rotations), s.scale_modifier,
            t(s.cov3D_precomp), t(s.viewmatrix.reshape(4, 4)), t(s.projmatrix.reshape(4, 4)), s.tanfovx, s.tanfovy, s.H, s.W,
            t(s.shs), s.sh_degree, t(s.campos), s.prefiltered, False)
    R, color, radii, geom, binning, img = N.rasterize_gaussians(*args, need_backward=True)
    N.rasterize_gaussians_backward(args[0], args[1], radii, args[2], args[4], args[5], s.scale_modifier, args[7], args[8], args[9],
                                   s.tanfovx, s.tanfovy, t(dL), args[14], s.sh_degree, args[16], geom, R, binning, img, False)
    return N.grad_records(geom, s.P).cpu().numpy().astype(np.float64)


def main():
    a = sys.argv[1:]
    cases, rep, f32terms = [], 1, False
    i = 0
    while i < len(a):
        if a[i] == "--f32terms":
            f32terms = True; i += 1
        elif a[i] == "--range":
            cases += list(range(int(a[i + 1]), int(a[i + 2]))); i += 3
        elif a[i] == "--repeat":
            rep = int(a[i + 1]); i += 2
        else:
            cases.append(int(a[i])); i += 1
    if not cases:
        cases = [149, 14139, 14397] + list(range(0, 64))
    dev = torch.device("cuda:0")
    ref, orc = Reference("strict"), Oracle()
    names = ("mean2D", "conic", "colour", "opacity")
    stats = {n: [] for n in names}
    for c in cases:
        s, mode = F._case(c)
        if s.P == 0:
            continue
        dL = util.seeded_dL(s, seed=77 + c)
        r, gr = ref.forward_backward(s, dL)
        o, go = orc.forward_backward(s, dL, exact=True)
        if not f32terms:
            go = go["exact"]
        want = dict(mean2D=np.asarray(go["dL_dmean2D"], np.float64).reshape(s.P, -1)[:, :2],
                    conic=np.asarray(go["dL_dconic"], np.float64).reshape(s.P, 4)[:, [0, 1, 3]],
                    colour=np.asarray(go["dL_dcolor"], np.float64).reshape(s.P, 3),
                    opacity=np.asarray(go["dL_dopacity"], np.float64).reshape(s.P, 1))
        build = dict(mean2D=np.asarray(gr["dL_dmean2D"], np.float64).reshape(s.P, -1)[:, :2],
                     conic=np.asarray(gr["dL_dconic"], np.float64).reshape(s.P, 4)[:, [0, 1, 3]],
                     colour=np.asarray(gr["dL_dcolor"], np.float64).reshape(s.P, 3),
                     opacity=np.asarray(gr["dL_dopacity"], np.float64).reshape(s.P, 1))
        for _ in range(rep):
            rec = grad_records(s, dev, dL)
            lib = dict(mean2D=rec[:, 0:2], conic=rec[:, 2:5], colour=rec[:, 5:8], opacity=rec[:, 8:9])
            line = "case %5d P %4d %3dx%-3d" % (c, s.P, s.W, s.H)
            for n in names:
                m = np.abs(want[n]).max()
                if m == 0:
                    continue
                el, eb = np.abs(lib[n] - want[n]).max() / m, np.abs(build[n] - want[n]).max() / m
                # rms over the elements, too: the max is one element, the rms is what the per-Gaussian chain sees on average
                rl, rb = np.sqrt(((lib[n] - want[n]) ** 2).mean()) / m, np.sqrt(((build[n] - want[n]) ** 2).mean()) / m
                stats[n].append((el, eb, rl, rb))
                line += "  %s %.1e/%.1e" % (n[:4], el, eb)
            if len(cases) <= 16:
                print(line + "   (lib/ref max err of max|g|)")
                for n in ("mean2D", "conic"):
                    e = np.abs(lib[n] - want[n])
                    g, comp = np.unravel_index(e.argmax(), e.shape)
                    print("      worst %s: gaussian %d comp %d radius %d tiles %d lib %.8g ref %.8g exact %.8g  max|g| %.3g  conic_o %s" % (
                        n, g, comp, r["radii"][g], r["tiles_touched"][g], lib[n][g, comp], build[n][g, comp], want[n][g, comp],
                        np.abs(want[n]).max(), r["conic_opacity"][g]))
    print("%d cases x %d runs; error against %s, in units of max|g| of the tensor" % (
        len(cases), rep, "the oracle's double sums of float32 terms" if f32terms else "the float64 render backward"))
    print("%-8s %-34s %-34s %s" % ("tensor", "library: median / p90 / max", "reference build: median / p90 / max", "lib worse than ref (max | rms)"))
    for n in names:
        x = np.array(stats[n])
        if x.size == 0:
            continue
        q = lambda v: "%.2e / %.2e / %.2e" % (np.median(v), np.quantile(v, 0.9), v.max())  # noqa: E731
        print("%-8s %-34s %-34s %d | %d of %d   geomean lib/ref: max %.2f rms %.2f" % (
            n, q(x[:, 0]), q(x[:, 1]), int((x[:, 0] > x[:, 1]).sum()), int((x[:, 2] > x[:, 3]).sum()), x.shape[0],
            np.exp(np.mean(np.log((x[:, 0] + 1e-30) / (x[:, 1] + 1e-30)))), np.exp(np.mean(np.log((x[:, 2] + 1e-30) / (x[:, 3] + 1e-30))))))


if __name__ == "__main__":
    main()
