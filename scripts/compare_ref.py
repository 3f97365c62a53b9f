"""Headline workload: timing of the reference build's own kernels (oracle/_ref, hipEvents) next to the product's
per-stage timing, plus tile-list statistics.  Usage: python scripts/compare_ref.py [view] [workload] [W H]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import synth, camera
from oracle.oracle import Scene, Reference
import util
from diff_gaussian_rasterization import _native as N

view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
workload = sys.argv[2] if len(sys.argv) > 2 else "synth-THuman-800K"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
profile = sys.argv[5] if len(sys.argv) > 5 else "training"
dev = torch.device("cuda:0")
cloud = synth.make_cloud(workload, seed=0)
g = synth.make_gaussians(cloud, profile=profile, seed=1)
v = camera.circle_views(12, fov_deg=45., width_px=W, height_px=H)[view]
sc = Scene(W=W, H=H, tanfovx=v["tanfovx"], tanfovy=v["tanfovy"], bg=np.ones(3, np.float32), means3D=g["means3D"],
           opacities=g["opacities"], viewmatrix=v["viewmatrix"].numpy(), projmatrix=v["projmatrix"].numpy(),
           campos=v["campos"].numpy(), shs=g["shs"], scales=g["scales"], rotations=g["rotations"], sh_degree=1)
dL = util.seeded_dL(sc)
res = {"workload": workload, "view": view, "W": W, "H": H, "profile": profile}
for variant in ("strict", "fast"):
    if Reference.available(variant):
        f, b = Reference(variant).bench(sc, dL, warmup=2, iters=5)
        res["ref_%s_ms" % variant] = {"forward": round(f, 3), "backward": round(b, 3)}
N.set_profiling(True)
for it in range(6):
    if it == 1:
        N.get_profile()
    p, gp = util.run_product(sc, dev, dL_dpix=dL) if it == 0 else (None, None)
    if it > 0:
        def t(a):
            return torch.empty(0) if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        if it == 1:
            args = (t(sc.bg), t(sc.means3D), t(None), t(sc.opacities), t(sc.scales), t(sc.rotations), 1.0, t(None),
                    t(sc.viewmatrix.reshape(4, 4)), t(sc.projmatrix.reshape(4, 4)), sc.tanfovx, sc.tanfovy, H, W, t(sc.shs), 1,
                    t(sc.campos), False, False)
            tdL = t(dL)
            N.get_profile()
        R, color, radii, geom, binning, img = N.rasterize_gaussians(*args, need_backward=True)
        N.rasterize_gaussians_backward(args[0], args[1], radii, args[2], args[4], args[5], 1.0, args[7], args[8], args[9],
                                       sc.tanfovx, sc.tanfovy, tdL, args[14], 1, args[16], geom, R, binning, img, False)
torch.cuda.synchronize()
ms = {}
for name, tt in N.get_profile():
    ms.setdefault(name, []).append(tt)
res["product_ms"] = {k: round(float(np.mean(x)), 4) for k, x in ms.items()}
res["product_fwd_ms"] = round(sum(v_ for k, v_ in res["product_ms"].items() if "backward" not in k), 4)
res["product_bwd_ms"] = round(sum(v_ for k, v_ in res["product_ms"].items() if "backward" in k), 4)
need = N.query("TILE_NEED", sc.P, W, H, R, geom, binning, img).cpu().numpy()
rng = N.query("RANGES", sc.P, W, H, R, geom, binning, img).cpu().numpy()
ln = (rng[:, 1] - rng[:, 0])
res["tiles"] = {"T": int(ln.size), "nonempty": int((ln > 0).sum()), "len_max": int(ln.max()), "len_mean_nonempty": float(ln[ln > 0].mean()),
                "need_sum": int(need.sum()), "need_max": int(need.max()), "need_p99": float(np.percentile(need[ln > 0], 99)),
                "need_p90": float(np.percentile(need[ln > 0], 90)), "need_median": float(np.median(need[ln > 0])),
                "R": int(R)}
print(json.dumps(res))
