"""Where a single-stream frame's wall time goes on the host side: per-call wall times of the binding (forward incl.
its one stream sync, loss, backward launches) against the GPU's own kernel time.  usage: python scripts/host_timeline.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import synth, camera
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _native as N

dev = torch.device("cuda:0")
W, H = 1920, 1080
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
views = camera.circle_views(12, fov_deg=45., width_px=W, height_px=H)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
means, opac, scales, rots, shs = (t(g[k]).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "shs"))
means2D = torch.zeros_like(means, requires_grad=True)
G = torch.rand(3, H, W, device=dev) * 2 - 1
bg = torch.ones(3, device=dev)
sets = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=v["tanfovx"], tanfovy=v["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=v["viewmatrix"].to(dev), projmatrix=v["projmatrix"].to(dev), sh_degree=1, campos=v["campos"].to(dev),
        prefiltered=False, debug=False) for v in views]
acc = {"forward": 0.0, "loss": 0.0, "backward": 0.0, "zero_grad": 0.0}
def frame(i, rec):
    t0 = time.perf_counter()
    img, radii = GaussianRasterizer(sets[i % 12])(means3D=means, means2D=means2D, shs=shs, colors_precomp=None, opacities=opac,
                                                 scales=scales, rotations=rots, cov3D_precomp=None)
    t1 = time.perf_counter()
    loss = (img * G).sum()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    for x in (means, opac, scales, rots, shs, means2D):
        x.grad = None
    t4 = time.perf_counter()
    if rec:
        acc["forward"] += t1 - t0; acc["loss"] += t2 - t1; acc["backward"] += t3 - t2; acc["zero_grad"] += t4 - t3
for i in range(12): frame(i, False)
torch.cuda.synchronize()
n = 48
t0 = time.perf_counter()
for i in range(n): frame(i, True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
print("wall per frame %.3f ms" % (wall * 1e3), {k: round(v / n * 1e3, 3) for k, v in acc.items()})
# same loop with profiling on: GPU kernel time per frame of the library's own stages
N.set_profiling(True); N.get_profile()
for i in range(12): frame(i, False)
torch.cuda.synchronize()
ms = {}
for name, tt in N.get_profile(): ms[name] = ms.get(name, 0.0) + tt / 12
print("library kernel time per frame %.3f ms" % sum(ms.values()), {k: round(v, 3) for k, v in ms.items()})
