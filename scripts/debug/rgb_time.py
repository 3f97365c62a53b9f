"""debug: where the time of the 12-view SH pass at 1024^2 goes, literal per-view calls vs one rasterize_views call"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import camera, synth, raster_passes as rp
from diff_gaussian_rasterization import _native as N
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
means, shs, opac, rots, scales = t(g["means3D"]), t(g["shs"]), t(g["opacities"]), t(g["rotations"]), t(g["scales"])
sf = float(cloud["scale_factor"]); radius = float(np.sqrt(3) / sf * 6)
dec_s = (scales / radius).contiguous()
Hs = camera.circle_path(12, 0, 3, [90, 0]).unsqueeze(0)
def run(bv):
    with torch.no_grad():
        return rp.rasterize_views([means], [opac], [dec_s], [rots], Hs, 512, 512, 45.0, torch.ones(3), sf, shs_list=[shs], sh_degree=1, batch_views=bv)
for bv in (False, True, False, True):
    for _ in range(2): run(bv)
    torch.cuda.synchronize()
    N.set_profiling(True); N.get_profile()
    t0 = time.perf_counter(); run(bv); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    prof = {}
    for k, v in N.get_profile(): prof[k] = prof.get(k, 0.0) + v
    N.set_profiling(False)
    print("batch_views=%s: host %.2f ms, total %.2f ms; library stages (ms):" % (bv, (t1 - t0) * 1e3, (t2 - t0) * 1e3), {k: round(v, 3) for k, v in prof.items()}, "sum %.2f" % sum(prof.values()))
