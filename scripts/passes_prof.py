import sys, os, time, json
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import raster_passes as rp, camera, synth
from diff_gaussian_rasterization import _native
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-256", seed=0)
g = synth.make_gaussians(cloud, profile="inference", seed=1)
sf = cloud["scale_factor"]; radius = float(np.sqrt(3) / sf * 6)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
means, shs, opac, rots = t(g["means3D"]), t(g["shs"]), t(g["opacities"]), t(g["rotations"])
dec_s = t((g["scales"] / radius).astype(np.float32)); normals = torch.nn.functional.normalize(means, dim=-1)
Hs = camera.circle_path(12, 0, 3, [90, 0]); h = w = 512; bg = torch.ones(3)
def fused():
    return rp.render_passes(means, opac, dec_s, rots, shs, Hs, h, w, 45.0, bg, sf, normals=normals)
for _ in range(3): fused()
torch.cuda.synchronize()
_native.set_profiling(True)
t0 = time.perf_counter()
for _ in range(5): fused()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5 * 1e3
prof = _native.get_profile(); _native.set_profiling(False)
acc = {}
for n, ms in prof: acc[n] = acc.get(n, 0) + ms / 5
print(json.dumps({"ms": dt, "stages": {k: round(v, 4) for k, v in acc.items()}, "sum": sum(acc.values())}))
