"""The reference's actual use: 12 circle views x 4 passes (xyz, rgb, hitmap, normal) at 512^2 x super-sample 2 on
a voxelised 200K cloud -- literal per-pass calls vs the fused render_passes; and the reference build's kernels."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import raster_passes as rp, camera, synth
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-256", seed=0)
g = synth.make_gaussians(cloud, profile="inference", seed=1)
sf = cloud["scale_factor"]; radius = float(np.sqrt(3) / sf * 6)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
means, shs, opac, rots = t(g["means3D"]), t(g["shs"]), t(g["opacities"]), t(g["rotations"])
dec_s = t((g["scales"] / radius).astype(np.float32)); normals = torch.nn.functional.normalize(means, dim=-1)
Hs = camera.circle_path(12, 0, 3, [90, 0]); h = w = 512; bg = torch.ones(3)
def literal():
    Hb = Hs.unsqueeze(0)
    with torch.no_grad():
        a = rp.rasterize_views([means], [opac], [dec_s], [rots], Hb, h, w, 45.0, bg, sf, colors_list=[means])
        b = rp.rasterize_views([means], [opac], [dec_s], [rots], Hb, h, w, 45.0, bg, sf, shs_list=[shs])
        c = rp.rasterize_views([means], [opac], [dec_s], [rots], Hb, h, w, 45.0, bg, sf, colors_list=[torch.ones_like(means)])
        d = rp.rasterize_views([means], [opac], [dec_s], [rots], Hb, h, w, 45.0, bg, sf, colors_list=[normals], normalize_camera_normal=True)
    return a, b, c, d
def fused():
    return rp.render_passes(means, opac, dec_s, rots, shs, Hs, h, w, 45.0, bg, sf, normals=normals)
res = {}
for name, fn in (("literal_4x12_calls", literal), ("fused_render_passes", fused)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); res[name + "_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
print(json.dumps(res))
