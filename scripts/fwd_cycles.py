import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import synth, camera
from diff_gaussian_rasterization import _native as N
W, H = 1920, 1080
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
v = camera.circle_views(12, fov_deg=45., width_px=W, height_px=H)[0]
def t(a): return torch.empty(0) if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
args = (torch.ones(3, device=dev), t(g["means3D"]), t(None), t(g["opacities"]), t(g["scales"]), t(g["rotations"]), 1.0, t(None),
        v["viewmatrix"].to(dev), v["projmatrix"].to(dev), v["tanfovx"], v["tanfovy"], H, W, t(g["shs"]), 1, v["campos"].to(dev), False, False)
for it in range(3):
    R, color, radii, geom, binning, img = N.rasterize_gaussians(*args, need_backward=True)
torch.cuda.synchronize()
P = g["means3D"].shape[0]
clk = N.query("TILE_CLOCK", P, W, H, R, geom, binning, img).cpu().numpy().astype(np.uint64)
need = N.query("TILE_NEED", P, W, H, R, geom, binning, img).cpu().numpy()
dur = (clk[:, 1].astype(np.int64) - clk[:, 0].astype(np.int64)) * 10e-3
stage = (clk[:, 2] >> np.uint64(32)).astype(np.int64); loop = (clk[:, 2] & np.uint64(0xffffffff)).astype(np.int64); groups = clk[:, 3].astype(np.int64)
idx = np.argsort(-dur)[:12]
print("tile need dur_us rounds total_cyc loop_cyc groups(q0) cyc/group")
for i in idx:
    rounds = (need[i] + 63) // 64
    print(i, need[i], round(dur[i], 1), rounds, stage[i], loop[i], groups[i], round(loop[i] / max(groups[i], 1), 1), round(stage[i] / max(rounds, 1)))
ne = need > 0
print("all nonempty: mean cyc/group %.1f ; total stage %.3g loop %.3g" % ((loop[ne] / np.maximum(groups[ne], 1)).mean(), stage[ne].sum(), loop[ne].sum()))
