"""Host-side cost of the literal per-view call (GaussianRasterizer per view, forward only): wall time per call with the
GPU kept busy vs. the kernels' own time, and a cProfile of the Python path."""
import sys, os, time, json, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import raster_passes as rp, camera, synth
from diff_gaussian_rasterization import GaussianRasterizer, _native
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-256", seed=0)
g = synth.make_gaussians(cloud, profile="inference", seed=1)
sf = cloud["scale_factor"]; radius = float(np.sqrt(3) / sf * 6)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
means, shs, opac, rots = t(g["means3D"]), t(g["shs"]), t(g["opacities"]), t(g["rotations"])
scales = t(g["scales"])
Hs = camera.circle_path(12, 0, 3, [90, 0])
sts = rp.settings_for_views(Hs, 512, 512, 45.0, dev, sh_degree=1, bg=torch.ones(3), super_sample_rate=2)
means2D = torch.zeros_like(means)
def loop(n):
    with torch.no_grad():
        for i in range(n):
            GaussianRasterizer(sts[i % 12])(means3D=means, means2D=means2D, shs=shs, colors_precomp=None, opacities=opac, scales=scales, rotations=rots, cov3D_precomp=None)
loop(24); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(96); t_host = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
_native.set_profiling(True); loop(24); torch.cuda.synchronize(); prof = _native.get_profile(); _native.set_profiling(False)
k = sum(ms for _, ms in prof) / 24
pr = cProfile.Profile(); pr.enable(); loop(48); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print(s.getvalue()[:3500])
print(json.dumps({"ms_per_call_wall": t_all / 96 * 1e3, "ms_per_call_host_enqueue": t_host / 96 * 1e3, "kernels_ms_per_call": k}))
