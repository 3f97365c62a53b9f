"""Where the HOST's time goes in the literal per-view loop (the reference caller's loop body: settings built per call,
GaussianRasterizer(settings)(...), loss, backward; one thread, one stream): cProfile over 96 frames, top functions by own time, the
wall time per frame next to the GPU's kernel time.  usage: python scripts/host_profile.py [frames]"""
import cProfile, os, pstats, sys, time, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import synth, camera, raster_passes as rp
from diff_gaussian_rasterization import GaussianRasterizer, _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda:0")
W, H = 1920, 1080
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
Hs = camera.circle_path(12, 0, 3, [90, 0])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
L = dict(means3D=t(g["means3D"]).requires_grad_(True), shs=t(g["shs"]).requires_grad_(True), opacities=t(g["opacities"]).requires_grad_(True),
         scales=t(g["scales"]).requires_grad_(True), rotations=t(g["rotations"]).requires_grad_(True))
L["means2D"] = torch.zeros_like(L["means3D"], requires_grad=True)
G = torch.rand(3, H, W, device=dev) * 2 - 1
bg = torch.ones(3, device=dev)
seg = {"settings": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0, "zero_grad": 0.0}


def frame(i, rec=False):
    t0 = time.perf_counter()
    st = rp.settings_for_view(Hs[i % 12], W, H, 45.0, dev, sh_degree=1, bg=bg, super_sample_rate=1)
    t1 = time.perf_counter()
    img, _ = GaussianRasterizer(st)(**L)
    t2 = time.perf_counter()
    loss = (img * G).sum()
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    for x in L.values():
        x.grad = None
    t5 = time.perf_counter()
    if rec:
        for k, d in zip(seg, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            seg[k] += d


for i in range(36):
    frame(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    frame(i, True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
print("literal loop: wall %.3f ms per frame = %.1f frames/s; host segments (ms per frame): %s" % (
    wall * 1e3, 1.0 / wall, {k: round(v / n * 1e3, 3) for k, v in seg.items()}))
# the same with the host NOT waiting inside forward for the pair count (how much of `forward` is that wait?)
N.set_profiling(True); N.get_profile()
for i in range(24):
    frame(i)
torch.cuda.synchronize()
ms = {}
for name, x in N.get_profile():
    ms.setdefault(name, []).append(x)
N.set_profiling(False)
print("GPU kernel time per frame (stage events): %.3f ms  %s" % (sum(float(np.mean(v)) for v in ms.values()), {k: round(float(np.mean(v)), 3) for k, v in ms.items()}))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    frame(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:60]))
