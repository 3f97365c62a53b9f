"""cProfile of the library's Python path around the C calls (forward and backward of the per-view API) with the C entry points
stubbed out, so only host-side Python / torch bookkeeping is timed: what the GPU waits for before the first kernel of a call."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np, torch
from pcrender import raster_passes as rp, camera, synth
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizer, _native
dev = torch.device("cuda:0")
W, H = 1920, 1080
cloud = synth.make_cloud("synth-THuman-256", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
D = g["sh_degree"]
leaf = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)  # noqa: E731
m3 = leaf(g["means3D"])
L = dict(means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), shs=leaf(g["shs"]), opacities=leaf(g["opacities"]),
         scales=leaf(g["scales"]), rotations=leaf(g["rotations"]))
G = torch.zeros((3, H, W), device=dev)
Hs = camera.circle_path(12, 0, 3, [90, 0])
sts = rp.settings_for_views(Hs, W, H, 45.0, dev, sh_degree=D, bg=torch.ones(3, device=dev), super_sample_rate=1)
# one real call (capacity hint), then stubs
img, _ = GaussianRasterizer(sts[0])(**L)
img.sum().backward()
torch.cuda.synchronize()
real_f, real_b, real_l = _native.lib.gsr_forward_batch, _native.lib.gsr_backward_batch, _native.lib.gsr_last_list_pairs
_native.lib.gsr_forward_batch = lambda *a: 0
_native.lib.gsr_backward_batch = lambda *a: 0
def fake_pairs(arr, V):
    arr[0] = 1000000
    return 0
_native.lib.gsr_last_list_pairs = fake_pairs


def fwd_only(n):
    for i in range(n):
        with torch.no_grad():
            GaussianRasterizer(sts[i % 12])(**L)


def fwd_bwd(n):
    for i in range(n):
        img, _ = GaussianRasterizer(sts[i % 12])(**L)
        img.backward(G)
        for t in L.values():
            t.grad = None


for fn, n in ((fwd_only, 2000), (fwd_bwd, 1000)):
    fn(100)
    t0 = time.perf_counter(); fn(n); dt = (time.perf_counter() - t0) / n
    print("%s: %.1f us per call (C entry points stubbed)" % (fn.__name__, dt * 1e6))
    pr = cProfile.Profile(); pr.enable(); fn(n); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:40]))
