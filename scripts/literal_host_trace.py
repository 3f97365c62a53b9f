"""Where the host's time goes in the reference caller's literal loop (settings built per call, GaussianRasterizer(settings)(...),
loss.backward(); synth-THuman-800K, 1920x1080) -- perf_counter stamps at the boundaries of the library's Python and C layers,
averaged over the steady state, next to the frame time and the kernels' own time.  What runs between the caller's last blocking
host->device copy and the first kernel launch is time the GPU idles in (the copies drain the stream every frame).
usage: python scripts/literal_host_trace.py [frames=240] [pkg_dir=gaussian-pcloud-render_amd]   (pkg_dir: another copy of the
Python package, for A/B runs in one lease)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 240
PKG = sys.argv[2] if len(sys.argv) > 2 else "gaussian-pcloud-render_amd"
PKG = PKG if os.path.isabs(PKG) else os.path.join(ROOT, PKG)
if PKG != os.path.join(ROOT, "gaussian-pcloud-render_amd"):
    os.environ.setdefault("GSR_LIB", os.path.join(ROOT, "gaussian-pcloud-render_amd", "diff_gaussian_rasterization", "libgsr_hip.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), PKG]
import numpy as np, torch
from pcrender import raster_passes as rp, camera, synth
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizer, _native

dev = torch.device("cuda:0")
W, H = 1920, 1080
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
D = g["sh_degree"]
leaf = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)  # noqa: E731
m3 = leaf(g["means3D"])
L = dict(means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), shs=leaf(g["shs"]), opacities=leaf(g["opacities"]),
         scales=leaf(g["scales"]), rotations=leaf(g["rotations"]))
G = torch.from_numpy(np.random.default_rng(123).uniform(-1, 1, (3, H, W)).astype(np.float32)).to(dev)
Hs = camera.circle_path(12, 0, 3, [90, 0])
bg = torch.ones(3, device=dev)

now = time.perf_counter
stamps = {}


def stamp(k):
    stamps.setdefault(k, []).append(now())


lib = _native.lib


class Wrapped:
    """times a ctypes entry point from the Python side"""

    def __init__(self, fn, name):
        self.fn, self.name = fn, name

    def __call__(self, *a):
        stamp(self.name + "_in")
        r = self.fn(*a)
        stamp(self.name + "_out")
        return r


lib.gsr_forward_batch = Wrapped(lib.gsr_forward_batch, "cfwd")
lib.gsr_backward_batch = Wrapped(lib.gsr_backward_batch, "cbwd")
_f0, _b0 = dgr._RasterizeGaussians.forward, dgr._RasterizeGaussians.backward


def fwd(ctx, *a):
    stamp("pyfwd_in")
    r = _f0(ctx, *a)
    stamp("pyfwd_out")
    return r


def bwd(ctx, *a):
    stamp("pybwd_in")
    r = _b0(ctx, *a)
    stamp("pybwd_out")
    return r


dgr._RasterizeGaussians.forward = staticmethod(fwd)
dgr._RasterizeGaussians.backward = staticmethod(bwd)


def frame(i):
    stamp("frame_in")
    st = rp.settings_for_view(Hs[i % 12], W, H, 45.0, dev, sh_degree=D, bg=bg, super_sample_rate=1)
    stamp("settings_out")
    img, _ = GaussianRasterizer(st)(**L)
    stamp("call_out")
    (img * G).sum().backward()
    stamp("backward_out")
    for t in L.values():
        t.grad = None
    stamp("frame_out")


for i in range(36):
    frame(i)
torch.cuda.synchronize()
stamps.clear()
t0 = now()
for i in range(FRAMES):
    frame(i)
torch.cuda.synchronize()
wall = (now() - t0) / FRAMES
_native.set_profiling(True)
st_keep = dict(stamps)
for i in range(24):
    frame(i)
torch.cuda.synchronize()
prof = _native.get_profile()
_native.set_profiling(False)
ksum = sum(ms for _, ms in prof) / 24
s = {k: np.array(v[:FRAMES]) for k, v in st_keep.items()}
us = lambda a, b: float(np.mean(s[b] - s[a]) * 1e6)  # noqa: E731
out = {
    "pkg": os.path.relpath(PKG, ROOT), "frames": FRAMES, "frame_us": wall * 1e6, "kernels_us": ksum * 1e3,
    "host_us": {
        "caller: settings built (3 blocking copies)": us("frame_in", "settings_out"),
        "GaussianRasterizer(st) + Module.__call__ + autograd apply -> Function.forward": us("settings_out", "pyfwd_in"),
        "Function.forward -> C call (the library's Python preamble)": us("pyfwd_in", "cfwd_in"),
        "C forward (launches + the wait for num_rendered)": us("cfwd_in", "cfwd_out"),
        "C return -> Function.forward returns": us("cfwd_out", "pyfwd_out"),
        "-> caller has (img, radii)": us("pyfwd_out", "call_out"),
        "loss + autograd until Function.backward": us("call_out", "pybwd_in"),
        "Function.backward -> C call": us("pybwd_in", "cbwd_in"),
        "C backward": us("cbwd_in", "cbwd_out"),
        "C return -> backward() returns to the caller": us("cbwd_out", "backward_out"),
        "grads dropped": us("backward_out", "frame_out"),
        "whole frame on the host": us("frame_in", "frame_out"),
    },
    "exposed_us (settings_out -> first launch ~ cfwd_in)": us("settings_out", "cfwd_in"),
}
print(json.dumps(out, indent=1))
