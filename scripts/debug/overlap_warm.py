"""How long does the per-view loop with the library's overlap of consecutive calls take to reach its steady rate, and is the
caching allocator still growing the side streams' pools meanwhile?  (blocks of 48 frames, one caller thread)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-pcloud-render_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _native
from pcrender import camera, synth
dev = torch.device("cuda:0")
W, H = 1920, 1080
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
views = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)
bg = torch.ones(3, device=dev)
rs = [GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=v["tanfovx"], tanfovy=v["tanfovy"], bg=bg,
      scale_modifier=1.0, viewmatrix=v["viewmatrix"].to(dev), projmatrix=v["projmatrix"].to(dev), sh_degree=g["sh_degree"],
      campos=v["campos"].to(dev), prefiltered=False, debug=False)) for v in views]
leaf = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)  # noqa: E731
m3 = leaf(g["means3D"])
L = dict(means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), shs=leaf(g["shs"]), opacities=leaf(g["opacities"]),
         scales=leaf(g["scales"]), rotations=leaf(g["rotations"]))
G = torch.from_numpy(np.random.default_rng(123).uniform(-1, 1, (3, H, W)).astype(np.float32)).to(dev)
for _ in range(int(os.environ.get("PRE_STREAMS", "0"))):     # shifts which hardware queues the side streams land on
    torch.cuda.Stream(device=dev)
for on in (True, False):
    _native.set_overlap(on)
    print("overlap", on, _native.OVERLAP_STATS)
    for blk in range(5):
        st0 = torch.cuda.memory_stats(dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(48):
            img, _ = rs[i % 12](**L)
            (img * G).sum().backward()
            for t in L.values():
                t.grad = None
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st1 = torch.cuda.memory_stats(dev)
        print("  block %d: %7.1f frames/s   device allocations %d  frees %d  reserved %.1f GB" % (
            blk, 48 / dt, st1["num_device_alloc"] - st0["num_device_alloc"], st1["num_device_free"] - st0["num_device_free"],
            st1["reserved_bytes.all.current"] / 2**30))
