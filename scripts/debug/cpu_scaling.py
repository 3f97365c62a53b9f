"""debug: CPU oracle frame time (headline frame, forward+backward) against the thread count, on this box's host cores"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np
from oracle.oracle import Oracle, Scene
from pcrender import camera, synth
W, H = 1920, 1080
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
v0 = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)[0]
sc = Scene(W=W, H=H, tanfovx=v0["tanfovx"], tanfovy=v0["tanfovy"], bg=np.ones(3, np.float32), means3D=g["means3D"],
           opacities=g["opacities"], viewmatrix=v0["viewmatrix"].numpy(), projmatrix=v0["projmatrix"].numpy(),
           campos=v0["campos"].numpy(), shs=g["shs"], scales=g["scales"], rotations=g["rotations"], sh_degree=1)
G = np.random.default_rng(123).uniform(-1, 1, (3, H, W)).astype(np.float32)
o = Oracle()
print("cpu_count", os.cpu_count())
for nt in [int(x) for x in (sys.argv[1:] or ["256", "128", "64", "32", "16", "1"])]:
    ts = []
    for _ in range(3 if nt > 1 else 1):
        t = time.perf_counter(); o.forward(sc, nthreads=nt); t1 = time.perf_counter(); o.forward_backward(sc, G, nthreads=nt); t2 = time.perf_counter()
        ts.append((t1 - t, t2 - t1))
    print("%4d threads: forward %.3f s, forward+backward %.3f s" % (nt, min(a for a, b in ts), min(b for a, b in ts)))
