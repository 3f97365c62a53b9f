"""How the CPU baseline (plain-C oracle, OpenMP) scales with threads on this host, under different OpenMP placement settings:
seconds per frame (forward + backward, view 0 of synth-THuman-800K at 1080p).  Each setting runs in a child process (libgomp
reads its environment once).  usage: python scripts/cpu_scaling_probe.py [child threads...]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import util
    from oracle.oracle import Oracle
    from pcrender import camera, synth
    W, H = 1920, 1080
    cloud = synth.make_cloud("synth-THuman-800K", seed=0)
    g = synth.make_gaussians(cloud, profile="training", seed=1)
    view = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)[0]
    s = util.scene_from(g, view, W, H, bg=(1, 1, 1))
    G = np.random.default_rng(123).uniform(-1, 1, (3, H, W)).astype(np.float32)
    o = Oracle()
    out = {}
    for nt in [int(x) for x in sys.argv[2:]]:
        o.forward_backward(s, G, nthreads=nt)
        ts = []
        for _ in range(2):
            t = time.perf_counter(); o.forward_backward(s, G, nthreads=nt); ts.append(time.perf_counter() - t)
        out[nt] = round(min(ts), 3)
    print(json.dumps(out))
    sys.exit(0)
print("os.cpu_count() =", os.cpu_count(), " sched_getaffinity =", len(os.sched_getaffinity(0)))
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip(), "(cgroup cpu.max)")
except OSError:
    pass
os.system("lscpu | grep -E 'Socket|Core|Thread|NUMA node\\(s\\)|Model name' | head -6")
for name, env in (("default", {}), ("OMP_PROC_BIND=close OMP_PLACES=cores", {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}),
                  ("OMP_PROC_BIND=spread OMP_PLACES=cores", {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"}),
                  ("OMP_WAIT_POLICY=active", {"OMP_WAIT_POLICY": "active"})):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", "16", "32", "64", "128"], capture_output=True, text=True,
                       env=dict(os.environ, **env))
    print("%-40s s/frame by threads: %s" % (name, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]))
