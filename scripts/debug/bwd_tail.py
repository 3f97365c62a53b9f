"""Instrumentation run (GSR_EXTRA_FLAGS=-DGSR_STATS): render backward at 1 and 12 views per call -- kernel time against the summed wave
time per wave slot and the longest-lived wave (is the single-view launch a packing problem like the forward's?)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-pcloud-render_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from pcrender import camera, synth
from diff_gaussian_rasterization import _native as N
import test_gpu_batch as TB
dev = torch.device("cuda:0")
cloud = synth.make_cloud("synth-THuman-800K", seed=0)
g = synth.make_gaussians(cloud, profile="training", seed=1)
views = camera.circle_views(12, fov_deg=45.0, width_px=1920, height_px=1080)
out = (C.c_ulonglong * 8)()
for V in (1, 12):
    args = TB._batch_args(g, views[:V], 1920, 1080, dev)
    G = torch.rand((V, 3, 1080, 1920), device=dev)
    def once():
        r = N.rasterize_gaussians_batch(*args, need_backward=True)
        counts, color, radii, geom, binning, img = r
        bg, means3D, colors, opacity, scales, rotations, sm, cov, vm, pm, tx, ty, H, W, sh, deg, cam, pref, dbg = args
        N.rasterize_gaussians_backward_batch(bg, means3D, radii, colors, scales, rotations, sm, cov, vm, pm, tx, ty, G, sh, deg, cam, geom, binning, img, False)
    once(); once(); torch.cuda.synchronize()
    N.lib.gsr_debug_bwd_times(out, 1)
    N.set_profiling(True); once(); torch.cuda.synchronize()
    prof = dict(N.get_profile()); N.set_profiling(False)
    N.lib.gsr_debug_bwd_times(out, 0)
    life, wait, setup, stage, ev, longest, groups, iw = [int(out[i]) for i in range(8)]
    items, waves = iw >> 20, iw & ((1 << 20) - 1)
    k = prof["render_backward"]
    print("V=%d: kernel %.3f ms; %d waves ran, %d items; summed wave time %.1f ms = %.3f ms per slot of 5120; longest wave %.3f ms; wait %.1f setup %.1f stage %.1f eval %.1f ms"
          % (V, k, waves, items, life * 1e-5, life * 1e-5 / 5120, longest * 1e-5, wait * 1e-5, setup * 1e-5, stage * 1e-5, ev * 1e-5))
    # timeline of the launch: how many workgroups (= waves) are alive over time, when the longest ones started
    n = 1 << 17
    buf = (C.c_uint * (n * 8))()
    N.lib.gsr_debug_bwd_records.argtypes = [C.POINTER(C.c_uint), C.c_int]
    got = N.lib.gsr_debug_bwd_records(buf, n)
    r = np.frombuffer(buf, dtype=np.uint32).reshape(n, 8)[:got]
    r = r[r[:, 0] != 0]
    life = r[:, 0].astype(np.float64) * 0.01
    start = (r[:, 5] - r[:, 5].min()).astype(np.uint32).astype(np.float64) * 0.01
    end = start + life
    step = k * 1e3 / 10
    print("   alive at t = " + ", ".join("%.0f us: %d" % (t, ((start <= t) & (end > t)).sum()) for t in np.arange(0.5, 10.5) * step))
    print("   not yet started at those times: " + ", ".join("%d" % (start > t).sum() for t in np.arange(0.5, 10.5) * step))
    top = np.argsort(-end)[:6]
    for i in top:
        print("   ends last: start %.1f life %.1f end %.1f us; groups %d items %d; eval %.1f stage %.1f setup %.1f us" % (start[i], life[i], end[i], r[i, 6], r[i, 7], r[i, 4] * 0.01, r[i, 3] * 0.01, r[i, 2] * 0.01))
    print("   life percentiles 50 / 90 / 99 / max: %s us; groups per wave 50 / 90 / 99 / max: %s" % (" / ".join("%.1f" % np.percentile(life, q) for q in (50, 90, 99, 100)), " / ".join("%d" % np.percentile(r[:, 6], q) for q in (50, 90, 99, 100))))
