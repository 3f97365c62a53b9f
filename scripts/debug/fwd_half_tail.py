"""Instrumentation run (GSR_EXTRA_FLAGS=-DGSR_STATS): the waves of a SINGLE-VIEW forward render (half-quadrant kernel): how many are
alive over time, and what the longest-lived ones did (rounds, steps of four entries, list length; time waiting / staging /
evaluating)."""
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
vi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 1        # views in the submission (1: half-quadrant kernel; more: the 8 x 8 kernel, steps = pairs)
args = TB._batch_args(g, (views + views)[vi:vi + nv], 1920, 1080, dev)
for _ in range(2):
    N.rasterize_gaussians_batch(*args, need_backward=True)
torch.cuda.synchronize()
out8 = (C.c_ulonglong * 8)()
N.lib.gsr_debug_fwd_times(out8, 1)
N.set_profiling(True)
N.rasterize_gaussians_batch(*args, need_backward=True)
torch.cuda.synchronize()
prof = dict(N.get_profile()); N.set_profiling(False)
n = 1 << 19
buf = (C.c_uint * (n * 10))()
N.lib.gsr_debug_fwd_records.argtypes = [C.POINTER(C.c_uint), C.c_int]
got = N.lib.gsr_debug_fwd_records(buf, n)
r = np.frombuffer(buf, dtype=np.uint32).reshape(n, 10)[:got]
r = r[r[:, 4] == 1]
life = r[:, 0].astype(np.float64) * 0.01          # us
start = (r[:, 9] - r[:, 9].min()).astype(np.float64) * 0.01
end = start + life
k = prof["render_forward"] * 1e3
print("view %d x%d: kernel %.1f us; %d waves recorded; summed wave life %.1f ms; waves longer than 2 us: %d" % (vi, nv, k, len(r), life.sum() * 1e-3, (life > 2).sum()))
for t in ((5, 10, 20, 30, 50, 75, 100, 125, 150, 175, 200, 220) if nv == 1 else [k * f for f in (0.02, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95, 0.98)]):
    alive = ((start <= t) & (end > t)).sum()
    print("  t = %3d us: %5d waves alive (%.0f %% of 5 120 slots at 5 per SIMD), %5d not yet started" % (t, alive, 100.0 * alive / 5120, (start > t).sum()))
print("rounds: total %d, steps: total %d; wave life percentiles (us) 50 / 90 / 99 / 99.9 / max: %s" % (
    r[:, 5].sum(), r[:, 6].sum(), " / ".join("%.1f" % np.percentile(life, q) for q in (50, 90, 99, 99.9, 100))))
top = np.argsort(-end)[:14]
print("the 14 waves that END last:")
for i in top:
    print("  start %6.1f life %6.1f end %6.1f us | list %6d entries, rounds %4d, steps %4d | waiting %.1f staging+cull %.1f evaluating %.1f us | %.2f us per step, %.2f us per round"
          % (start[i], life[i], end[i], r[i, 7], r[i, 5], r[i, 6], r[i, 1] * 0.01, r[i, 2] * 0.01, r[i, 3] * 0.01,
             r[i, 3] * 0.01 / max(1, r[i, 6]), life[i] / max(1, r[i, 5])))
top = np.argsort(-life)[:8]
print("the 8 longest-lived waves:")
for i in top:
    print("  start %6.1f life %6.1f end %6.1f us | list %6d entries, rounds %4d, steps %4d | waiting %.1f staging+cull %.1f evaluating %.1f us"
          % (start[i], life[i], end[i], r[i, 7], r[i, 5], r[i, 6], r[i, 1] * 0.01, r[i, 2] * 0.01, r[i, 3] * 0.01))
# work against list length: what would a better launch order key on?
L = r[:, 7].astype(np.int64)
steps = r[:, 6].astype(np.int64)
rounds = r[:, 5].astype(np.int64)
print("list length bucket | waves | steps mean / p90 / max | rounds mean | life mean / max us | start mean us")
edges = [1, 64, 128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (L >= lo) & (L < hi)
    if m.sum() == 0:
        continue
    print("  %6d .. %6d | %5d | %6.1f / %6.1f / %4d | %6.1f | %6.1f / %6.1f | %6.1f" % (
        lo, hi - 1, m.sum(), steps[m].mean(), np.percentile(steps[m], 90), steps[m].max(), rounds[m].mean(), life[m].mean(), life[m].max(), start[m].mean()))
os.makedirs(os.path.join(ROOT, "gpurun_out", "dbg"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "dbg", "waves_v%d_%d.npy" % (vi, nv)), np.stack([L, steps, rounds, (life * 100).astype(np.int64), (start * 100).astype(np.int64)], 1).astype(np.int32))
