"""Instrumentation run (GSR_EXTRA_FLAGS=-DGSR_STATS): where the waves of a single-view forward render ran (XCD / SE / CU / SIMD from
HW_ID) and for how long -- is the launch's duration the load of its busiest SIMD?"""
import ctypes as C, os, sys, collections
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
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = TB._batch_args(g, views[:V], 1920, 1080, dev)
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
ran = r[:, 4] == 1
r = r[ran]
idx = np.nonzero(ran)[0]
life = r[:, 0].astype(np.float64) * 0.01          # us
start = (r[:, 9] - r[:, 9].min()).astype(np.float64) * 0.01
hw = r[:, 8]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = hw >> 28
key = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
work = life > 2.0
print("kernel %.1f us; %d waves ran, %d of them longer than 2 us; distinct (xcc, se, sh, cu, simd): %d" % (prof["render_forward"] * 1e3, len(r), work.sum(), len(np.unique(key))))
print("xcc values", np.unique(xcc), "se", np.unique(se), "sh", np.unique(sh), "cu", np.unique(cu), "simd", np.unique(simd))
# busy time per SIMD = union of its waves' lifetimes is what matters; the sum of lifetimes says how many waves overlapped
per = collections.defaultdict(list)
for k, s0, l in zip(key[work], start[work], life[work]):
    per[k].append((s0, s0 + l))
ends = np.array([max(e for _, e in v) for v in per.values()])
sums = np.array([sum(e - s0 for s0, e in v) for v in per.values()])
cnts = np.array([len(v) for v in per.values()])
print("per SIMD: working waves min / mean / max %d / %.1f / %d; summed wave life mean %.0f us, max %.0f us; last wave ends: mean %.0f us, p90 %.0f, max %.0f"
      % (cnts.min(), cnts.mean(), cnts.max(), sums.mean(), sums.max(), ends.mean(), np.percentile(ends, 90), ends.max()))
# how does the dispatcher place consecutive workgroups?  first 64 working workgroups: blockIdx -> (xcc, se, cu, simd), start
o = np.argsort(idx[work])[:48]
for j in o:
    i = np.nonzero(work)[0][j]
    print("  wg %6d  xcc %d se %d sh %d cu %2d simd %d  start %7.1f us life %7.1f us" % (idx[i], xcc[i], se[i], sh[i], cu[i], simd[i], start[i], life[i]))
# the 12 longest-lived waves: when did they start, what else ran on their SIMD
top = np.argsort(-life)[:12]
for i in top:
    mates = [(round(s0), round(e)) for (s0, e) in per[key[i]]]
    print("  long wave wg %6d life %6.1f start %6.1f  rounds %d pairs %d; us waiting %.1f staging %.1f evaluating %.1f; on its SIMD: %d working waves %s"
          % (idx[i], life[i], start[i], r[i, 5], r[i, 6], r[i, 1] * 0.01, r[i, 2] * 0.01, r[i, 3] * 0.01, len(mates), sorted(mates)[:6]))
print("raw XCC_ID / HW_ID samples:", [hex(int(x)) for x in hw[:8]])
tot_pairs = r[:, 6].astype(np.float64); tot_rounds = r[:, 5].astype(np.float64)
print("all waves: rounds %.0f pairs %.0f; us per pair (eval / pairs) %.3f; us per round staging %.3f" % (tot_rounds.sum(), tot_pairs.sum(), r[:, 3].sum() * 0.01 / max(tot_pairs.sum(), 1), r[:, 2].sum() * 0.01 / max(tot_rounds.sum(), 1)))
