"""GPU idle / copy audit of a single-stream bench run from a rocprofv3 trace (--kernel-trace --memory-copy-trace --hip-trace,
csv): for the steady-state part of the run
  * the fraction of the wall time in which NO kernel was running (host stalls show up here), with the longest gaps and the
    kernels either side of them;
  * device->host copies per rasterizer call (the C ABI allows exactly one per forward: the 32-B counter block);
  * HIP API calls that block the host (hipStreamSynchronize, hipEventSynchronize, hipDeviceSynchronize, hipMemcpy) per call.
usage: python scripts/timeline.py <trace dir> [skip_fraction=0.5] [views_per_call=12]"""
import collections, csv, glob, os, sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
vpc = int(sys.argv[3]) if len(sys.argv) > 3 else 12


def rows_of(pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def short(n):
    return n.split("(")[0].replace("gsr::", "").replace("void ", "")[:44]


k = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows_of("*kernel_trace.csv"))
t0, t1 = k[0][0], max(r[1] for r in k)
cut = t0 + (t1 - t0) * skip
ks = [r for r in k if r[0] >= cut]
# align the window to whole calls: from the first k_preprocess to the last k_preprocess_backward
first = next(i for i, r in enumerate(ks) if r[2].startswith("k_preprocess<"))
last = max(i for i, r in enumerate(ks) if r[2].startswith("k_preprocess_backward"))
ks = ks[first:last + 1]
w0, w1 = ks[0][0], max(r[1] for r in ks)
busy = 0
gaps = []
cur_end = ks[0][0]
for i, (s, e, n) in enumerate(ks):
    if s > cur_end:
        gaps.append((s - cur_end, ks[i - 1][2], n))
    busy_from = max(s, cur_end)
    if e > busy_from:
        busy += e - busy_from
    cur_end = max(cur_end, e)
span = w1 - w0
calls = sum(1 for r in ks if r[2].startswith("k_preprocess<"))
ksum = sum(e - s for s, e, n in ks)
print("window: %.2f ms, %d rasterizer calls (%d views each), %d kernels" % (span / 1e6, calls, vpc, len(ks)))
print("GPU busy %.2f ms (%.1f%%), idle %.2f ms (%.1f%%); kernel-time sum %.2f ms; wall / kernel sum = %.3f" % (
    busy / 1e6, 100. * busy / span, (span - busy) / 1e6, 100. * (span - busy) / span, ksum / 1e6, span / ksum))
print("per frame: wall %.4f ms, kernels %.4f ms" % (span / 1e6 / calls / vpc, ksum / 1e6 / calls / vpc))
per_pair = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    per_pair[(a, b)][0] += g; per_pair[(a, b)][1] += 1
print("idle time by (previous kernel -> next kernel), us per call:")
for (a, b), (tot, cnt) in sorted(per_pair.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %8.1f  (%5.1f x %.1f us)  %s -> %s" % (tot / 1e3 / calls, cnt / calls, tot / cnt / 1e3, a, b))
mc = [r for r in rows_of("*memory_copy_trace.csv") if w0 <= int(r["Start_Timestamp"]) <= w1]
kinds = collections.Counter(r.get("Direction", r.get("Kind", "?")) for r in mc)
print("memory copies in the window:", {kk: "%d (%.2f per call)" % (v, v / calls) for kk, v in kinds.items()})
api = [r for r in rows_of("*hip_api_trace.csv") if w0 <= int(r["Start_Timestamp"]) <= w1]
blocking = collections.Counter()
btime = collections.Counter()
for r in api:
    f = r["Function"]
    if f in ("hipStreamSynchronize", "hipEventSynchronize", "hipDeviceSynchronize", "hipMemcpy", "hipMemcpyDtoH", "hipStreamWaitEvent") or f.startswith("hipMemcpy"):
        blocking[f] += 1
        btime[f] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("host-side sync / copy API calls per call:", {f: "%.2f (%.0f us)" % (c / calls, btime[f] / 1e3 / calls) for f, c in blocking.items()})
launches = sum(1 for r in api if "LaunchKernel" in r["Function"] or r["Function"] == "hipModuleLaunchKernel")
print("kernel launches per call: %.1f" % (launches / calls))
