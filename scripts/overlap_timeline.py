"""Concurrency analysis of a rocprofv3 --kernel-trace CSV of a multi-stream bench run: how much of the wall time has
0 / 1 / 2 / ... kernels in flight, and how long each kernel type takes when frames overlap.
usage: python scripts/overlap_timeline.py <dir with *kernel_trace.csv> [skip_fraction]"""
import collections, csv, glob, os, sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("gsr::", "").replace("void ", ""),
                     r.get("Queue_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + (t1 - t0) * skip          # analyse the steady state only
rows = [r for r in rows if r[0] >= cut]
ev = []
for s, e, n, q in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = collections.Counter(); cur = 0; last = ev[0][0]
for t, dlt in ev:
    hist[cur] += t - last; last = t; cur += dlt
span = ev[-1][0] - ev[0][0]
print("steady-state span %.2f ms, %d kernels, queues: %s" % (span / 1e6, len(rows), sorted(set(r[3] for r in rows))))
for k in sorted(hist):
    print("  %d kernels in flight: %5.1f%% of the time" % (k, 100.0 * hist[k] / span))
dur = collections.defaultdict(list)
for s, e, n, q in rows:
    dur[n].append(e - s)
print("%-28s %6s %10s %10s" % ("kernel", "calls", "avg_us", "total_ms"))
for n in sorted(dur, key=lambda n: -sum(dur[n]))[:14]:
    print("%-28s %6d %10.1f %10.2f" % (n[:28], len(dur[n]), sum(dur[n]) / len(dur[n]) / 1e3, sum(dur[n]) / 1e6))
# (by prefix: the template arguments differ between rounds and call shapes -- k_render_backward<2>, k_render_forward_half)
def launches(prefix):
    return sum(len(v) for n, v in dur.items() if n.startswith(prefix))
frames = launches("k_render_backward") or launches("k_render_forward")
print("frames in window: %d -> %.3f ms/frame" % (frames, span / 1e6 / max(frames, 1)))
