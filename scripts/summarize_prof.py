"""Summarise rocprofv3 output of scripts/profile_gpu.sh: per-kernel time (kernel-trace stats) and per-kernel
FETCH_SIZE / WRITE_SIZE / SQ counters (PMC passes), averaged per launch."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def short(n):
    n = n.split("(")[0]
    return n.replace("gsr::", "").replace("void ", "")
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
print("== kernel-trace stats (rocprofv3 --kernel-trace --stats) ==")
for f in stats:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("TotalDurationNs".lower(), 0)) or 0))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("%-44s %8s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:24]:
        print("%-44s %8s %12.1f %12.2f %6.1f%%" % (short(r["Name"])[:44], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                  float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    files = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("== %s: no counter_collection.csv ==" % tag); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); c = r["Counter_Name"]; v = float(r["Counter_Value"])
            acc[k][c] += v
            key = (k, r.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key); cnt[k] += 1
    print("== %s (per launch averages) ==" % tag)
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        print("%-44s launches=%5d  " % (k[:44], cnt[k]) + "  ".join("%s=%.4g" % (c, acc[k][c] / max(cnt[k], 1)) for c in sorted(acc[k])))
