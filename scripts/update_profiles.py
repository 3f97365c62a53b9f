"""Copy the judged artefacts of a scripts/profile_gpu.sh run from gpurun_out/prof_<tag>/ into profiles/ and rebuild
profiles/pmc_traffic.json (HBM-side bytes per launch per kernel from the FETCH_SIZE / WRITE_SIZE passes).
usage: python scripts/update_profiles.py [tag [views_per_launch]]"""
import csv, glob, json, os, shutil, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
views_per_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 12     # bench.py --views-per-call of the profiled command
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, "%s_bench_rocprofv3_summary.txt" % tag))
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "%s_bench_kernel_stats.csv" % tag))


def short(n):
    return n.split("(")[0].replace("gsr::", "").replace("void ", "")


raw = collections.defaultdict(dict)
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = collections.defaultdict(float); launches = collections.defaultdict(set)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            acc[k] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
    for k in acc:
        raw[k][counter + "_KiB"] = acc[k] / max(len(launches[k]), 1)
out = {
    "views_per_launch": views_per_launch,
    "source": "profiles/%s_bench_rocprofv3_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --streams 1)" % tag,
    "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch; x2 is the gfx950 FETCH_SIZE half-count correction of "
               "MI355X_MICROARCH.md (calibrated there for wide coalesced reads; the render kernels' 16-B gathers are not "
               "separately calibrated, so this is an upper bound for them); FETCH_SIZE counts fabric requests, "
               "Infinity-Cache hits included",
    "bytes_per_launch": {k: int((2 * v.get("FETCH_SIZE_KiB", 0.0) + v.get("WRITE_SIZE_KiB", 0.0)) * 1024) for k, v in raw.items()
                         if k.startswith("k_")},
    "raw": {k: v for k, v in raw.items() if k.startswith("k_")},
}
sq = collections.defaultdict(lambda: collections.defaultdict(float)); sql = collections.defaultdict(set)
for f in glob.glob(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith("k_render"):
            sq[k][r["Counter_Name"]] += float(r["Counter_Value"]); sql[k].add(r["Dispatch_Id"])
out["valu_wave_instructions_per_launch"] = {k: int(v["SQ_INSTS_VALU"] / max(len(sql[k]), 1)) for k, v in sq.items() if "SQ_INSTS_VALU" in v}
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out["bytes_per_launch"], indent=1, sort_keys=True))
print(out["valu_wave_instructions_per_launch"])
