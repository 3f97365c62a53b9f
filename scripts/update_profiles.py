"""Copy the judged artefacts of a scripts/profile_gpu.sh run from gpurun_out/prof_<tag>/ into profiles/ and rebuild
profiles/pmc_traffic.json: per kernel and per launch, the HBM-side bytes from the FETCH_SIZE / WRITE_SIZE passes, the VALU
instruction count, and the kernel's average duration in the kernel trace OF THE SAME LEASE, keyed by the workload and call
shape of the profiled command (bench.py prints `traffic` only for a run with the same key).
usage: python scripts/update_profiles.py [tag]"""
import csv, glob, json, os, shutil, sys, collections, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, "%s_bench_rocprofv3_summary.txt" % tag))
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "%s_bench_kernel_stats.csv" % tag))
if os.path.exists(os.path.join(src, "timeline.txt")):
    shutil.copy(os.path.join(src, "timeline.txt"), os.path.join(dst, "%s_timeline.txt" % tag))
if os.path.exists(os.path.join(src, "fetch_calibration.txt")):
    shutil.copy(os.path.join(src, "fetch_calibration.txt"), os.path.join(dst, "%s_fetch_calibration.txt" % tag))
for extra in ("bench_driver_shape", "bench_config1", "bench_config3", "bench_config4"):
    f = os.path.join(src, extra + ".json")
    if os.path.exists(f):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(dst, "%s_%s.json" % (tag, extra)), "w"), indent=1)
bench = json.loads([l for l in open(os.path.join(src, "bench.json")).read().splitlines() if l.startswith("{")][-1])
json.dump(bench, open(os.path.join(dst, "%s_bench_line.json" % tag), "w"), indent=1)
wl = bench["config"]["workload"]
W, H = [int(x) for x in wl.split()[1].split("x")]
key = {"workload": wl.split()[0], "points": bench["config"]["points"], "width": W, "height": H,
       "views_per_launch": bench["views_per_call"], "profile": "training" if "training profile" in wl else "inference",
       "forward_only": "fwd+bwd" not in wl}


def short(n):
    return n.split("(")[0].replace("gsr::", "").replace("void ", "")


# FETCH_SIZE factor per access pattern, from the calibration probe of the same lease (bytes of the 64-B lines touched /
# (FETCH_SIZE x 1024)): wide streaming kernels 2.0, the record gathers of the render kernels and the pair emission by k_gather*
factor = {"default": 2.0}
cal = os.path.join(src, "fetch_calibration.txt")
if os.path.exists(cal):
    for l in open(cal):
        p = l.split()
        if len(p) >= 9 and p[0] in ("k_stream16", "k_gather16", "k_gather36") and float(p[3]) > 0:
            factor["_" + p[0]] = round(float(p[2]) / (float(p[3]) * 1024.0), 3)      # line bytes per counted byte
    if "_k_stream16" in factor:
        factor["default"] = factor["_k_stream16"]
    for k in ("k_render_backward<0>", "k_render_backward<1>", "k_render_backward<2>", "k_render_forward<0>", "k_render_forward_half"):
        if "_k_gather36" in factor:
            factor[k] = factor["_k_gather36"]
    if "_k_gather16" in factor:
        factor["k_duplicate<unsigned short>"] = factor["_k_gather16"]

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
avg_us = {}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k.startswith("k_"):
            avg_us[k] = round(float(r["AverageNs"]) / 1e3, 3)
# the shader clock the profiled (kernel-trace) run sustained: its own bench line, printed under rocprofv3
prof_clk = None
try:
    tl = [l for l in open(os.path.join(src, "bench_trace.log")).read().splitlines() if l.startswith("{")]
    if tl:
        prof_clk = (json.loads(tl[-1]).get("sclk_mhz") or {}).get("timed_region")
except OSError:
    pass
for name in ("bench_config1", "bench_config3", "bench_config4"):
    f = os.path.join(src, name + ".json")
    if os.path.exists(f):
        ls = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if ls:
            json.dump(json.loads(ls[-1]), open(os.path.join(dst, "%s_%s.json" % (tag, name)), "w"), indent=1)
out = {
    "key": key,
    # which kernels the counters belong to: bench.py quotes them only for a tree with the same kernel sources
    "kernels_sha": bench.get("kernels_sha"),
    "lease": "gpurun_out/prof_%s (%s)" % (tag, time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(os.path.join(src, "bench.json"))))),
    "sclk_mhz": prof_clk,
    "source": "profiles/%s_bench_rocprofv3_summary.txt: rocprofv3 kernel trace + separate --pmc FETCH_SIZE / WRITE_SIZE / SQ passes of "
              "`bench.py --streams 1` and the bench line profiles/%s_bench_line.json, all in one gpurun lease" % (tag, tag),
    "formula": "(fetch_factor x FETCH_SIZE + WRITE_SIZE) x 1024 bytes per launch; fetch_factor from profiles/%s_fetch_calibration.txt "
               "(same lease): 64-B lines touched per counted byte for the kernel's access pattern (wide streaming read: the gfx950 "
               "half count of MI355X_MICROARCH.md; 36-B-of-a-line record gathers for the render kernels); FETCH_SIZE counts fabric "
               "requests, Infinity-Cache hits included" % tag,
    "fetch_factor": factor,
    "bytes_per_launch": {k: int((factor.get(k, factor["default"]) * v.get("FETCH_SIZE_KiB", 0.0) + v.get("WRITE_SIZE_KiB", 0.0)) * 1024)
                         for k, v in raw.items() if k.startswith("k_")},
    "avg_us": avg_us,
    "raw": {k: v for k, v in raw.items() if k.startswith("k_")},
}
sq = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ("pmc_sq", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_render"):
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
    for k in acc:
        for c, v in acc[k].items():
            if c == "GRBM_GUI_ACTIVE" and sub == "pmc_sq2" and "GRBM_GUI_ACTIVE" in sq[k]:
                continue
            sq[k][c] = v / max(len(launches[k]), 1)
out["valu_wave_instructions_per_launch"] = {k: int(v["SQ_INSTS_VALU"]) for k, v in sq.items() if "SQ_INSTS_VALU" in v}
# How busy the SIMDs are: SQ_ACTIVE_INST_VALU counts, summed over every SIMD of the chip, the QUAD-cycles a wave's
# VALU instruction (matrix instructions included) occupies its SIMD (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles), so
#   valu_busy = 4 x SQ_ACTIVE_INST_VALU / (SIMDs x cycles of the launch),  SIMDs = 256 CUs x 4,
# with the launch's cycles from GRBM_GUI_ACTIVE of the same pass (summed over the 8 XCDs by the profiler: / 8; cross-checked against
# duration x shader clock of the kernel trace).  1.0 = every SIMD executing a vector instruction in every cycle of the launch.
N_SIMD, N_XCD = 1024, 8
vb = {}
for k, v in sq.items():
    if "SQ_ACTIVE_INST_VALU" not in v or "GRBM_GUI_ACTIVE" not in v:
        continue
    cyc_trace = avg_us.get(k, 0.0) * (prof_clk or 0.0)       # us x MHz = cycles
    # (whether the profiler reports the counter per XCD or summed over the eight is read off the trace's duration x clock)
    div = N_XCD if not cyc_trace or abs(v["GRBM_GUI_ACTIVE"] / N_XCD - cyc_trace) < abs(v["GRBM_GUI_ACTIVE"] - cyc_trace) else 1
    cyc = v["GRBM_GUI_ACTIVE"] / div
    e = {"SQ_ACTIVE_INST_VALU_quad_cycles": int(v["SQ_ACTIVE_INST_VALU"]), "GRBM_GUI_ACTIVE": int(v["GRBM_GUI_ACTIVE"]),
         "GRBM_GUI_ACTIVE_divided_by": div, "cycles_per_launch": int(cyc), "cycles_from_trace_duration_x_clock": int(cyc_trace) if cyc_trace else None,
         "valu_busy": round(4.0 * v["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cyc), 4) if cyc else None}
    if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
        # the same, per wave: share of a resident wave's life in which it executes vector instructions; x waves per SIMD = valu_busy
        e["valu_share_of_wave_cycles"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 4)
        e["waves_resident_per_simd"] = round(4.0 * v["SQ_WAVE_CYCLES"] / (N_SIMD * cyc), 3) if cyc else None
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_TRANS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
              "SQ_WAIT_INST_ANY", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS",
              "SQ_LDS_BANK_CONFLICT", "SQ_WAVES"):
        if c in v:
            e[c] = int(v[c])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and cyc:
        e["mfma_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc), 4)     # (cycles, per SIMD: the guide's unit note)
    if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
        e["lanes_active_per_valu_cycle"] = round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 4.0), 2)
    vb[k] = e
out["valu_busy"] = vb
out["valu_busy_formula"] = ("4 x SQ_ACTIVE_INST_VALU (quad-cycles, all SIMDs) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), per launch, "
                            "same PMC pass; profiles/%s_isa_mix.txt prices the same loops statically" % tag)
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
# the lease's own bench line could not know these counters yet (they were collected after it, minutes later on the same box):
# complete its roofline block from them, and say so
rf = bench.get("roofline")
if rf:
    kn = next((k for k in sorted(avg_us, key=lambda k: -avg_us[k]) if k.startswith("k_" + rf["kernel"])), rf["kernel"])
    rf["traffic"] = out["bytes_per_launch"].get(kn)
    rf["traffic_source"] = ("filled in by scripts/update_profiles.py from the PMC passes of the SAME lease (profiles/pmc_traffic.json): "
                            "(%s x FETCH_SIZE + WRITE_SIZE) per launch" % factor.get(kn, factor["default"]))
    if kn in avg_us:
        rf["profile_avg_ms"] = round(avg_us[kn] / 1e3, 4)
        rf["frac_profile"] = round(rf["algorithmic_bytes"] / (avg_us[kn] * 1e-6) / 1e9 / rf["peak"], 5)
        rel = rf["avg_ms"] / (avg_us[kn] / 1e3)
        rf["live_vs_profile"] = {"ratio": round(rel, 3), "agree_within_10pct": bool(abs(rel - 1.0) <= 0.10)}
        lclk = (bench.get("sclk_mhz") or {}).get("timed_region")      # rf["avg_ms"] is measured inside the timed region
        if prof_clk and lclk:
            reln = rel * lclk / prof_clk
            rf["live_vs_profile"].update({"sclk_mhz_live": lclk, "sclk_mhz_profile": prof_clk, "ratio_clock_normalised": round(reln, 3),
                                          "agree_within_10pct_clock_normalised": bool(abs(reln - 1.0) <= 0.10)})
    vi = out["valu_wave_instructions_per_launch"].get(kn)
    if vi:
        rate = vi / (rf["avg_ms"] * 1e-3)
        rf["valu"] = {"wave_instructions": int(vi), "G_wave_instr_per_s": round(rate / 1e9, 1), "peak_G_wave_instr_per_s": 1228.9,
                      "frac": round(rate / 1e9 / 1228.9, 4), "source": "SQ_INSTS_VALU per launch (same lease); duration measured live"}
        rf["issue_frac"] = rf["valu"]["frac"]
    b = vb.get(kn)
    if b and b.get("valu_busy") is not None:
        rf["valu_busy"] = dict(b, source="PMC passes of the same lease (profiles/pmc_traffic.json)", formula=out["valu_busy_formula"])
        if str(rf.get("binding_roof", "")).startswith("valu"):
            rf["binding_frac"] = b["valu_busy"]
    json.dump(bench, open(os.path.join(dst, "%s_bench_line.json" % tag), "w"), indent=1)
# context figure: the reference's own kernels (oracle/_ref, hipify-perl build) on the same GPU and view, scripts/compare_ref.py
cr = os.path.join(src, "compare_ref.json")
if os.path.exists(cr):
    ls = [l for l in open(cr).read().splitlines() if l.startswith("{")]
    if ls:
        r = json.loads(ls[-1])
        ref = r.get("ref_strict_ms") or r.get("ref_fast_ms")
        if ref:
            json.dump({"label": "the reference's kernels via hipify-perl (oracle/_ref), one view per call: context, not a baseline",
                       "frames_per_s": round(1e3 / (ref["forward"] + ref["backward"]), 1), "ms": ref, "workload": r.get("workload"),
                       "W": r.get("W"), "H": r.get("H"), "view": r.get("view"), "lease": out["lease"],
                       "source": "scripts/compare_ref.py (hipEvents around Reference.bench: 2 warm-up + 5 timed iterations)"},
                      open(os.path.join(dst, "reference_build.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("key", "fetch_factor", "avg_us")}, indent=1, sort_keys=True))
print(json.dumps(out["bytes_per_launch"], indent=1, sort_keys=True))
print(out["valu_wave_instructions_per_launch"])
