"""The `roofline` object of the bench line (the contract: algorithmic HBM bytes of the dominant kernel over its measured duration
against the 8 TB/s peak), plus what actually binds the render kernels: counter traffic and the SIMDs' occupancy from the PMC passes
of a profile lease (profiles/pmc_traffic.json, quoted only for the same workload, call shape and kernel sources)."""
import json
import os

import numpy as np

from .workload import ROOT, kernels_sha

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # what a float4 copy reaches (same guide)
VALU_PEAK_GWIPS = 1228.9     # 157.3 TFLOP/s fp32 vector spec / (64 lanes x 2 flop): wave64 instructions per second, in 1e9

STAGE_KERNEL = {"render_backward": "k_render_backward", "render_forward": "k_render_forward", "preprocess": "k_preprocess<",
                "preprocess_backward": "k_preprocess_backward", "duplicate": "k_duplicate"}


def stage_kernel(stage, table):
    """the profiled kernel of a stage: the entry of `table` (kernel name -> anything) whose name starts with the stage's kernel
    prefix -- template arguments change between rounds and call shapes (k_render_backward<2>, k_render_forward_half)"""
    pre = STAGE_KERNEL.get(stage, stage)
    hits = [k for k in (table or {}) if k.startswith(pre)]
    return hits[0] if len(hits) == 1 else (max(hits, key=lambda k: table[k] if isinstance(table[k], (int, float)) else 0) if hits else pre)




def build(avg_ms, inreg_ms, inreg, bytes_per, VPC, args, P, W, H, sclk_timed, sclk_stage, pmc_path=None):
    """avg_ms: per-stage ms per launch (stage pass); inreg_ms / inreg: the render kernels inside the timed region; bytes_per: the
    algorithmic bytes per frame and stage; returns the roofline dict (None without stage timings)."""
    pmc_why = None
    dom = max(avg_ms, key=avg_ms.get) if avg_ms else None
    roofline = None
    # HBM bytes / VALU instructions / kernel duration per launch from the committed rocprofv3 passes -- only if they were taken
    # on THIS workload and call shape (scripts/profile_gpu.sh writes the key); anything else would be a number about
    # another run
    pmc, pmc_why = None, None
    key = {"workload": args.workload, "points": P, "width": W, "height": H, "views_per_launch": VPC, "profile": args.profile,
           "forward_only": bool(args.forward_only)}
    try:
        with open(pmc_path or os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("key") != key:
            pmc_why = "profiles/pmc_traffic.json was taken on %s, this run is %s" % (json.dumps(pmc.get("key")), json.dumps(key))
            pmc = None
        elif pmc.get("kernels_sha") != kernels_sha():
            pmc_why = ("profiles/pmc_traffic.json was taken on other kernel sources (kernels_sha %s, lease %s; this tree is %s): "
                       "its counters are not quoted for this run" % (pmc.get("kernels_sha"), pmc.get("lease"), kernels_sha()))
            pmc = None
    except (OSError, ValueError) as ex:
        pmc_why = "profiles/pmc_traffic.json: %r" % (ex,)
    if dom is not None:
        # the dominant kernel's duration: hipEvents around it INSIDE the timed region when it is one of the render kernels
        # (always, so far), else the per-stage pass
        dom_ms = inreg_ms.get(dom, avg_ms[dom])
        dom_clk = float(np.median(sclk_timed)) if (dom in inreg_ms and sclk_timed) else (float(np.median(sclk_stage)) if sclk_stage else None)
        achieved = bytes_per[dom] * VPC / (dom_ms * 1e-3) / 1e9     # a launch covers VPC views
        kname = stage_kernel(dom, (pmc or {}).get("avg_us"))
        # `bound` names the roof `achieved` / `peak` / `frac` are quoted against (the contract: algorithmic HBM bytes over the
        # kernel's duration against the 8 TB/s peak); `binding_roof` names what actually limits the kernel
        render_dom = dom in ("render_backward", "render_forward")
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "hbm_frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "binding_roof": "valu (fp32 vector + matrix issue on the SIMDs; see `valu`)" if render_dom else "hbm",
                    "binding_frac": None,
                    "frac_of_achievable_6300": round(achieved / HBM_ACHIEVABLE_GBS, 5),
                    "algorithmic_bytes": int(bytes_per[dom] * VPC), "avg_ms": round(dom_ms, 4),
                    "avg_ms_measured": "hipEvents on the launch stream around every launch of this kernel inside the timed region "
                                       "(%d launches)" % len(inreg.get(dom, [])) if dom in inreg_ms else "per-stage pass after the timed region",
                    "avg_ms_stage_pass": round(avg_ms[dom], 4),
                    "views_per_launch": VPC,
                    "note": "the render kernels are VALU-bound by two orders of magnitude of arithmetic intensity (SURVEY 8d): "
                            "the fraction of the HBM roofline is structurally small; see `valu`"}
        if pmc is not None:
            roofline["traffic"] = pmc.get("bytes_per_launch", {}).get(kname)
            roofline["traffic_source"] = ("builder lease %s (kernels_sha %s = this tree's): (FETCH_SIZE x %s + WRITE_SIZE) per launch, "
                                          "rocprofv3 PMC passes of this command in the same lease as the kernel trace "
                                          "(profiles/pmc_traffic.json; the FETCH factor is the one the lease's fetch calibration "
                                          "measures for this kernel's access pattern)"
                                          % (pmc.get("lease"), pmc.get("kernels_sha"),
                                             pmc.get("fetch_factor", {}).get(kname, pmc.get("fetch_factor", {}).get("default", 2))))
            pa = pmc.get("avg_us", {}).get(kname)
            if pa:
                roofline["profile_avg_ms"] = round(pa / 1e3, 4)
                roofline["frac_profile"] = round(bytes_per[dom] * VPC / (pa * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
                rel = dom_ms / (pa / 1e3)
                lv = {"ratio": round(rel, 3), "agree_within_10pct": bool(abs(rel - 1.0) <= 0.10)}
                # the same kernel on another box / in a profiled pass runs at another clock: compare CYCLES (ms x shader clock)
                pclk = pmc.get("sclk_mhz")
                if pclk and dom_clk:
                    lclk = dom_clk
                    reln = rel * lclk / float(pclk)
                    lv.update({"sclk_mhz_live": round(lclk, 1), "sclk_mhz_profile": round(float(pclk), 1),
                               "ratio_clock_normalised": round(reln, 3),
                               "agree_within_10pct_clock_normalised": bool(abs(reln - 1.0) <= 0.10)})
                roofline["live_vs_profile"] = lv
            valu = pmc.get("valu_wave_instructions_per_launch", {}).get(kname)
            if valu:  # the render kernels are VALU-bound: wave64 fp32 issue rate against the 157.3 TFLOP/s vector spec
                rate = valu / (dom_ms * 1e-3)
                roofline["valu"] = {"wave_instructions": int(valu), "G_wave_instr_per_s": round(rate / 1e9, 1),
                                    "peak_G_wave_instr_per_s": VALU_PEAK_GWIPS, "frac": round(rate / 1e9 / VALU_PEAK_GWIPS, 4),
                                    "source": "SQ_INSTS_VALU per launch, profiles/pmc_traffic.json; duration measured live",
                                    "note": "instructions per second against one plain wave64 instruction per SIMD per 2 cycles; "
                                            "packed fp32 (4 cycles), transcendentals (8) and the backward's fp32 MFMAs (32) hold "
                                            "their SIMD longer than that, so the SIMDs are busier than this fraction says"}
                roofline["issue_frac"] = roofline["valu"]["frac"]
            vb = pmc.get("valu_busy", {}).get(kname)
            if vb and vb.get("valu_busy") is not None:
                # what binds the render kernels, measured: the share of all SIMD cycles of the launch in which a vector (or
                # matrix) instruction executes -- counters of the profile lease; the live run only contributes the duration check
                roofline["valu_busy"] = dict(vb, source="profiles/pmc_traffic.json (lease %s, kernels_sha = this tree's)" % pmc.get("lease"),
                                             formula=pmc.get("valu_busy_formula"))
                if render_dom:
                    roofline["binding_frac"] = vb["valu_busy"]
        else:
            roofline["traffic_source"] = "null: " + pmc_why
    return roofline
