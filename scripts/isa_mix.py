"""Instruction mix of a kernel's hot loop from the compiler's assembly (hipcc -save-temps).
usage: python scripts/isa_mix.py <file.s> <kernel symbol substring> <first label> [<last label>]
Counts the instructions between <first label> and the next label that follows the loop's back edge (or <last label>), by
class, with the issue cost measured on this part (ns per wave64 instruction per SIMD)."""
import collections, re, sys

path, kernel, first = sys.argv[1], sys.argv[2], sys.argv[3]
last = sys.argv[4] if len(sys.argv) > 4 else None
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kernel in l and l.split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].strip() == "s_endpgm")
body = lines[start:end + 1]
i0 = next(i for i, l in enumerate(body) if l.startswith(first + ":"))
if last:
    i1 = next(i for i, l in enumerate(body) if l.startswith(last + ":"))
else:
    # the loop ends at the last branch back to the first label
    i1 = max(i for i, l in enumerate(body) if re.search(r"s_cbranch\w*\s+" + re.escape(first) + r"\b", l) or re.search(r"s_branch\s+" + re.escape(first) + r"\b", l)) + 1
COST = {"pk_f32": 1.9, "valu_f32": 1.05, "valu_int": 1.05, "trans": 3.5, "dpp": 1.9, "permlane": 3.4, "cndmask": 1.9, "cmp": 1.9,
        "mov": 1.05, "cvt_ldexp_rndne": 1.9, "mfma_f32": 19.6, "salu": 0.0, "s_nop": 0.0, "s_waitcnt": 0.0, "lds": 0.0, "vmem": 0.0, "branch": 0.0, "minmax": 1.9}


def classify(op, rest):
    if op.startswith("v_mfma"): return "mfma_f32"   # 16x16x4 f32 among VALU work: 19.6 ns (13.7 back to back), mfma_overlap_probe
    if op.startswith("v_permlane"): return "permlane"
    if "dpp" in op or "row_" in rest or "quad_perm" in rest: return "dpp"
    if op.startswith("v_pk_") and "f32" in op: return "pk_f32"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")): return "trans"
    if op.startswith(("v_ldexp", "v_rndne", "v_cvt", "v_fract", "v_frexp")): return "cvt_ldexp_rndne"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith(("v_min", "v_max", "v_med3")): return "minmax"
    if op.startswith(("v_mov", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_writelane")): return "mov"
    if op.startswith("v_") and ("f32" in op or "f16" in op): return "valu_f32"
    if op.startswith("v_"): return "valu_int"
    if op == "s_nop": return "s_nop"
    if op == "s_waitcnt": return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    return "other"


cnt = collections.Counter()
ops = collections.defaultdict(collections.Counter)
for l in body[i0:i1]:
    t = l.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    t = t.split(";")[0].strip()
    if not t:
        continue
    parts = t.split(None, 1)
    op, rest = parts[0], parts[1] if len(parts) > 1 else ""
    c = classify(op, rest)
    cnt[c] += 1
    ops[c][op] += 1
total = sum(cnt.values())
valu = sum(v for k, v in cnt.items() if COST.get(k, 1.0) > 0)
ns = sum(COST.get(k, 1.0) * v for k, v in cnt.items())
print("%s: %s .. line %d of the kernel (%d instructions, %d of them VALU, %.0f ns of vector issue per trip at the measured costs)" % (
    kernel, first, i1, total, valu, ns))
print("%-18s %6s %8s %8s   %s" % ("class", "count", "ns each", "ns", "instructions"))
for k, v in sorted(cnt.items(), key=lambda kv: -COST.get(kv[0], 1.0) * kv[1] - 1e-3 * kv[1]):
    print("%-18s %6d %8.2f %8.1f   %s" % (k, v, COST.get(k, 1.0), COST.get(k, 1.0) * v,
                                          ", ".join("%s x%d" % (o, n) for o, n in ops[k].most_common(6))))
