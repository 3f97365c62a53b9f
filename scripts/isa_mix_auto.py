"""Static instruction mix of the shipped render kernels' hot loops, priced in SIMD issue cycles.
usage: python scripts/isa_mix_auto.py [build dir = gaussian-pcloud-render_amd/build]   (needs `build.py --save-temps`)

For every kernel of interest the loops are found from the assembly's back edges; reported are the innermost loop that holds the
per-entry alpha evaluation (v_exp_f32) and, for the backward, the loop around it that also holds the matrix instructions (the
batch flush).  Cycles per wave64 instruction on one SIMD (MI355X_MICROARCH.md "Per-instruction cycle constants" + what earlier
rounds measured on this part): plain VALU 2, packed fp32 / cmp / cndmask / cvt / ldexp / min-max / DPP 4,
transcendental 8, v_mfma_f32_16x16x4_f32 32, v_permlane32_swap 8; scalar, LDS, memory and branch instructions issue beside them."""
import collections, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gaussian-pcloud-render_amd", "build")
CYC = {"plain": 2, "packed_f32": 4, "cmp": 4, "cndmask": 4, "cvt_ldexp_rndne": 4, "minmax": 4, "dpp": 4, "trans": 8, "mfma_f32": 32,
       "permlane": 8, "mov": 2, "salu": 0, "lds": 0, "vmem": 0, "branch": 0, "wait_nop": 0, "other": 0}


def classify(op, rest):
    if op.startswith("v_mfma"): return "mfma_f32"
    if op.startswith("v_permlane"): return "permlane"
    if "dpp" in op or "row_" in rest or "quad_perm" in rest: return "dpp"
    if op.startswith("v_pk_") and "f32" in op: return "packed_f32"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")): return "trans"
    if op.startswith(("v_ldexp", "v_rndne", "v_cvt", "v_fract", "v_frexp")): return "cvt_ldexp_rndne"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith(("v_min", "v_max", "v_med3")): return "minmax"
    if op.startswith(("v_mov", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_writelane")): return "mov"
    if op.startswith("v_"): return "plain"
    if op in ("s_nop", "s_waitcnt", "s_sleep"): return "wait_nop"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    return "other"


def kernel_body(lines, sym):
    st = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and l.split(";")[0].strip().endswith(":"))
    end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    return [l.strip() for l in lines[st:end + 1]]


def loops_of(body):
    lab = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = i
    out = {}
    for i, l in enumerate(body):
        m = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and lab.get(m.group(1), 1 << 30) < i:
            out[m.group(1)] = (lab[m.group(1)], max(i, out.get(m.group(1), (0, 0))[1]))
    return sorted(out.values())


def instrs(seg):
    for l in seg:
        t = l.split(";")[0].strip()
        if not t or t.startswith((".", "//")) or t.endswith(":"):
            continue
        p = t.split(None, 1)
        yield p[0], (p[1] if len(p) > 1 else "")


def mix(seg, exclude=None):
    cnt, ops = collections.Counter(), collections.defaultdict(collections.Counter)
    for op, rest in instrs(seg):
        c = classify(op, rest)
        cnt[c] += 1
        ops[c][op] += 1
    if exclude:
        for c, n in exclude[0].items():
            cnt[c] -= n
        for c in exclude[1]:
            for o, n in exclude[1][c].items():
                ops[c][o] -= n
    return cnt, ops


def report(title, cnt, ops):
    total = sum(cnt.values())
    valu = sum(n for c, n in cnt.items() if CYC.get(c, 0) > 0)
    cyc = sum(CYC.get(c, 0) * n for c, n in cnt.items())
    print("%s: %d instructions, %d of them vector / matrix, %d SIMD cycles of issue per trip = %.2f cycles per vector instruction" % (
        title, total, valu, cyc, cyc / max(valu, 1)))
    print("   %-16s %6s %8s %8s   %s" % ("class", "count", "cyc each", "cycles", "instructions"))
    for c, n in sorted(cnt.items(), key=lambda kv: (-CYC.get(kv[0], 0) * kv[1], -kv[1])):
        if n > 0:
            print("   %-16s %6d %8d %8d   %s" % (c, n, CYC.get(c, 0), CYC.get(c, 0) * n,
                                                ", ".join("%s x%d" % (o, k) for o, k in ops[c].most_common(6) if k > 0)))
    return valu, cyc


def has(seg, pat):
    return any(op.startswith(pat) for op, _ in instrs(seg))


JOBS = [("render_bwd-hip-amdgcn-amd-amdhsa-gfx950.s", "k_render_backwardILi0E", "k_render_backward<0>: the common path of the default k_render_backward<2> (same loops; <2> adds one compare per staged entry, a ballot per batch and the cold sub-quadrant path)", True),
        ("render_fwd-hip-amdgcn-amd-amdhsa-gfx950.s", "k_render_forwardILi0E", "k_render_forward<0> (8 x 8 quadrants: batches)", False),
        ("render_fwd-hip-amdgcn-amd-amdhsa-gfx950.s", "k_render_forward_half", "k_render_forward_half (single-view submissions)", False)]
for fn, sym, title, bwd in JOBS:
    lines = open(os.path.join(BUILD, fn)).read().splitlines()
    body = kernel_body(lines, sym)
    lp = loops_of(body)
    exp_loops = [(a, b) for a, b in lp if has(body[a:b + 1], "v_exp_f32")]
    a, b = min(exp_loops, key=lambda ab: ab[1] - ab[0])
    print("## " + title)
    c0, o0 = mix(body[a:b + 1])
    unit = "one trip = a group of FOUR list entries for the 64 pixels of a quadrant" if bwd else "one trip = the entries of one step (a pair per lane half / four per step in the half-quadrant kernel) for the wave's pixels"
    v0, y0 = report("alpha-evaluation loop, asm lines %d..%d of the kernel (%s)" % (a, b, unit), c0, o0)
    if bwd:
        mf = [(x, y) for x, y in lp if has(body[x:y + 1], "v_mfma") and x <= a and y >= b]
        x, y = min(mf, key=lambda ab: ab[1] - ab[0])
        c1, o1 = mix(body[x:y + 1], exclude=(c0, o0))
        v1, y1 = report("batch flush = the loop around it minus the group loop, asm lines %d..%d (once per batch of EIGHT entries; 16 matrix "
                        "instructions of which those whose 2x2 pixel block no entry hits are skipped: 12 of 16 run on the benchmark views)" % (x, y), c1, o1)
        y1r = y1 - 4 * 32
        print("   per batch of eight entries: 2 group trips + 1 flush with 12 of 16 MFMAs = %d + %d = %d cycles for %d vector instructions "
              "= %.2f cycles per vector instruction; matrix share of the issue cycles %.0f %%" % (
                  2 * y0, y1r, 2 * y0 + y1r, 2 * v0 + v1 - 4, (2 * y0 + y1r) / (2 * v0 + v1 - 4.0), 100.0 * 12 * 32 / (2 * y0 + y1r)))
    print()
