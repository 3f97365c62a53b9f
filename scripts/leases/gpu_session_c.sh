#!/usr/bin/env bash
# variants + SQ counters of the render backward on the default build
set -u
OUT=$PWD/gpurun_out/session_c
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/gpu_variants.sh "$1"
BENCH="python $PWD/bench.py --steps 12 --warmup 3 --repeats 2 --no-cpu-baseline --no-per-view --streams 1"
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > $OUT/sq_counters.txt)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_a -o pmc -- $BENCH > $OUT/pmc_a.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_b -o pmc -- $BENCH > $OUT/pmc_b.log 2>&1)
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("pmc_a", "pmc_b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            n[(k, r["Counter_Name"])] += 1
    for k in acc:
        if "render" in k:
            print(tag, k, {c: "%.4g" % (v / n[(k, c)]) for c, v in acc[k].items()})
PY
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
