#!/usr/bin/env bash
# PMC counters for the render kernels only (bench workload, few steps)
set -u
OUT=$PWD/gpurun_out/pmc_render
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 12 --warmup 2 --repeats 1 --no-cpu-baseline --streams 1 ${1:-}"
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/a -o pmc -- $BENCH > $OUT/a.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $OUT/b -o pmc -- $BENCH > $OUT/b.log 2>&1)
python - <<PY
import csv, glob, collections
for tag in ("a","b"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); seen=set()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("gsr::","").replace("void ","")
            if "render" not in k: continue
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            key=(k,r["Dispatch_Id"])
            if key not in seen: seen.add(key); cnt[k]+=1
    for k in acc:
        print(k, "launches", cnt[k], " ".join("%s=%.4g"%(c,acc[k][c]/cnt[k]) for c in sorted(acc[k])))
PY
