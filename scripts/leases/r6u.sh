#!/usr/bin/env bash
# round 6, lease U: what the render kernels' waves do when they are not issuing vector instructions: scalar / LDS / branch counters
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6u}
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 24 --warmup 3 --repeats 2 --no-cpu-baseline --no-per-view --streams 1"
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE -d $OUT/pmc -o pmc -- $BENCH > $OUT/log 2>&1)
python - $OUT <<'PY'
import csv,glob,os,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1],"pmc","**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("gsr::","").replace("void ","")
        if k.startswith("k_render"):
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, len(n[k]), {c: "%.4g" % (v/len(n[k])) for c,v in sorted(acc[k].items())})
PY
