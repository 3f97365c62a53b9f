#!/usr/bin/env bash
set -u
OUT=$PWD/gpurun_out/r03f
mkdir -p $OUT
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 24 --warmup 12 --repeats 1 --no-cpu-baseline --no-per-view --streams 1 --views-per-call 1"
for cap in 1024 2048; do
(cd /tmp && GSR_COOP_CAP=$cap rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$cap -o t -- $B > $OUT/t$cap.log 2>&1)
python - $OUT/t$cap <<'PY'
import csv,glob,sys,os
f=glob.glob(os.path.join(sys.argv[1],"**","*kernel_stats.csv"),recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "render" in n: print(n[:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done
find $OUT -name "*.csv" -size +2M -delete
