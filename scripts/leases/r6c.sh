#!/usr/bin/env bash
# round 6, lease C: blocks numbered by blockIdx instead of tickets (emission + look-back scatter): selftest, parity subset, kernel
# trace at 1 / 12 views per call for sort mode 0 / 1 and tickets on / off
set -u
OUT=$PWD/gpurun_out/r6c
mkdir -p $OUT
export TMPDIR=/tmp
for mode in 1 0; do
echo "== GSR_SORT_MODE=$mode"
GSR_SORT_MODE=$mode timeout 1200 python -m pytest tests -q -m gpu -x -k "parity or batch or api or configs or footprint or hostpath" 2>&1 | grep -v amdgpu.ids | tail -3
done
cd /tmp
for cfg in "1 0" "0 0" "0 1"; do
set -- $cfg; mode=$1; tk=$2
for vpc in 1 12; do
GSR_TICKETS=$tk GSR_SORT_MODE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${mode}_${tk}_$vpc -o t -- python /root/repo/bench.py --steps 24 --warmup 6 --repeats 2 --no-cpu-baseline --no-per-view --no-stage-events --views-per-call $vpc > $OUT/p_${mode}_${tk}_$vpc.json 2>$OUT/p_${mode}_${tk}_$vpc.err
f=$(find $OUT/prof_${mode}_${tk}_$vpc -name "*kernel_stats.csv" | head -1)
echo "== sort mode $mode tickets $tk vpc $vpc"; python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'gsr::' in r["Name"] and 'render' not in r["Name"] and 'preprocess' not in r["Name"]:
        print("%-78s calls %5s avg %9.1f us" % (r["Name"][:78], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
done
