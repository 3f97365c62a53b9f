#!/usr/bin/env bash
# round 6, lease D: emission status words published in 8 copies (hot-line fan-in / 8): parity subset, kernel trace at 1 / 12 views per call
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6d}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x -k "parity or batch or api or configs or footprint or hostpath" 2>&1 | grep -v amdgpu.ids | tail -3
cd /tmp
for vpc in 1 12; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$vpc -o t -- python /root/repo/bench.py --steps 24 --warmup 6 --repeats 2 --no-cpu-baseline --no-per-view --no-stage-events --views-per-call $vpc > $OUT/p_$vpc.json 2>$OUT/p_$vpc.err
f=$(find $OUT/prof_$vpc -name "*kernel_stats.csv" | head -1)
echo "== vpc $vpc"; python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    if 'gsr::' in r["Name"]:
        print("%-78s calls %5s avg %9.1f us" % (r["Name"][:78], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
