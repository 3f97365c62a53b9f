#!/usr/bin/env bash
# round 6, lease B: kernel trace of the look-back sorts (mode 1) at 1 view per call
set -u
OUT=$PWD/gpurun_out/r6b
mkdir -p $OUT
export TMPDIR=/tmp
python - <<'PY' 2>&1 | tail -3
import sys; sys.path.insert(0,'gaussian-pcloud-render_amd'); sys.path.insert(0,'tests')
import torch
from diff_gaussian_rasterization import _native as N
N.selftest(torch.device('cuda:0')); print('selftest ok')
PY
cd /tmp
for mode in 1; do
for vpc in 1 12; do
GSR_SORT_MODE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${mode}_$vpc -o t -- python /root/repo/bench.py --steps 24 --warmup 6 --repeats 2 --no-cpu-baseline --no-per-view --no-stage-events --views-per-call $vpc > $OUT/p_${mode}_$vpc.json 2>$OUT/p_${mode}_$vpc.err
f=$(find $OUT/prof_${mode}_$vpc -name "*kernel_stats.csv" | head -1)
echo "== mode $mode vpc $vpc"; python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
done
