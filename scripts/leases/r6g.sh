#!/usr/bin/env bash
# round 6, lease G: adaptive sub-quadrant moments in the render backward (GSR_BWD_SUBQ=2, the new default): tests, accuracy of
# modes 0 / 2 / 1 against the float64 render backward, kernel time of modes 0 / 2 at 12 views and 1 view per call
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6g}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "parity or batch or api or bwd_batches or fullsize" 2>&1 | grep -v amdgpu.ids | tail -3
for mode in 0 2 1; do
echo "== GSR_BWD_SUBQ=$mode"
GSR_BWD_SUBQ=$mode python scripts/bwd_accuracy.py 149 14139 14397 --range 0 500 > $OUT/acc_$mode.txt 2> $OUT/acc_$mode.err; tail -8 $OUT/acc_$mode.txt
done
for mode in 0 2 0 2; do
for vpc in 12 1; do
GSR_BWD_SUBQ=$mode python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view --views-per-call $vpc > $OUT/b_${mode}_$vpc.json 2>$OUT/b_${mode}_$vpc.err
python - $OUT/b_${mode}_$vpc.json $mode <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("subq mode", sys.argv[2], d["views_per_call"], "views/call:", d["value"], "fps; render_backward %.4f (in-region %s) sum %.4f" % (k["render_backward"], d["roofline"].get("avg_ms"), sum(k.values())))
except Exception as e:
    print("no result", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
done
