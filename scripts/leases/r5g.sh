#!/usr/bin/env bash
# round 5, lease G: the render backward with moments about the four sub-quadrant centres (GSR_BWD_SUBQ=1) against the default:
# tests, accuracy against the float64 render backward, kernel time, fuzz-sweep exits
set -u
OUT=$PWD/gpurun_out/r5g
mkdir -p $OUT
export TMPDIR=/tmp
GSR_BWD_SUBQ=1 timeout 1500 python -m pytest tests -q -m gpu -x -k "parity or fullsize or batch or api or configs or fuzz" 2>&1 | grep -v amdgpu.ids | tail -4
for sq in 0 1; do
  GSR_BWD_SUBQ=$sq python scripts/bwd_accuracy.py 149 14139 14397 --range 0 400 --range 14100 14200 > $OUT/accuracy_subq$sq.txt 2>&1
  echo "== GSR_BWD_SUBQ=$sq"; tail -5 $OUT/accuracy_subq$sq.txt
done
for sq in 0 1 0 1; do
  GSR_BWD_SUBQ=$sq python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view > $OUT/bench_subq$sq.json 2> $OUT/bench_subq$sq.err
  python - $OUT/bench_subq$sq.json $sq <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("subq=%s value %.1f  bwd %.4f (in-region %.4f ms per launch) fwd %.4f" % (sys.argv[2], d["value"], k["render_backward"], d["roofline"]["avg_ms"], k["render_forward"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
GSR_BWD_SUBQ=1 python scripts/fuzz_sweep.py 16000 --workers 16 --runs 1 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
