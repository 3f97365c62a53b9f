#!/usr/bin/env bash
# round 5, lease AA: launch order for the dynamic batch forward: work estimate vs plain length (an upper bound of the work)
set -u
OUT=$PWD/gpurun_out/r5aa
mkdir -p $OUT
export TMPDIR=/tmp
run() { tag=$1; shift
  env "$@" python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --drop-in-processes 0 --no-per-view > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - $OUT/bench_$tag.json $tag <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s: value %.1f | fwd %.4f bwd %.4f sum %.4f" % (sys.argv[2], d["value"], d["kernels_ms_per_frame"]["render_forward"], d["kernels_ms_per_frame"]["render_backward"], sum(d["kernels_ms_per_frame"].values())))
PY
}
run default A=1
run knee0 GSR_ORDER_KNEE=0
run k4096e02 GSR_ORDER_KNEE=4096 GSR_ORDER_EXP=0.2
run k2048e01 GSR_ORDER_KNEE=2048 GSR_ORDER_EXP=0.1
run default2 A=1
run knee0b GSR_ORDER_KNEE=0
