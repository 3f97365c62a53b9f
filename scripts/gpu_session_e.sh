#!/usr/bin/env bash
set -u
OUT=$PWD/gpurun_out/r03e
mkdir -p $OUT
B="timeout 300 python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view --streams 1 --views-per-call 1"
for cfg in "0 0" "2 0" "2 4096" "2 6144" "2 8192" "0 6144"; do
  set -- $cfg
  GSR_COOP_MAX_VIEWS=$1 GSR_ORDER_FOLD=$2 $B > $OUT/v1_c$1_f$2.json 2>$OUT/v1_c$1_f$2.err
  python - $OUT/v1_c$1_f$2.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["kernels_ms_per_frame"]["render_forward"], d["kernels_ms_per_frame"]["render_backward"])
except Exception as e:
    print("no result", e)
PY
done
