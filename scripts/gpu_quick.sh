#!/usr/bin/env bash
# quick check of a kernel change: selftest + parity subset, then the bench's kernel times (12 views per call and per-view)
set -u
OUT=$PWD/gpurun_out/quick
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
for extra in "" "--views-per-call 1"; do
python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view $extra > $OUT/b.json 2>$OUT/b.err
python - $OUT/b.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["views_per_call"], "views/call:", d["value"], "fps; single-stream", d["single_stream"]["frames_per_s"], {k: v for k, v in d["kernels_ms_per_frame"].items()})
except Exception as e:
    print("no result", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
