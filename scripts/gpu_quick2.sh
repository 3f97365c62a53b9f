#!/usr/bin/env bash
# quick GPU check of a kernel change: parity + fuzz-free subset, then the default bench line (tag = $1)
set -u
TAG=${1:-quick}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_hostpath.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -5 > $OUT/tests.log
cat $OUT/tests.log
python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "per-view", d.get("per_view_api_frames_per_s"))
    print("batched", d["kernels_ms_per_frame"])
    print("drop-in", d.get("drop_in_api", {}).get("kernels_ms_per_frame"), d.get("drop_in_api", {}).get("kernel_sum_ms_per_frame"))
except Exception as e:
    print("bench unreadable", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-2000:])
PY
