#!/usr/bin/env bash
# round-3 session B: cooperative forward kernel -- parity (default threshold and threshold 1), then per-view timings
set -u
OUT=$PWD/gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp
for sc in random_aniso deep_stack capsule_circle big_splats; do
GSR_COOP_MAX_VIEWS=0 python scripts/debug/coop_diff.py dump $sc /tmp/a.npz && python scripts/debug/coop_diff.py dump $sc /tmp/b.npz && python scripts/debug/coop_diff.py cmp /tmp/a.npz /tmp/b.npz | grep "bad pixels"
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_hostpath.py -x -q -m gpu > $OUT/tests_default.log 2>&1
tail -3 $OUT/tests_default.log
B="timeout 300 python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view --streams 1 --views-per-call 1"
GSR_COOP_MAX_VIEWS=0 $B > $OUT/v1_nocoop.json 2>$OUT/v1_nocoop.err
$B > $OUT/v1_coop768.json 2>$OUT/v1_coop768.err
for thr in ${THRS:-1 256 2048}; do
  GSR_EXTRA_FLAGS="-DGSR_COOP_MIN_LIST=$thr" python gaussian-pcloud-render_amd/build.py --force > $OUT/build_$thr.log 2>&1
  if [ $thr = 1 ]; then
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/tests_thr1.log 2>&1
    tail -3 $OUT/tests_thr1.log
  fi
  $B > $OUT/v1_coop$thr.json 2>$OUT/v1_coop$thr.err
done
python gaussian-pcloud-render_amd/build.py --force > $OUT/build_restore.log 2>&1
for f in $OUT/v1_*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["kernels_ms_per_frame"])
except Exception as e:
    print("no result", e)
PY
done
