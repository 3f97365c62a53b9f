#!/usr/bin/env bash
# round 5, lease P: host-side trace of the literal per-view loop (where the exposed host time goes)
set -u
OUT=$PWD/gpurun_out/r5p
mkdir -p $OUT
export TMPDIR=/tmp
python scripts/literal_host_trace.py 240 > $OUT/trace.json 2> $OUT/trace.err
cat $OUT/trace.json
tail -3 $OUT/trace.err
if [ -d scratch_ab/pkg_old ]; then   # (an older copy of the Python package put there by hand for an A/B in one lease)
  python scripts/literal_host_trace.py 240 scratch_ab/pkg_old > $OUT/trace_old.json 2> $OUT/trace_old.err
  cat $OUT/trace_old.json; tail -3 $OUT/trace_old.err
  python scripts/literal_host_trace.py 240 > $OUT/trace2.json 2>> $OUT/trace.err
  python - <<'PY'
import json
for f in ("trace", "trace_old", "trace2"):
    d = json.load(open("gpurun_out/r5p/%s.json" % f))
    print(f, "frame_us %.1f kernels_us %.1f fps %.1f" % (d["frame_us"], d["kernels_us"], 1e6 / d["frame_us"]))
PY
fi
python -m pytest tests/test_gpu_api.py tests/test_gpu_hostpath.py tests/test_gpu_batch.py tests/test_gpu_integration.py tests/test_gpu_passes.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
