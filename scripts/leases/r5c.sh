#!/usr/bin/env bash
# round 5, lease C: the whole -m gpu suite on the tree with the call-trace fixture, the per-view extra channels, the opt-in overlap
# and the new bench line; then the default bench line itself
set -u
OUT=$PWD/gpurun_out/r5c
mkdir -p $OUT
export TMPDIR=/tmp
PYTEST_X= bash scripts/gpu_tests.sh r5c
grep -E "^FAILED|^ERROR" $OUT/tests.log | head -20
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5c/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "drop_in", json.dumps(d["drop_in"]))
    print("drop_in_api", json.dumps(d["drop_in_api"]["frames_per_s"]), d["drop_in_api"].get("overlapped_calls"))
    print("roofline", json.dumps({k: d["roofline"][k] for k in ("bound", "frac", "binding_roof", "binding_frac", "traffic", "traffic_source")}))
    print(d["kernels_ms_per_frame"], d["drop_in_api"]["kernels_ms_per_frame"])
except Exception as e:
    print("unreadable", e)
PY
