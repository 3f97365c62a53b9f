#!/usr/bin/env bash
# round 5, lease A: how T / (1 - alpha) is formed in the render backward -- rcp only (GSR_BWD_DIV=0) against rcp + residual step (1):
# kernel times at 12 views per call and at one view per call, the accuracy table of each against the float64 render backward,
# then the full -m gpu suite on the rcp-only build
set -u
OUT=$PWD/gpurun_out/r5a
mkdir -p $OUT
export TMPDIR=/tmp
for v in "div1|-DGSR_BWD_DIV=1" "div0|-DGSR_BWD_DIV=0"; do
  name=${v%%|*}; flags=${v#*|}
  GSR_EXTRA_FLAGS="$flags" python gaussian-pcloud-render_amd/build.py --force > $OUT/build_$name.log 2>&1 || { echo "$name: build failed"; tail -5 $OUT/build_$name.log; continue; }
  python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - $OUT/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]; di=d.get("drop_in_api",{})
    print("%-6s %7.1f fps  bwd %.4f fwd %.4f | per-view api %s  kernels %s" % (sys.argv[2], d["value"], k["render_backward"], k["render_forward"], di.get("frames_per_s"), di.get("kernels_ms_per_frame")))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
  python scripts/bwd_accuracy.py 149 14139 14397 --range 0 400 --range 14100 14200 > $OUT/accuracy_$name.txt 2>&1
  tail -6 $OUT/accuracy_$name.txt
done
bash scripts/gpu_tests.sh r5a
