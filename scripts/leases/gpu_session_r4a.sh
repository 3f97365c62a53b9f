#!/usr/bin/env bash
# round 4, session A: how T / (1 - alpha) is formed in the render backward -- accuracy of the render-level sums against the
# oracle's double sums (scripts/bwd_accuracy.py) and kernel time, per variant, in ONE lease
set -u
OUT=$PWD/gpurun_out/r4a
mkdir -p $OUT
for v in "div0|-DGSR_BWD_DIV=0" "div1|-DGSR_BWD_DIV=1" "div2|-DGSR_BWD_DIV=2" "div3|-DGSR_BWD_DIV=3" "div2nofma|-DGSR_BWD_DIV=2 -DGSR_BWD_NOFMA=1"; do
  name=${v%%|*}; flags=${v#*|}
  GSR_EXTRA_FLAGS="$flags" python gaussian-pcloud-render_amd/build.py --force > $OUT/build_$name.log 2>&1 || { echo "$name: build failed"; tail -5 $OUT/build_$name.log; continue; }
  echo "=== $name ($flags)"
  timeout 900 python scripts/bwd_accuracy.py 149 14139 14397 --range 0 400 --range 14100 14200 2>&1 | tail -8 | tee $OUT/acc_$name.txt
  timeout 600 python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view > $OUT/$name.json 2>$OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("%-14s %7.1f fps  bwd %.4f fwd %.4f tile_sort %.4f dup %.4f pre %.4f prebwd %.4f" % (sys.argv[2], d["value"], k["render_backward"], k["render_forward"], k["tile_sort"], k["duplicate"], k["preprocess"], k["preprocess_backward"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
echo "=== default build: tests"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
