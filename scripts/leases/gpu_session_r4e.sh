#!/usr/bin/env bash
# round 4, session E: dL/dmean2D through the per-entry constant (GSR_BWD_MEAN_C0): accuracy, sweep, kernel time
set -u
OUT=$PWD/gpurun_out/r4e
mkdir -p $OUT
for v in "c0off|-DGSR_BWD_MEAN_C0=0" "c0on|-DGSR_BWD_MEAN_C0=1"; do
  name=${v%%|*}; flags=${v#*|}
  GSR_EXTRA_FLAGS="$flags" python gaussian-pcloud-render_amd/build.py --force > $OUT/build_$name.log 2>&1 || { echo "$name: build failed"; tail -5 $OUT/build_$name.log; continue; }
  echo "=== $name ($flags)"
  timeout 900 python scripts/bwd_accuracy.py 149 14139 14397 7140 4981 1300 --range 0 400 --range 14100 14200 2>&1 | tail -6 | tee $OUT/acc_$name.txt
  timeout 900 python scripts/fuzz_sweep.py 16000 --workers 16 --runs 2 2>&1 | grep "^{\|FAILED" | tee $OUT/fuzz_$name.log
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
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
