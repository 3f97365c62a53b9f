#!/usr/bin/env bash
# round 4, session I: the list position in the staged record (no scalar bit extraction in the backward's group loop)
set -u
OUT=$PWD/gpurun_out/r4i
mkdir -p $OUT
for v in "pos0|-DGSR_BWD_POS_IN_STAGE=0" "pos1|-DGSR_BWD_POS_IN_STAGE=1"; do
  name=${v%%|*}; flags=${v#*|}
  GSR_EXTRA_FLAGS="$flags" python gaussian-pcloud-render_amd/build.py --force > $OUT/build_$name.log 2>&1 || { echo "$name: build failed"; tail -5 $OUT/build_$name.log; continue; }
  echo "=== $name ($flags)"
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bwd_batches.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -2
  for rep in 1 2; do
  timeout 600 python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view > $OUT/$name.json 2>$OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("%-14s %7.1f fps  bwd %.4f fwd %.4f tile_sort %.4f dup %.4f pre %.4f prebwd %.4f sclk %s" % (sys.argv[2], d["value"], k["render_backward"], k["render_forward"], k["tile_sort"], k["duplicate"], k["preprocess"], k["preprocess_backward"], d["sclk_mhz"]["timed_blocks"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
  done
done
python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
