#!/usr/bin/env bash
# build variants given as "name|extra flags" and report the 12-views-per-call kernel times of each
set -u
OUT=$PWD/gpurun_out/variants
mkdir -p $OUT
IFS=';' read -ra VARS <<< "$1"
for v in "${VARS[@]}"; do
  name=${v%%|*}; flags=${v#*|}
  GSR_EXTRA_FLAGS="$flags" python gaussian-pcloud-render_amd/build.py --force > $OUT/build_$name.log 2>&1 || { echo "$name: build failed"; tail -5 $OUT/build_$name.log; continue; }
  python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view ${BENCH_EXTRA:-} > $OUT/$name.json 2>$OUT/$name.err
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
