#!/usr/bin/env bash
# round 4, session M: events without the system-scope fence (landing-zone event, per-stage timing events): suite + headline both shapes
set -u
OUT=$PWD/gpurun_out/r4m
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
  for shape in "long|" "short|--steps 20 --warmup 5"; do
    sn=${shape%%|*}; sf=${shape#*|}
    timeout 900 python bench.py --no-cpu-baseline $sf > $OUT/${sn}_$rep.json 2> $OUT/${sn}_$rep.err
    python - $OUT/${sn}_$rep.json ${sn}_$rep <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("%-8s %7.1f fps blocks %s gpu/step %s wall/gpu %s | bwd %.4f fwd %.4f tile_sort %.4f dup %.4f pre %.4f sum %.4f | drop_in %s | live_vs_profile %s" % (sys.argv[2], d["value"], d["ms_per_step_blocks"], d["gpu_ms_per_step_timed"], d["wall_over_gpu"], k["render_backward"], k["render_forward"], k["tile_sort"], k["duplicate"], k["preprocess"], sum(k.values()), d["drop_in_api"]["frames_per_s"], d["roofline"].get("live_vs_profile")))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
  done
done
