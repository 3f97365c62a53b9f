#!/usr/bin/env bash
# round 4, session K: headline with and without the overlap of consecutive calls (same box), both block lengths
set -u
OUT=$PWD/gpurun_out/r4k
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
for rep in 1 2; do
for v in "ovl1|1" "ovl0|0"; do
  name=${v%%|*}; on=${v#*|}
  for shape in "long|" "short|--steps 20 --warmup 5"; do
    sn=${shape%%|*}; sf=${shape#*|}
    GSR_OVERLAP=$on timeout 900 python bench.py --no-cpu-baseline --no-per-view $sf > $OUT/${name}_$sn.json 2> $OUT/${name}_$sn.err
    python - $OUT/${name}_$sn.json ${name}_$sn <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("%-12s %7.1f fps blocks %s wall/gpu %s ovl %s | bwd %.4f fwd %.4f tile_sort %.4f dup %.4f sum %.4f sclk %s" % (sys.argv[2], d["value"], d["ms_per_step_blocks"], d["wall_over_gpu"], d["overlap_of_consecutive_calls"]["calls_overlapped_in_timed_region"], k["render_backward"], k["render_forward"], k["tile_sort"], k["duplicate"], sum(k.values()), d["sclk_mhz"]["timed_blocks"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
  done
done
done
