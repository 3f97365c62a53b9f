#!/usr/bin/env bash
set -u
OUT=$PWD/gpurun_out/r4n
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
for rep in 1 2; do
    timeout 900 python bench.py --no-cpu-baseline > $OUT/long_$rep.json 2> $OUT/long_$rep.err
    python - $OUT/long_$rep.json long_$rep <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("%-8s %7.1f fps blocks %s wall/gpu %s | bwd %.4f fwd %.4f sum %.4f | drop_in %s %s | live_vs_profile %s" % (sys.argv[2], d["value"], d["ms_per_step_blocks"], d["wall_over_gpu"], k["render_backward"], k["render_forward"], sum(k.values()), d["drop_in_api"]["frames_per_s"], d["drop_in_api"].get("overlapped_calls"), d["roofline"].get("live_vs_profile")))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
