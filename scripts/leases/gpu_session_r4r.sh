#!/usr/bin/env bash
# round 4, session R: final build: full suite, sweep, headline both shapes
set -u
OUT=$PWD/gpurun_out/r4r
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 900 python scripts/fuzz_sweep.py 16000 --workers 16 --runs 1 2>&1 | grep "^{\|FAILED"
for shape in "long|" "short|--steps 20 --warmup 5"; do
    sn=${shape%%|*}; sf=${shape#*|}
    timeout 900 python bench.py $sf > $OUT/$sn.json 2> $OUT/$sn.err
    python - $OUT/$sn.json $sn <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("%-8s %7.1f fps blocks %s wall/gpu %s sclk %s | bwd %.4f fwd %.4f sum %.4f | drop_in %s fwd_only %s | cpu %s | live_vs_profile %s" % (sys.argv[2], d["value"], d["ms_per_step_blocks"], d["wall_over_gpu"], d["sclk_mhz"]["timed_blocks"], k["render_backward"], k["render_forward"], sum(k.values()), d["drop_in_api"]["frames_per_s"], d["forward_only"], (d["cpu_baseline"] or {}).get("value"), d["roofline"].get("live_vs_profile")))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
