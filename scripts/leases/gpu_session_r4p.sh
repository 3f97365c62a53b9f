#!/usr/bin/env bash
set -u
OUT=$PWD/gpurun_out/r4p
mkdir -p $OUT
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 600 python scripts/debug/overlap_warm.py 2>&1 | grep -v amdgpu.ids | grep -v "block [1-6]"
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
    timeout 900 python bench.py --no-cpu-baseline > $OUT/long_$rep.json 2> $OUT/long_$rep.err
    python - $OUT/long_$rep.json long_$rep <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("%-8s %7.1f fps blocks %s wall/gpu %s | bwd %.4f fwd %.4f sum %.4f | drop_in %s %s fwd_only %s| live_vs_profile %s" % (sys.argv[2], d["value"], d["ms_per_step_blocks"], d["wall_over_gpu"], k["render_backward"], k["render_forward"], sum(k.values()), d["drop_in_api"]["frames_per_s"], d["drop_in_api"].get("overlapped_calls"), d["forward_only"], d["roofline"].get("live_vs_profile")))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
