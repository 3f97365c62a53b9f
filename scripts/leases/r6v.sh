#!/usr/bin/env bash
# round 6, lease V: A/B of the backward contraction: two accumulator chains (first run), then block-skip tests per row of blocks / none
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6v}
mkdir -p $OUT
export TMPDIR=/tmp
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
for v in "" "-DGSR_BWD_ROWSKIP" "-DGSR_BWD_NOSKIP" "" "-DGSR_BWD_ROWSKIP" "-DGSR_BWD_NOSKIP"; do
GSR_EXTRA_FLAGS="$v" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1
python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view > $OUT/b.json 2>$OUT/b.err
python - $OUT/b.json "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels_ms_per_frame"]
print("%-18s value %7.1f render_backward %.4f in-region %.4f" % (sys.argv[2] or "(shipped)", d["value"], k["render_backward"], d["roofline"]["avg_ms"]))
PY
done
GSR_EXTRA_FLAGS="-DGSR_BWD_ROWSKIP" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1
timeout 900 python -m pytest tests -q -m gpu -x -k "parity or bwd_batches or api" 2>&1 | grep -v amdgpu.ids | tail -2
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
