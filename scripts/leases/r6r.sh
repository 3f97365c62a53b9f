#!/usr/bin/env bash
# round 6, lease R: A/B of the 8x8 forward with pairs skipped when no pixel of the wave counts either entry (-DGSR_FWD_SKIP_EMPTY)
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6r}
mkdir -p $OUT
export TMPDIR=/tmp
cp gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so /tmp/libgsr_plain.so
for v in "" "-DGSR_FWD_SKIP_EMPTY" "" "-DGSR_FWD_SKIP_EMPTY"; do
GSR_EXTRA_FLAGS="$v" python gaussian-pcloud-render_amd/build.py --force > $OUT/build.log 2>&1
python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view > $OUT/b.json 2>$OUT/b.err
python - $OUT/b.json "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels_ms_per_frame"]
print("%-24s value %7.1f render_forward %.4f sum %.4f" % (sys.argv[2] or "(shipped)", d["value"], k["render_forward"], sum(k.values())))
PY
done
cp /tmp/libgsr_plain.so gaussian-pcloud-render_amd/diff_gaussian_rasterization/libgsr_hip.so
