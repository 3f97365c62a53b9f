#!/usr/bin/env bash
# round 4, session C: render-backward accuracy against the float64 arbiter, per division variant; then the 16 000-case sweep
set -u
OUT=$PWD/gpurun_out/r4c
mkdir -p $OUT
for v in "div0|-DGSR_BWD_DIV=0" "div1|-DGSR_BWD_DIV=1" "div2|-DGSR_BWD_DIV=2" "div2nofma|-DGSR_BWD_DIV=2 -DGSR_BWD_NOFMA=1" "emul64shift64|-DGSR_BWD_EMUL=2 -DGSR_BWD_SHIFT64"; do
  name=${v%%|*}; flags=${v#*|}
  GSR_EXTRA_FLAGS="$flags" python gaussian-pcloud-render_amd/build.py --force > $OUT/build_$name.log 2>&1 || { echo "$name: build failed"; tail -5 $OUT/build_$name.log; continue; }
  echo "=== $name ($flags)"
  timeout 900 python scripts/bwd_accuracy.py 149 14139 14397 --range 0 400 --range 14100 14200 2>&1 | tail -6 | tee $OUT/acc_$name.txt
done
python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
echo "=== default build: against the float32-term sums (the old yardstick)"
timeout 900 python scripts/bwd_accuracy.py --f32terms 149 14139 14397 --range 0 400 --range 14100 14200 2>&1 | tail -6 | tee $OUT/acc_default_f32terms.txt
echo "=== default build: fuzz sweep"
timeout 3000 python scripts/fuzz_sweep.py 16000 --workers 16 --runs 1 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_16000_run0.log | tail -30
