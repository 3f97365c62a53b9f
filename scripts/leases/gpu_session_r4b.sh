#!/usr/bin/env bash
# round 4, session B: where the render backward's sums lose accuracy -- matrix-core summation vs moment shift (diagnostic builds)
set -u
OUT=$PWD/gpurun_out/r4b
mkdir -p $OUT
for v in "default|" "emul32|-DGSR_BWD_EMUL=1" "emul64|-DGSR_BWD_EMUL=2" "shift64|-DGSR_BWD_SHIFT64" "emul64shift64|-DGSR_BWD_EMUL=2 -DGSR_BWD_SHIFT64"; do
  name=${v%%|*}; flags=${v#*|}
  GSR_EXTRA_FLAGS="$flags" python gaussian-pcloud-render_amd/build.py --force > $OUT/build_$name.log 2>&1 || { echo "$name: build failed"; tail -5 $OUT/build_$name.log; continue; }
  echo "=== $name ($flags)"
  timeout 900 python scripts/bwd_accuracy.py --range 0 200 2>&1 | tail -6 | tee $OUT/acc_$name.txt
  timeout 300 python scripts/bwd_accuracy.py 149 14139 14397 3 7 2>&1 | grep -v amdgpu.ids | head -20
done
python gaussian-pcloud-render_amd/build.py --force > /dev/null 2>&1
