#!/usr/bin/env bash
# round 6, lease A: look-back ("onesweep") sorts + ranges/order from the tile histogram: selftest, parity subset in both sort modes,
# per-stage kernel times at 1 and 12 views per call with the three-launch passes (mode 0) and the look-back passes (mode 1)
set -u
OUT=$PWD/gpurun_out/r6a
mkdir -p $OUT
export TMPDIR=/tmp
python - <<'PY' 2>&1 | tail -3
import sys; sys.path.insert(0,'gaussian-pcloud-render_amd'); sys.path.insert(0,'tests')
import torch
from diff_gaussian_rasterization import _native as N
N.selftest(torch.device('cuda:0')); print('selftest ok')
PY
for mode in 1 0; do
echo "== GSR_SORT_MODE=$mode"
GSR_SORT_MODE=$mode timeout 1200 python -m pytest tests -q -m gpu -x -k "parity or batch or api or configs or footprint or hostpath" 2>&1 | grep -v amdgpu.ids | tail -3
done
for mode in 0 1; do
for vpc in 1 12; do
GSR_SORT_MODE=$mode python bench.py --steps 48 --warmup 12 --repeats 3 --no-cpu-baseline --no-per-view --views-per-call $vpc > $OUT/b_${mode}_$vpc.json 2>$OUT/b_${mode}_$vpc.err
python - $OUT/b_${mode}_$vpc.json $mode <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels_ms_per_frame"]
    print("mode", sys.argv[2], d["views_per_call"], "views/call:", d["value"], "fps; sum %.4f" % sum(k.values()), {a: round(b,4) for a,b in k.items()})
except Exception as e:
    print("no result", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
done
