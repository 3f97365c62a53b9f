#!/usr/bin/env bash
# the -m gpu suite the driver runs, plus the smoke entry; log under gpurun_out/<tag>/
set -u
TAG=${1:-tests}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 2700 python -m pytest tests ${PYTEST_X--x} -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" > $OUT/tests.log
grep -E "passed|failed|error|fuzz tally|fuzz exit|variant [0-9]:|product vs contracted|FMA contraction" $OUT/tests.log | tail -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
