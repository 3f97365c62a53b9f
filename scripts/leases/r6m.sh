#!/usr/bin/env bash
# round 6, lease M: the whole -m gpu suite with the opt-in code paths forced on: look-back sorts (GSR_SORT_MODE=1) and ticket numbering
# (GSR_TICKETS=1); the default paths ran in lease r6j
set -u
OUT=$PWD/gpurun_out/${LEASE:-r6m}
mkdir -p $OUT
export TMPDIR=/tmp
echo "== GSR_SORT_MODE=1 GSR_TICKETS=1"
GSR_SORT_MODE=1 GSR_TICKETS=1 timeout 2700 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tee $OUT/tests_lookback_tickets.log | tail -3
echo "== GSR_SORT_MODE=1 (blockIdx numbering)"
GSR_SORT_MODE=1 timeout 2700 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tee $OUT/tests_lookback.log | tail -3
