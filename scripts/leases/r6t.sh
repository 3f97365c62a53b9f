#!/usr/bin/env bash
# round 6, lease T: standalone sort yardstick: three launches per pass vs look-back passes (tickets / blockIdx) vs rocprim, tile-sort shape
set -u
export TMPDIR=/tmp
for tk in 1 0; do ./scripts/probe/sort_probe 7300000 13 $tk 2>&1 | grep -v "^three\|^rocprim" | tail -2; done
./scripts/probe/sort_probe 7300000 13 1 2>&1 | grep "^three\|^rocprim" | sort | uniq -c | head -12
./scripts/probe/sort_probe 11800000 13 1 2>&1 | tail -10
