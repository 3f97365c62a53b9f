#!/usr/bin/env bash
# round 6, lease X: the -m gpu suite three times over (flakiness check of the tests added this round) + smoke
set -u
export TMPDIR=/tmp
for i in 1 2 3; do timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -1; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
