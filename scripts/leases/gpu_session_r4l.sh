#!/usr/bin/env bash
# round 4, session L: full suite on the final build, then the round-4 profile set (one lease)
set -u
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python scripts/fuzz_sweep.py 4000 --workers 16 --runs 1 2>&1 | grep "^{\|FAILED"
bash scripts/profile_gpu.sh r04 2>&1 | tail -60
