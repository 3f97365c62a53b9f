#!/usr/bin/env bash
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
for i in 1 2 3; do
for cfg in "GSR_OVERLAP_CALIBRATE=0" "GSR_OVERLAP_CALIBRATE=1" "GSR_OVERLAP_CALIBRATE=0 PRE_STREAMS=2" "GSR_OVERLAP_CALIBRATE=1 PRE_STREAMS=2"; do
  echo "== $cfg"
  env $cfg timeout 300 python scripts/debug/overlap_warm.py 2>&1 | grep "block [2-4]\|^overlap False" | awk '/overlap/ {print $0} /block/ {printf "%s ", $3} END {print ""}' | cut -c1-260
done
done
