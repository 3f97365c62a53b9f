#!/usr/bin/env bash
# full GPU test suite, default bench, single-stream timeline traces (12 views per call and per-view calls)
set -u
TAG=${1:-r03a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1500 $OUT/bench_default.json
B="python $PWD/bench.py --steps 48 --warmup 12 --repeats 1 --no-cpu-baseline --no-per-view --streams 1 --no-stage-events"
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d $OUT/tl12 -o tl -- $B > $OUT/tl12.log 2>&1)
python scripts/timeline.py $OUT/tl12 0.5 12 > $OUT/timeline_v12.txt 2>&1
cat $OUT/timeline_v12.txt
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d $OUT/tl1 -o tl -- $B --views-per-call 1 > $OUT/tl1.log 2>&1)
python scripts/timeline.py $OUT/tl1 0.5 1 > $OUT/timeline_v1.txt 2>&1
cat $OUT/timeline_v1.txt
find $OUT -name "*.csv" -size +8M -delete
find $OUT -name "*.db" -delete
