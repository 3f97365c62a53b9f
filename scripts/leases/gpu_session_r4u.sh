#!/usr/bin/env bash
# round 4, session U: the host-side timeline audit alone (window inside the timed block)
set -u
OUT=$PWD/gpurun_out/prof_r04
mkdir -p $OUT
export TMPDIR=/tmp
python gaussian-pcloud-render_amd/build.py > /dev/null 2>&1
TL="python $PWD/bench.py --steps 480 --warmup 12 --repeats 1 --warmup-seconds 0 --no-cpu-baseline --no-per-view --streams 1 --no-stage-events"
rm -rf $OUT/tl12 $OUT/tl1
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d $OUT/tl12 -o tl -- $TL > $OUT/tl12.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d $OUT/tl1 -o tl -- $TL --views-per-call 1 > $OUT/tl1.log 2>&1)
{ echo "== bench.py --streams 1 --no-stage-events, 12 views per rasterize_views call =="; python $PWD/scripts/timeline.py $OUT/tl12 0.74 12;
  echo; echo "== the same through the per-view call (GaussianRasterizer.forward + backward per view; the library overlaps consecutive calls) =="; python $PWD/scripts/timeline.py $OUT/tl1 0.74 1; } > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
find $OUT/tl12 $OUT/tl1 -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
