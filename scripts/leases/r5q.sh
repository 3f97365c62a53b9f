#!/usr/bin/env bash
# round 5, lease Q: cProfile of the Python path of the per-view API with the C calls stubbed
set -u
OUT=$PWD/gpurun_out/r5q
mkdir -p $OUT
python scripts/preamble_prof.py > $OUT/prof.txt 2> $OUT/prof.err
cat $OUT/prof.txt; tail -3 $OUT/prof.err
