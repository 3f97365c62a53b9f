#!/usr/bin/env bash
# Run on the MI355X box (via gpurun): kernel-trace stats + separate PMC passes for bench.py; results under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r02}
EXTRA=${2:-}            # extra bench.py arguments, e.g. "--views-per-call 12"
MODE=${3:-full}         # "trace": kernel-trace stats only (no PMC passes)
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 12 --warmup 3 --repeats 2 --no-cpu-baseline --streams 1 $EXTRA"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.log 2>&1)
if [ "$MODE" != "trace" ]; then
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/bench_pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/bench_pmc_write.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/bench_pmc_sq.log 2>&1)
fi
find $OUT -name "*.csv" | head -30
tail -2 $OUT/bench_trace.log
python $PWD/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merge small: drop the raw per-dispatch traces, keep stats + counter CSVs
find $OUT -name "*kernel_trace.csv" -size +4M -delete
