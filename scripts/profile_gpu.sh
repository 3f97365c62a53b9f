#!/usr/bin/env bash
# Run on the MI355X box (via gpurun), ONE lease: the bench line, a kernel-trace stats pass, separate PMC passes of the same
# command (FETCH_SIZE, WRITE_SIZE, SQ counters: never combined with trace domains) and the FETCH_SIZE calibration probe.
# Results under gpurun_out/prof_<tag>/; scripts/update_profiles.py <tag> copies what is judged into profiles/.
set -u
TAG=${1:-r06}
EXTRA=${2:-}            # extra bench.py arguments, e.g. "--views-per-call 12"
MODE=${3:-full}         # "trace": kernel-trace stats only (no PMC passes)
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python $PWD/bench.py $EXTRA > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
BENCH="python $PWD/bench.py --steps 24 --warmup 3 --repeats 2 --no-cpu-baseline --no-per-view --streams 1 $EXTRA"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.log 2>&1)
if [ "$MODE" != "trace" ]; then
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/bench_pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/bench_pmc_write.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/bench_pmc_sq.log 2>&1)
# second SQ pass (8 slots per pass): what the VALU time is made of, and what the waves wait for (roofline.valu_busy, update_profiles.py)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/bench_pmc_sq2.log 2>&1)
# what FETCH_SIZE / WRITE_SIZE mean for this library's access patterns, on this box
mkdir -p $OUT/calib
if [ ! -x scripts/probe/fetch_calib ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/probe/fetch_calib scripts/probe/fetch_calib.hip; fi
CAL=$PWD/scripts/probe/fetch_calib
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/calib/fetch -o c -- $CAL > $OUT/calib/calib.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/calib/write -o c -- $CAL > $OUT/calib/calib_w.log 2>&1)
python $PWD/scripts/probe/fetch_calib_report.py $OUT/calib > $OUT/fetch_calibration.txt 2>&1
cat $OUT/fetch_calibration.txt
fi
# host-side audit: GPU idle time, copies and blocking API calls per rasterizer call (single stream, no stage events)
# (one long timed block; the untimed blocks before it -- priming, instrumentation warm-up, the block behind the garbage collection --
# each end in a synchronize, so the audited window is cut from the last quarter of the trace: inside the timed block)
TL="python $PWD/bench.py --steps 480 --warmup 12 --repeats 1 --warmup-seconds 0 --no-cpu-baseline --no-per-view --streams 1 --no-stage-events $EXTRA"
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d $OUT/tl12 -o tl -- $TL > $OUT/tl12.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d $OUT/tl1 -o tl -- $TL --views-per-call 1 > $OUT/tl1.log 2>&1)
{ echo "== bench.py --streams 1 --no-stage-events, 12 views per rasterize_views call =="; python $PWD/scripts/timeline.py $OUT/tl12 0.74 12;
  echo; echo "== the same through the per-view call (GaussianRasterizer.forward + backward per view) =="; python $PWD/scripts/timeline.py $OUT/tl1 0.74 1; } > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
find $OUT/tl12 $OUT/tl1 -name "*.csv" -size +1M -delete
# secondary lines: configs[1] (200K voxelised, 1080p, forward only) and configs[4] (2M points, 4K, fwd+bwd, the
# 8 views of a turn in one call on this one GPU)
if [ "$MODE" != "trace" ]; then
python $PWD/bench.py --config 1 --no-cpu-baseline --no-per-view > $OUT/bench_config1.json 2> $OUT/bench_config1.err
python $PWD/bench.py --config 4 --steps 16 --warmup 8 --no-cpu-baseline --no-per-view > $OUT/bench_config4.json 2> $OUT/bench_config4.err
python $PWD/bench.py --config 3 --steps 16 --warmup 8 --no-cpu-baseline --no-per-view > $OUT/bench_config3.json 2> $OUT/bench_config3.err
tail -c 300 $OUT/bench_config1.err $OUT/bench_config4.err $OUT/bench_config3.err
fi
python $PWD/scripts/compare_ref.py 0 > $OUT/compare_ref.json 2> $OUT/compare_ref.err
# the round driver's command line (20-step blocks: one 12-view and one 8-view submission each)
python $PWD/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err
python $PWD/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
head -40 $OUT/summary.txt
# keep the merge small: drop the raw per-dispatch traces, keep stats + counter CSVs
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*.db" -delete
