#!/bin/bash
# Collect the round's profiles on the MI355X box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) of a short bench run
# Outputs land in gpurun_out/; copy the summaries into profiles/ afterwards (tools/pmc_summary.py).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-live-pmc > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-live-pmc > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-live-pmc > $OUT/pmc_write.log 2>&1
cd $REPO
python3 bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log | cut -c1-400
find $OUT/stats -name "*kernel_stats.csv" | head -2
