#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + HBM PMC passes for the bench workload.
# Usage: bash tools/profile_bench.sh <tag>     -> gpurun_out/<tag>/...
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json | cut -c1-300
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1 || { echo "rocprof stats failed"; tail -5 $OUT/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $OUT/pmc_sq.log; }
cd $REPO
find $OUT -name "*.csv" | head -20
python tools/summarize_prof.py $OUT | tee $OUT/summary.txt
