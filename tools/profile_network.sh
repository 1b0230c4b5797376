#!/bin/bash
# rocprofv3 kernel stats + SQ counters for the network solve kernel on the config-5-shaped synthetic network (run via gpurun).
set -u
TAG=${1:-r01_network}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/tools/gpu_bench_network.py 8192 > $OUT/stats.log 2>&1 || { echo stats failed; tail -5 $OUT/stats.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/tools/gpu_bench_network.py 2048 > $OUT/pmc_sq.log 2>&1 || { echo pmc failed; tail -5 $OUT/pmc_sq.log; }
cd $REPO
grep model $OUT/stats.log
python tools/summarize_prof.py $OUT | grep -E "net_solve|==" | tee $OUT/summary.txt
