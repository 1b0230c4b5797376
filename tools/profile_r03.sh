#!/bin/bash
# Round-3 profiles, ONE WORKLOAD PER PROFILE (run on the GPU box via gpurun):   bash tools/profile_r03.sh <what> <tag>
#   config3   bench.py's default command: rocprofv3 kernel stats of the full timed region + separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ, f64 mix)
#   network5  bench.py --only-network (BASELINE config 5 shape at rtol = atol = 1e-8): kernel stats + SQ counters of exactly that launch
#   sens / sens_rand   the forward-sensitivity kernel (distmod n = 8, B = 65536 / randmod n = 4, B = 16384)
#   rand7     randmod n = 7, B = 1024, theta ~ U(0, 20): the parity-elimination kernel (pk_rand_parity.hpp)
#   rand8     randmod n = 8, B = 1024, theta ~ U(0, 20): the parity-elimination kernel (pk_rand_parity.hpp)
#   sens_rows distmod n = 30, B = 4096: the rows-per-lane sensitivity kernel (pk_sens_rows.hpp)
#   tpr       the thread-per-replica kernel at BASELINE config 1 size (distmod n = 4, B = 524288): kernel stats + HBM / SQ counters
set -u
WHAT=${1:-config3}
TAG=${2:-r03_$WHAT}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
case $WHAT in
  config3) CMD="python3 $REPO/bench.py --no-cpu-baseline --no-secondary"; CMDS="$CMD --steps 3 --warmup 1" ;;
  network5) CMD="python3 $REPO/bench.py --only-network"; CMDS="$CMD" ;;
  tpr) CMD="python3 $REPO/tools/gpu_one.py 0 4 524288 20"; CMDS="python3 $REPO/tools/gpu_one.py 0 4 524288 3" ;;
  sens) CMD="python3 $REPO/tools/gpu_sens_one.py distmod 8 65536 10"; CMDS="python3 $REPO/tools/gpu_sens_one.py distmod 8 65536 2" ;;
  sens_rows) CMD="python3 $REPO/tools/gpu_sens_one.py distmod 30 4096 5"; CMDS="python3 $REPO/tools/gpu_sens_one.py distmod 30 4096 2" ;;
  sens_rand) CMD="python3 $REPO/tools/gpu_sens_one.py randmod 4 16384 5"; CMDS="python3 $REPO/tools/gpu_sens_one.py randmod 4 16384 2" ;;
  rand7) CMD="python3 $REPO/tools/gpu_one.py 2 7 1024 10"; CMDS="python3 $REPO/tools/gpu_one.py 2 7 1024 2" ;;
  rand8) CMD="python3 $REPO/tools/gpu_one.py 2 8 1024 5"; CMDS="python3 $REPO/tools/gpu_one.py 2 8 1024 2" ;;
  *) echo "unknown workload $WHAT"; exit 2 ;;
esac
cd /tmp
$CMD > $OUT/bench.json 2> $OUT/bench.err || { echo "plain run failed"; tail -5 $OUT/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || { echo "rocprof stats failed"; tail -5 $OUT/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMDS > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMDS > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; }
rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sq -- $CMDS > $OUT/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $OUT/pmc_sq.log; }
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_f64 -- $CMDS > $OUT/pmc_f64.log 2>&1 || { echo "pmc f64 mix failed (counters may not exist on gfx950)"; tail -3 $OUT/pmc_f64.log; }
cd $REPO
python tools/summarize_prof.py $OUT | tee $OUT/summary.txt
