#!/bin/bash
# Round-3 profile set, run on the GPU box (via gpurun); outputs under gpurun_out/$1/, summarised into profiles/r03_*
# by tools/pmc_summary_r03.py.  Every rocprofv3 pass is its own run; counter passes carry only --pmc.
#   1. the bench.py line (all BASELINE configurations: gru forward, PER-DQN step, conv / transformer forward and steps)
#   2. kernel-trace stats of the same command
#   3. FETCH_SIZE / WRITE_SIZE passes (HBM traffic per launch of every kernel)
#   4. two SQ passes: matrix-pipe busy / issue mix, and LDS / wait counters
set -e
TAG=${1:-r03prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-trees"
timeout -k 10 500 python3 bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- $BENCH --steps 50 --warmup 5 > $OUT/stats_bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $BENCH --steps 10 --warmup 2 --settle-ms 0 > $OUT/pmc_$c.log 2>&1
done
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- $BENCH --steps 10 --warmup 2 --settle-ms 0 > $OUT/pmc_sq1.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -- $BENCH --steps 10 --warmup 2 --settle-ms 0 > $OUT/pmc_sq2.log 2>&1
echo done
