#!/bin/bash
# Round-2 profile set, run on the GPU box (via gpurun); outputs under gpurun_out/$1/, summarised into profiles/
# by tools/pmc_summary.py.  Every rocprofv3 pass is its own run; counter passes carry only --pmc.
#   1. bench.py line + kernel-trace stats of the same command (leaf evaluation, config 2)
#   2. kernel-trace stats of the PER-DQN train step (tools/train_loop.py, config 3)
#   3. FETCH_SIZE / WRITE_SIZE passes over bench.py (leaf kernel) and over the train step (its kernels)
#   4. SQ pass (matrix-pipe busy share) over bench.py
set -e
TAG=${1:-r02prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-trees > $OUT/stats_bench.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -- python3 tools/train_loop.py gru/gru/gru 1024 100 > $OUT/stats_train.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_bench_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-trees > $OUT/pmc_bench_$c.log 2>&1
  MCTSQ_REPLAY=1000000 timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_train_$c -- python3 tools/train_loop.py gru/gru/gru 1024 20 > $OUT/pmc_train_$c.log 2>&1
done
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-train --no-trees > $OUT/pmc_sq.log 2>&1
echo done
