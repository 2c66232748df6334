#!/bin/bash
# A/B of the PER-DQN train step on ONE box: the in-tree library against an experimental build
# (make -C mcts_q_amd/csrc OUT=../lib/libmctsq_hip_exp.so OBJDIR=../../build/obj_exp EXTRA=-D...).
# usage: tools/ab_train.sh TAG [combo] [B]      -> gpurun_out/TAG/{base,exp}
set -e
TAG=${1:-ab}; COMBO=${2:-gru/gru/gru}; B=${3:-1024}
export TMPDIR=/tmp
for v in ${VARIANTS:-base exp}; do
  OUT=$PWD/gpurun_out/$TAG/$v; mkdir -p $OUT
  if [ $v = base ]; then unset MCTSQ_HIP_LIB; else export MCTSQ_HIP_LIB=$PWD/mcts_q_amd/lib/libmctsq_hip_$v.so; fi
  for rep in 1 2; do timeout -k 10 120 python3 tools/train_loop.py $COMBO $B 400 2>&1 | tail -1; done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/train_loop.py $COMBO $B 100 > $OUT/stats.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:18]:
    if float(r['AverageNs']) < 5e6:
        print(f"  $v {r['Name'][:60]:60s} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
done
