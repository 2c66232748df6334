#!/bin/bash
# rocprofv3 kernel stats of the PER-DQN train step; outputs under gpurun_out/$1/
set -e
TAG=${1:-ptrain}; COMBO=${2:-gru/gru/gru}; B=${3:-1024}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python3 tools/train_loop.py $COMBO $B 200 > $OUT/plain.log 2>&1
tail -1 $OUT/plain.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/train_loop.py $COMBO $B 100 > $OUT/stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
