#!/bin/bash
# rocprofv3 kernel stats of one encoder combo's forward (tools/bench_one.py); outputs under gpurun_out/$1/
set -e
TAG=${1:-pc}; COMBO=${2:-transformer/transformer/gru}; B=${3:-16384}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_one.py $COMBO $B > $OUT/stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
