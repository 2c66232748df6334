#!/bin/bash
# quick per-kernel stats of the headline bench (no configs): gpurun -- 'bash tools/kstats.sh TAG'
TAG=${1:-ks}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline --no-trees --no-configs --steps 50 --warmup 5 > $OUT/stats.log 2>&1 || exit 1
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cut -d, -f1-4,6,7 "$f" | cut -c1-170 | head -${2:-16}
