#!/bin/bash
# Run on the GPU box (via gpurun): bench line, rocprofv3 kernel-trace stats, and PMC passes for HBM
# traffic of the leaf-evaluation kernel.  Outputs under gpurun_out/$1/.
set -e
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-trees > $OUT/stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-trees > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-trees > $OUT/pmc_write.log 2>&1
echo done
