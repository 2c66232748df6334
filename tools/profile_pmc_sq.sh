#!/bin/bash
# SQ counter pass (matrix-pipe utilisation evidence) for one command; outputs under gpurun_out/$1/
#   tools/profile_pmc_sq.sh tag python3 bench.py ...      (the program itself follows the tag)
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- "$@" > $OUT/pmc_sq.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_sq/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "mctsq" not in k: continue
    print(k, {c: round(sum(v[1:]) / max(1, len(v) - 1)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
