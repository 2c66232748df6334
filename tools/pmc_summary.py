#!/usr/bin/env python3
"""Summarises the two rocprofv3 --pmc passes of tools/profile_bench.sh (FETCH_SIZE, WRITE_SIZE) for the
dominant kernel into profiles/<round>_pmc_traffic.json.  HBM bytes per launch = 2 * FETCH_SIZE KB (gfx950
counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section) + WRITE_SIZE KB, KB = 1024 B.
    python3 tools/pmc_summary.py gpurun_out/<tag> profiles/r01_pmc_traffic.json [kernel substring]"""
import csv, glob, json, sys

tag, out = sys.argv[1], sys.argv[2]
kern = sys.argv[3] if len(sys.argv) > 3 else "gru3_head_fwd_kernel<false>"
res = {"kernel": kern}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    vals = []
    for f in glob.glob(f"{tag}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    vals = vals[2:] if len(vals) > 4 else vals          # drop the cold first launches
    res[f"{name}_KB_avg"] = sum(vals) / max(1, len(vals))
    res[f"{name}_launches"] = len(vals)
res["correction"] = ("FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md section HBM); "
                     "WRITE_SIZE as is; KB = 1024 B")
res["hbm_bytes_per_launch"] = (2 * res["FETCH_SIZE_KB_avg"] + res["WRITE_SIZE_KB_avg"]) * 1024
res["command"] = ("rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --output-format csv -- python3 bench.py "
                  "--steps 10 --warmup 2 --no-cpu-baseline --no-train --no-trees  (tools/profile_bench.sh)")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
