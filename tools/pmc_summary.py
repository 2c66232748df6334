#!/usr/bin/env python3
"""Summarises the rocprofv3 passes of tools/profile_r02.sh into profiles/<round>_*:
    python3 tools/pmc_summary.py gpurun_out/<tag> r02
  profiles/r02_pmc_traffic.json      HBM bytes per launch of every hot-path kernel (FETCH_SIZE doubled: gfx950 counts
                                     128-B requests as 64 B, MI355X_MICROARCH.md section HBM; WRITE_SIZE as is; KB =
                                     1024 B) with the kernel's average duration from the kernel-trace pass of the same
                                     command -> achieved HBM GB/s; top-level keys describe the dominant kernel
  profiles/r02_sq_counters.json      SQ counters of the leaf-evaluation kernel, matrix-pipe busy share
  profiles/r02_*_kernel_stats.csv    the kernel-trace summaries themselves
"""
import collections
import csv
import glob
import json
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
DOMINANT = "gru3_head_fwd_kernel<false>"


def counter(sub, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{tag}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(sub):
    out = {}
    for f in glob.glob(f"{tag}/{sub}/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            out[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    return out


def short(k):
    return k.replace("mctsq::", "").replace("void ", "").split("(")[0]


res = {"correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE "
                     "as is; KB = 1024 B; first two launches of every kernel dropped (cold)",
       "commands": "tools/profile_r02.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate runs) -- python3 bench.py --steps 10 "
                   "--warmup 2 --no-cpu-baseline --no-train --no-trees   and   -- python3 tools/train_loop.py gru/gru/gru 1024 20; "
                   "durations from rocprofv3 --kernel-trace --stats of the same programs", "kernels": {}}
for what, stats in (("bench", "stats_bench"), ("train", "stats_train")):
    fetch, write, dur = counter(f"pmc_{what}_FETCH_SIZE", "FETCH_SIZE"), counter(f"pmc_{what}_WRITE_SIZE", "WRITE_SIZE"), durations(stats)
    for k in sorted(fetch):
        if "mctsq" not in k or (what == "train" and short(k) in res["kernels"]):
            continue
        fv, wv = fetch[k], write.get(k, [0.0])
        fv, wv = (fv[2:] if len(fv) > 4 else fv), (wv[2:] if len(wv) > 4 else wv)
        hbm = (2 * sum(fv) / len(fv) + sum(wv) / max(1, len(wv))) * 1024
        us = dur.get(k, (None, 0))[0]
        res["kernels"][short(k)] = {"program": "bench.py" if what == "bench" else "tools/train_loop.py gru/gru/gru B=1024, 1M-entry replay",
                                    "FETCH_SIZE_KB_avg": sum(fv) / len(fv), "WRITE_SIZE_KB_avg": sum(wv) / max(1, len(wv)),
                                    "launches": len(fv), "hbm_bytes_per_launch": hbm, "avg_us": us,
                                    "hbm_GBs": (hbm / (us * 1e-6) / 1e9) if us else None}
dom = res["kernels"].get(DOMINANT)
if dom:
    res["kernel"] = DOMINANT
    res["hbm_bytes_per_launch"] = dom["hbm_bytes_per_launch"]
json.dump(res, open(f"profiles/{rnd}_pmc_traffic.json", "w"), indent=1)
for k, v in res["kernels"].items():
    print(f"{k:40s} {v['hbm_bytes_per_launch'] / 1e6:9.3f} MB/launch  {v['avg_us'] or 0:8.1f} us  {v['hbm_GBs'] or 0:8.1f} GB/s")

sq = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{tag}/pmc_sq/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        sq[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY "
                  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline "
                  "--no-train --no-trees", "kernels": {}}
for k, d in sq.items():
    if "mctsq" not in k:
        continue
    avg = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in d.items()}
    # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD (4 per CU); SQ_BUSY_CU_CYCLES per CU
    if avg.get("SQ_BUSY_CU_CYCLES"):
        avg["mfma_busy_frac_of_4_simds"] = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * avg["SQ_BUSY_CU_CYCLES"])
    if avg.get("SQ_WAVE_CYCLES"):
        avg["wait_any_frac_of_wave_cycles"] = avg.get("SQ_WAIT_ANY", 0.0) / avg["SQ_WAVE_CYCLES"]
    out["kernels"][short(k)] = avg
    if DOMINANT in k:
        out["kernel"] = DOMINANT
        out["mfma_busy_frac"] = avg.get("mfma_busy_frac_of_4_simds")
json.dump(out, open(f"profiles/{rnd}_sq_counters.json", "w"), indent=1)
print("mfma_busy_frac", out.get("mfma_busy_frac"))
for what, name in (("stats_bench", "bench"), ("stats_train", "train_step_gru_B1024")):
    fs = glob.glob(f"{tag}/{what}/*/*_kernel_stats.csv")
    if fs:
        shutil.copy(fs[0], f"profiles/{rnd}_{name}_kernel_stats.csv")
if glob.glob(f"{tag}/bench.json"):
    shutil.copy(f"{tag}/bench.json", f"profiles/{rnd}_bench.json")
