#!/usr/bin/env python3
"""Summarises the rocprofv3 passes of tools/profile_r03.sh into profiles/r03_*:
    python3 tools/pmc_summary_r03.py gpurun_out/<tag>
  profiles/r03_bench.json             the bench.py line of the same box
  profiles/r03_bench_kernel_stats.csv rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-trees`
  profiles/r03_pmc_traffic.json       HBM bytes per launch of every hot-path kernel (FETCH_SIZE doubled: gfx950 counts
                                      128-B requests as 64 B, MI355X_MICROARCH.md section HBM; WRITE_SIZE as is; KB =
                                      1024 B), average duration from the kernel-trace pass -> achieved HBM GB/s
  profiles/r03_sq_counters.json       per kernel: matrix-pipe busy share, wait share, VALU / MFMA instruction counts,
                                      LDS busy / bank-conflict / wait shares (two SQ passes)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{tag}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(k):
    return k.replace("mctsq::", "").replace("void ", "").split("(")[0]


def mean(v):
    v = v[2:] if len(v) > 4 else v          # first launches of a kernel are cold
    return sum(v) / max(1, len(v))


dur = {}
for f in glob.glob(f"{tag}/stats_bench/*/*_kernel_stats.csv"):
    shutil.copy(f, os.path.join(ROOT, "profiles", "r03_bench_kernel_stats.csv"))
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
if os.path.exists(f"{tag}/bench.json"):
    line = open(f"{tag}/bench.json").read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(ROOT, "profiles", "r03_bench.json"), "w").write(line + "\n")

fetch, write = counters("pmc_FETCH_SIZE"), counters("pmc_WRITE_SIZE")
traffic = {"correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE as "
                         "is; KB = 1024 B; first two launches of every kernel dropped (cold)",
           "commands": "tools/profile_r03.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate runs) -- python3 bench.py "
                       "--no-cpu-baseline --no-trees --steps 10 --warmup 2 --settle-ms 0; durations from rocprofv3 "
                       "--kernel-trace --stats of the same program (--steps 50 --warmup 5)", "kernels": {}}
for k in sorted(fetch):
    if "mctsq" not in k:
        continue
    fv, wv = mean(fetch[k]["FETCH_SIZE"]), mean(write.get(k, {}).get("WRITE_SIZE", [0.0]))
    hbm = (2 * fv + wv) * 1024
    us = dur.get(k, (None, 0))[0]
    traffic["kernels"][short(k)] = {"FETCH_SIZE_KB_avg": fv, "WRITE_SIZE_KB_avg": wv, "launches": len(fetch[k]["FETCH_SIZE"]),
                                    "hbm_bytes_per_launch": hbm, "avg_us": us,
                                    "hbm_GBs": (hbm / (us * 1e-6) / 1e9) if us else None}
dom = traffic["kernels"].get("gru3_head_fwd_kernel<false, 4>")
if dom:
    traffic["hbm_bytes_per_launch"] = dom["hbm_bytes_per_launch"]
    traffic["kernel"] = "gru3_head_fwd_kernel<false, 4>"
json.dump(traffic, open(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json"), "w"), indent=1)

s1, s2 = counters("pmc_sq1"), counters("pmc_sq2")
sq = {"commands": "tools/profile_r03.sh: two rocprofv3 --pmc passes (8 SQ counters each) over python3 bench.py --no-cpu-baseline "
                  "--no-trees --steps 10 --warmup 2 --settle-ms 0",
      "definitions": {"mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): share of SIMD time the matrix pipe is busy",
                      "wait_any_frac": "SQ_WAIT_ANY / SQ_WAVE_CYCLES", "valu_per_mfma": "SQ_INSTS_VALU / SQ_INSTS_MFMA (VALU count includes MFMA)",
                      "lds_wait_frac": "SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES (pass 2)",
                      "lds_bank_conflict_frac": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"}, "kernels": {}}
for k in sorted(s1):
    if "mctsq" not in k:
        continue
    a = {c: mean(v) for c, v in s1[k].items()}
    b = {c: mean(v) for c, v in s2.get(k, {}).items()}
    rec = dict(a)
    rec.update(b)
    if a.get("SQ_BUSY_CU_CYCLES"):
        rec["mfma_busy_frac"] = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * a["SQ_BUSY_CU_CYCLES"])
    if a.get("SQ_WAVE_CYCLES"):
        rec["wait_any_frac"] = a.get("SQ_WAIT_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_INST_LDS" in b:
            rec["lds_wait_frac"] = b["SQ_WAIT_INST_LDS"] / a["SQ_WAVE_CYCLES"]
    if a.get("SQ_INSTS_MFMA"):
        rec["valu_per_mfma"] = a.get("SQ_INSTS_VALU", 0.0) / a["SQ_INSTS_MFMA"]
    if b.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_frac"] = b.get("SQ_LDS_BANK_CONFLICT", 0.0) / b["SQ_LDS_IDX_ACTIVE"]
    rec["launches"] = len(next(iter(s1[k].values())))
    sq["kernels"][short(k)] = rec
dom = sq["kernels"].get("gru3_head_fwd_kernel<false, 4>")
if dom:
    sq["mfma_busy_frac"] = dom.get("mfma_busy_frac")
    sq["kernel"] = "gru3_head_fwd_kernel<false, 4>"
json.dump(sq, open(os.path.join(ROOT, "profiles", "r03_sq_counters.json"), "w"), indent=1)
for name in ("gru3_head_fwd_kernel<false, 4>", "gru3_head_fwd_pair_kernel<2>", "gru_bwd_pair_kernel<2, 4>", "tf_ffn_pair_kernel<4>",
             "tf_attn_pair_kernel", "conv_enc_fwd_kernel<false>", "tf_ffn_bwd_kernel", "head_bwd_bf16_kernel"):
    r = sq["kernels"].get(name, {})
    t = traffic["kernels"].get(name, {})
    print(f"{name:32s} mfma_busy {r.get('mfma_busy_frac')}  wait {r.get('wait_any_frac')}  lds_wait {r.get('lds_wait_frac')}  "
          f"bank_conf {r.get('lds_bank_conflict_frac')}  us {t.get('avg_us')}  hbm MB {t.get('hbm_bytes_per_launch', 0) / 1e6:.2f}")
