#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 --kernel-trace csv: python3 tools/timeline.py <kernel_trace.csv> [step]
Prints start / end (us, relative to the step's first kernel), duration and stream of every kernel of one
steady-state PER-DQN step (steps are delimited by per_sample_kernel)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
want = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("per_sample_kernel") or "per_sample_kernel" in r["Kernel_Name"]]
a, b = starts[want], starts[want + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0][:48]
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
