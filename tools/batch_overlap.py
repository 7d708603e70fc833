#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv of `BATCH=K PRE=22 tools/msm_sweep.py 24`: gaps between consecutive accumulation kernels of the
last batch and the longest side kernel, to see whether the sort / tail phases really hide behind the accumulation."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = sorted([r for r in rows if "k_msm_accumulate" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))[-5:]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e6 for a, b in zip(acc, acc[1:])]
durs = [(int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e6 for a in acc]
t0, t1 = int(acc[0]["Start_Timestamp"]), int(acc[-1]["End_Timestamp"])
side = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1 and "accumulate" not in r["Kernel_Name"]]
worst = sorted(side, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))[-3:]
print("acc durations", [f"{d:.1f}" for d in durs], "gaps", [f"{g:.2f}" for g in gaps])
for r in worst:
    print("   longest side kernel:", r["Kernel_Name"].split("(")[0][:40], f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.1f} ms")
