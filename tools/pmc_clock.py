#!/usr/bin/env python3
"""Effective clock per kernel from one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass (csv): GRBM_GUI_ACTIVE / wall time of the dispatch
(MI355X_MICROARCH.md, "DVFS give-back").  The counter is reported summed over the XCDs it was sampled on; the per-XCD value is printed for 1 and 8.
    python tools/pmc_clock.py <dir with *_counter_collection.csv and *_kernel_trace.csv> "<command>" > profiles/rNN_pmc_clock_*.json"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
rows = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    t = dur.get(r["Dispatch_Id"])
    if t is None and "Start_Timestamp" in r:
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    if t and t > 0:
        rows[k].append((float(r["Counter_Value"]), t))
out = {"command": sys.argv[2] if len(sys.argv) > 2 else "", "note": "ghz_if_summed_over_8_xcds = GRBM_GUI_ACTIVE / 8 / wall; ghz_if_one_instance = GRBM_GUI_ACTIVE / wall", "kernels": {}}
for k, v in sorted(rows.items(), key=lambda kv: -max(t for _, t in kv[1])):
    big = [x for x in v if x[1] >= 0.5 * max(t for _, t in v)]
    c = sum(x[0] for x in big)
    t = sum(x[1] for x in big)
    out["kernels"][k] = {"launches": len(big), "avg_ms": t / len(big) * 1e3, "GRBM_GUI_ACTIVE_avg": c / len(big), "ghz_if_one_instance": c / t / 1e9, "ghz_if_summed_over_8_xcds": c / t / 8e9}
print(json.dumps(out, indent=1))
