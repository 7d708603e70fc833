#!/usr/bin/env python3
"""Fold one rocprofv3 --pmc pass of SQ counters (csv) into a per-kernel JSON: max over launches per kernel, plus issue / stall fractions.
    python tools/pmc_sq.py <counter_collection.csv> "<command>" > profiles/rNN_pmc_sq.json
SQ_* are quad-cycles summed over waves (/opt/skills/guides/MI355X_MICROARCH.md): WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES."""
import collections
import csv
import json
import sys

best = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    c, v = r["Counter_Name"], float(r["Counter_Value"])
    best[k][c] = max(best[k].get(c, 0.0), v)
out = {"command": sys.argv[2] if len(sys.argv) > 2 else "", "kernels": {}}
for k, d in sorted(best.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    e = dict(d)
    e["frac_issuing"] = d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    e["frac_issue_stalled"] = d.get("SQ_WAIT_INST_ANY", 0.0) / wc
    e["frac_parked_on_waitcnt"] = d.get("SQ_WAIT_ANY", 0.0) / wc
    e["frac_valu_of_issuing"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / max(1.0, d.get("SQ_ACTIVE_INST_ANY", 0.0))
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
