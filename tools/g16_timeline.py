#!/usr/bin/env python3
"""Timeline of the last Groth16 prove in a rocprofv3 --kernel-trace csv: per kernel start / end (ms) and queue, kernels > 0.1 ms."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = [r for r in rows if "k_msm_accumulate" in r["Kernel_Name"] and "G2" in r["Kernel_Name"]]
big = max(acc, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))  # the large circuit's G2 accumulation
last = [r for r in acc if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 0.5 * (int(big["End_Timestamp"]) - int(big["Start_Timestamp"]))][-1]
t_end = int(last["End_Timestamp"]) + 6_000_000
t_beg = int(last["Start_Timestamp"]) - 8_000_000
sel = [r for r in rows if t_beg <= int(r["Start_Timestamp"]) <= t_end]
base = int(sel[0]["Start_Timestamp"])
for r in sel:
    s = (int(r["Start_Timestamp"]) - base) / 1e6
    e = (int(r["End_Timestamp"]) - base) / 1e6
    if e - s > 0.25:
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
        nm = nm.replace("G2Cfg<BLS12_381_G2, BLS12_381_Fq, BLS12_381_Fr, Fp2LT<Fp28<BLS12_381_Fq28, BLS12_381_Fq>, false>, 255, 6, 2>", "G2")
        print(f"{nm[:34]:34s} q={r['Queue_Id']:>2} {s:8.2f} -> {e:8.2f} ({e - s:6.2f})")
