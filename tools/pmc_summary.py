#!/usr/bin/env python3
"""Fold the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv output) into profiles/rNN_pmc_traffic*.json.
    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <window_bits> > out.json
Units and the gfx950 caveat follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): counter value x 1024 B; FETCH_SIZE
reads exactly half of a wide coalesced stream on gfx950, other patterns uncalibrated, so read-side bytes are a lower bound (<= 2x)."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    best = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        best[k] = max(best[k], float(r["Counter_Value"]) * 1024.0)
    return best


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {k: {"FETCH_SIZE_bytes_max_launch": f.get(k, 0.0), "WRITE_SIZE_bytes_max_launch": w.get(k, 0.0)} for k in sorted(set(f) | set(w))}
    acc = [k for k in kernels if k.startswith("k_msm_accumulate")]
    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -f csv -- python bench.py --steps 1 --warmup 0 --no-cpu --no-ntt --no-skew --groth16-k 0 (separate passes)",
        "units": "counter value x 1024 bytes (KB); gfx950 note (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reads exactly 1/2 of a wide coalesced "
                 "stream, other patterns uncalibrated -> read-side bytes below are a lower bound, at most 2x higher",
        "workload": "bls12_381_g1_msm_2^24, precomputed table (bench.py default)",
        "kernels": kernels,
        "k_msm_accumulate_traffic_bytes": sum(kernels[k]["FETCH_SIZE_bytes_max_launch"] + kernels[k]["WRITE_SIZE_bytes_max_launch"] for k in acc),
        "window_bits": int(sys.argv[3]),
    }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
