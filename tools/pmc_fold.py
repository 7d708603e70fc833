#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs, csv output) into the traffic JSONs bench.py reads.
    python tools/pmc_fold.py msm <fetch.csv> <write.csv> <log_n> <window_bits> <table: 0|1>   -> profiles/r02_pmc_traffic.json
    python tools/pmc_fold.py ntt <fetch.csv> <write.csv> <log_n>                               -> profiles/r02_pmc_traffic_ntt.json
Units and the gfx950 caveat follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): counter value x 1024 B; FETCH_SIZE
reads exactly half of a wide coalesced stream on gfx950, other patterns uncalibrated, so read-side bytes are a lower bound (<= 2x)."""
import collections
import csv
import json
import sys

UNITS = ("counter value x 1024 bytes (KB).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B read requests at 64 B, i.e. it reports "
         "exactly 1/2 of the bytes of a wide coalesced read stream -- 'double it before comparing with a byte count'.  *_corrected = 2 x FETCH_SIZE + WRITE_SIZE; "
         "*_raw = FETCH_SIZE + WRITE_SIZE.  Calibration inside this very trace: k_msm_recode_wide reads the 32-B scalars once (2^log_n x 32 B) and "
         "k_bases_inf_flags the 128-B bases once -- their FETCH_SIZE is half of that to within 0.1 %.  The gathers of k_msm_accumulate (one lane, one 128-B "
         "line) are requests of the same kind; WRITE_SIZE is taken as counted.")


def rows(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["Counter_Value"]) * 1024.0))
    return sorted(out)


def per_kernel_max(rs):
    best = collections.defaultdict(float)
    for _, k, v in rs:
        best[k] = max(best[k], v)
    return best


def main():
    mode, fpath, wpath = sys.argv[1], sys.argv[2], sys.argv[3]
    f, w = rows(fpath, "FETCH_SIZE"), rows(wpath, "WRITE_SIZE")
    if mode == "msm":
        fm, wm = per_kernel_max(f), per_kernel_max(w)
        kernels = {k: {"FETCH_SIZE_bytes_max_launch": fm.get(k, 0.0), "WRITE_SIZE_bytes_max_launch": wm.get(k, 0.0)} for k in sorted(set(fm) | set(wm))}
        acc = [k for k in kernels if k.startswith("k_msm_accumulate")]
        out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -f csv -- python tools/msm_one.py <log_n> 0 <table c | -1> 1 (separate passes)",
               "units": UNITS, "log_n": int(sys.argv[4]), "window_bits": int(sys.argv[5]), "precomputed_table": bool(int(sys.argv[6])),
               "k_msm_accumulate_traffic_bytes": sum(2.0 * kernels[k]["FETCH_SIZE_bytes_max_launch"] + kernels[k]["WRITE_SIZE_bytes_max_launch"] for k in acc),
               "k_msm_accumulate_traffic_bytes_raw": sum(kernels[k]["FETCH_SIZE_bytes_max_launch"] + kernels[k]["WRITE_SIZE_bytes_max_launch"] for k in acc),
               "algorithmic_bytes": 128.0 * (1 << int(sys.argv[4])), "kernels": kernels}
    else:
        # one transform = the last P consecutive k_ntt_pass dispatches of a direction; take the final forward transform of the run:
        # dispatch order is forward passes then inverse passes per repetition
        fp = [(d, k, v) for d, k, v in f if k.startswith("k_ntt_pass")]
        wp = [(d, k, v) for d, k, v in w if k.startswith("k_ntt_pass")]
        log_n = int(sys.argv[4])
        P = 1 if log_n <= 10 else min(4, (log_n + 7) // 8)
        last_inv_f, last_inv_w = fp[-P:], wp[-P:]
        last_fwd_f, last_fwd_w = fp[-2 * P:-P], wp[-2 * P:-P]
        fwd = 2.0 * sum(v for _, _, v in last_fwd_f) + sum(v for _, _, v in last_fwd_w)
        inv = 2.0 * sum(v for _, _, v in last_inv_f) + sum(v for _, _, v in last_inv_w)
        fwd_raw = sum(v for _, _, v in last_fwd_f) + sum(v for _, _, v in last_fwd_w)
        out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -f csv -- python tools/ntt_one.py <log_n> 2 (separate passes)",
               "units": UNITS, "log_n": log_n, "passes": P,
               "traffic_bytes_per_transform": fwd, "traffic_bytes_inverse_transform": inv, "traffic_bytes_per_transform_raw": fwd_raw,
               "per_pass_forward": [{"kernel": k, "FETCH_SIZE_bytes": v, "WRITE_SIZE_bytes": wv} for (_, k, v), (_, _, wv) in zip(last_fwd_f, last_fwd_w)],
               "algorithmic_bytes": 64.0 * (1 << log_n)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
