#!/usr/bin/env python3
"""Per-launch timeline of the LAST burst of kernels in a rocprofv3 (rocpd sqlite) kernel trace: start offset, duration and the idle gap
in front of every launch (us).  A burst = launches separated by less than `gap_us` of idle time; the last burst of the trace is usually
the last call of the driving script.
    python tools/timeline.py gpurun_out/prof/x_results.db [gap_us=300] [burst_index_from_end=1] [only launches of at least min_us]"""
import sqlite3
import sys


def main(path, gap_us=300.0, which=1, min_us=0.0):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.cursor().execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else "0")
    rows = db.cursor().execute(f"select name, start, end, grid_x, workgroup_x, {q} from kernels order by start").fetchall()
    if not rows:
        print("no kernels")
        return
    bursts, cur = [], [rows[0]]
    for r in rows[1:]:
        last_end = max(x[2] for x in cur)
        if (r[1] - last_end) / 1e3 > gap_us:
            bursts.append(cur)
            cur = []
        cur.append(r)
    bursts.append(cur)
    b = bursts[-which]
    t0 = b[0][1]
    busy = 0.0
    prev_end = t0
    print(f"# burst {len(bursts) - which + 1} of {len(bursts)}: {len(b)} launches, span {(max(x[2] for x in b) - t0) / 1e3:.1f} us")
    print(f"{'kernel':<52} {'q':>3} {'start_us':>10} {'dur_us':>9} {'gap_us':>8} {'grid':>9} {'wg':>5}")
    for name, s, e, grid, wg, q in b:
        nm = name.split("(")[0].replace("void ", "")
        nm = nm.replace("G2Cfg<BLS12_381_G2, BLS12_381_Fq, BLS12_381_Fr, Fp2LT<Fp28<BLS12_381_Fq28, BLS12_381_Fq>, false>, 255, 6, 2>", "BlsG2")
        nm = nm.replace("G2Cfg<BN254_G2, BN254_Fq, BN254_Fr, Fp2LT<Fp28<BN254_Fq28, BN254_Fq>, false>, 254, 4, 3>", "BnG2")
        gap = (s - prev_end) / 1e3
        if (e - s) / 1e3 >= min_us:
            print(f"{nm[:52]:<52} {q:>3} {(s - t0) / 1e3:>10.1f} {(e - s) / 1e3:>9.1f} {gap:>8.1f} {grid:>9} {wg:>5}")
        busy += (e - s) / 1e3
        prev_end = max(prev_end, e)
    print(f"# sum of kernel durations {busy:.1f} us")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], float(a[2]) if len(a) > 2 else 300.0, int(a[3]) if len(a) > 3 else 1, float(a[4]) if len(a) > 4 else 0.0)
