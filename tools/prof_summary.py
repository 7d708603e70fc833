#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max (us) + registers.
    python tools/prof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt"""
import sqlite3
import sys


import re


def short(name: str) -> str:
    """kernel name without its argument list; the G2 group configurations abbreviated (BlsG2 / BnG2) so that the columns stay aligned"""
    n = name.split("(")[0].replace("void ", "")
    n = re.sub(r"G2Cfg<BLS12_381_G2,.*?, 255, 6, 2> ?", "BlsG2", n)
    n = re.sub(r"G2Cfg<BN254_G2,.*?, 254, 4, 3> ?", "BnG2", n)
    return n.replace(" >", ">")


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
        "max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    # average over the "full-size" launches of each kernel (duration > half of its max): the bench also issues small launches
    # of the same kernels (2^14 self-check MSM, Groth16's 2^20 MSMs), which would dilute a plain average
    big = {r[0]: (r[1], r[2]) for r in cur.execute(
        "select k.name, count(*), avg(k.duration) from kernels k join (select name, max(duration) as m from kernels group by name) t "
        "on k.name = t.name where k.duration > 0.5 * t.m group by k.name")}
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    # rocprofv3's vgpr_count is HALF the allocation of a wave64 kernel on this stack (checked against hipcc -S: k_msm_accumulate<BlsG1>
    # NumVgprs 166 -> 84 here; <BlsG2> 256 + 172 AGPRs -> 216): print the allocation
    print("# regs = registers allocated per lane (VGPR + AGPR, unified file) = 2 x the vgpr_count rocprofv3 reports on this stack for wave64 kernels "
          "(checked against hipcc -S: k_msm_accumulate<BlsG1> NumVgprs 166 -> rocprofv3 84; <BlsG2> 256 + 172 AGPRs -> 216); waves per SIMD = floor(512 / regs)")
    print(f"{'kernel':<70} {'calls':>6} {'total_us':>12} {'avg_us':>11} {'min_us':>11} {'max_us':>11} {'big_n':>5} {'big_avg_us':>11} {'%':>6} {'regs':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>7} {'grid':>10} {'wg':>5}")
    for r in rows:
        name = short(r[0])
        print(f"{name:<70} {r[1]:>6} {r[2]/1e3:>12.1f} {r[3]/1e3:>11.1f} {r[4]/1e3:>11.1f} {r[5]/1e3:>11.1f} {big[r[0]][0]:>5} {big[r[0]][1]/1e3:>11.1f} {100*r[2]/total:>6.2f} {2*r[6]:>5} {r[7]:>5} {r[8]:>5} {r[9]:>7} {r[10]:>7} {r[11]:>10} {r[12]:>5}")


def launches(path, substr, limit=40):
    """individual launches of the kernels whose name contains `substr`, in start order (e.g. the levels of the bucket reduction)"""
    db = sqlite3.connect(path)
    rows = db.cursor().execute("select name, start, duration, grid_x, workgroup_x from kernels where name like ? order by start limit ?", (f"%{substr}%", limit)).fetchall()
    print(f"# launches of *{substr}* in start order")
    for name, start, dur, grid, wg in rows:
        print(f"{short(name):<60} {dur/1e3:>10.1f} us  grid {grid:>9} wg {wg}")


if __name__ == "__main__":
    main(sys.argv[1])
    for sub in sys.argv[2:]:
        launches(sys.argv[1], sub)
