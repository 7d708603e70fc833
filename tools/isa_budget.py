#!/usr/bin/env python3
"""Itemised instruction budget of ONE mixed addition of k_msm_accumulate<BlsG1> from the ISA hipcc emits (VERDICT r4 "next" item 1b).
    python tools/isa_budget.py > profiles/r05_acc_instruction_budget.txt
Compiles openzl_amd/csrc/zl_msm_acc.hip exactly as build.py does (device only, -S), cuts the kernel into basic blocks, walks the blocks a wave executes in
one iteration of the chunk loop and classifies every instruction.  VALU issue cycles per wave64 instruction on gfx950 (profiles/r01_ubench_valu_rates.log,
re-read at the measured clock): 4 for v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32 / 64-bit shifts, adds and moves / v_mad_i64_i32, 2 for the 32-bit rest."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openzl_amd import build as zb  # noqa: E402

FOUR = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_lshl_add_u64", "v_mad_i64_i32", "v_mov_b64_e32", "v_mov_b64"}


def compile_s(group="BlsG1"):
    defs = next(d for n, s, d in zb._units() if n == f"zl_msm_acc_{group}")
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "acc.s")
    cmd = [zb._hipcc()] + zb.FLAGS + defs + ["--cuda-device-only", "-S", os.path.join(zb.CSRC, "zl_msm_acc.hip"), "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out, " ".join(cmd)


def blocks_of(path, kernel_prefix):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel_prefix) and l.rstrip().split(":")[0].endswith("j") and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = collections.OrderedDict(), None
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        m2 = re.match(r"^; %bb\.(\d+):", s)
        if m or m2:
            cur = m.group(1) if m else "%bb." + m2.group(1)
            blocks[cur] = []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        if cur is None:
            cur = "entry"
            blocks[cur] = []
        blocks[cur].append(s)
    meta = {}
    for l in lines[end:end + 120]:
        for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "codeLenInByte"):
            m = re.match(r"^; %s: (\d+)" % key, l.strip())
            if m and key not in meta:
                meta[key] = int(m.group(1))
    return blocks, meta


def classify(ins, prev):
    op = ins.split()[0]
    if op == "v_mad_u64_u32":
        return "mad (v_mad_u64_u32)"
    if op == "s_nop":
        return "s_nop (hipcc pads every inline-asm statement whose result is read next)"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "scalar ALU / branch / exec mask"
    if op.startswith("global_load") or op.startswith("global_store") or op.startswith("scratch_"):
        return "global load / store"
    if op == "v_mul_lo_u32":
        return "Montgomery factor m_k = lo * INV (v_mul_lo_u32)"
    if op == "v_lshrrev_b64":
        return "column shift acc >>= 28 (v_lshrrev_b64)"
    if op in ("v_mad_i64_i32", "v_ashrrev_i64", "v_lshl_add_u64", "v_mul_hi_u32"):
        return "weak reduction of the P == +-Q test (64-bit signed chain)"
    if op.startswith("v_and_b32") and "0xfffffff" in ins:
        return "28-bit masks (m_k, result limbs, carry passes)"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "register moves (loop-carried values, infinity, selects)"
    if op.startswith("v_cndmask"):
        return "selects (v_cndmask)"
    if op.startswith("v_cmp") or op.startswith("v_or") or op.startswith("v_xor"):
        return "compares / or-reductions (infinity, zero tests, loop)"
    if op.startswith("v_lshrrev_b32") or op.startswith("v_add3_u32") or op.startswith("v_alignbit"):
        return "carry passes (shift, add3, alignbit)"
    if op.startswith("v_sub") or op.startswith("v_add") or op.startswith("v_lshl_add_u32") or op.startswith("v_lshlrev_b32"):
        return "limb additions / subtractions / doublings (incl. address arithmetic)"
    return "other VALU (" + op + ")"


def cycles(ins):
    op = ins.split()[0]
    if not op.startswith("v_"):
        return 0
    return 4 if op in FOUR else 2


def main():
    path, cmd = compile_s()
    blocks, meta = blocks_of(path, "_Z16k_msm_accumulateI5BlsG1E")
    names = list(blocks)
    mads = {n: sum(1 for i in blocks[n] if i.startswith("v_mad_u64_u32")) for n in names}
    # roles: found from the structure, not from label numbers (they move between compiler versions)
    first = next(n for n in names if mads[n] == 784)            # u2 = qx zz, s2 = qy zzz (+ the differences and the weak reduction of the zero test)
    main_b = next(n for n in names if mads[n] == 2758)          # sqr, 3 mul, sqr, muladd, 2 mul of the general case
    dbl_b = next((n for n in names if mads[n] == 2275), None)   # P == Q: dbl_affine (never on random inputs)
    i_first, i_main = names.index(first), names.index(main_b)
    assert mads[first] + mads[main_b] == 3542
    loop_head = next(n for n in names if any("s_cbranch" in i for i in blocks[n]) and names.index(n) < i_first and any(first == x for x in []) is False and n.startswith(".LBB") and
                     any(i.startswith("v_cmp_eq_u32") for i in blocks[n]) and names.index(n) > names.index(next(m for m in names if len(blocks[m]) > 100)))
    i_head = names.index(loop_head)
    # latch: the labelled blocks between the prologue and the loop head (layout puts them in front of the header)
    latch = [n for n in names[:i_head] if n.startswith(".LBB") and names.index(n) > names.index(next(m for m in names if len(blocks[m]) > 100))]
    stores = [n for n in names[i_head:i_first] if any(i.startswith("global_store") for i in blocks[n])]
    load_b = next(n for n in names[i_head:i_first] if sum(1 for i in blocks[n] if i.startswith("global_load")) >= 8)
    i_load = names.index(load_b)
    flush = names[i_head + 1:i_load]
    pre = names[i_load:i_first]
    between = names[i_first + 1:i_main]
    after_main = names[i_main + 1:]
    inf_case = next((n for n in after_main if len(blocks[n]) >= 40 and not any(i.startswith("v_mad") for i in blocks[n]) and any(i.startswith("s_branch") for i in blocks[n])), None)

    def histo(block_names):
        h = collections.OrderedDict()
        for n in block_names:
            prev = ""
            for ins in blocks[n]:
                c = classify(ins, prev)
                e = h.setdefault(c, [0, 0])
                e[0] += 1
                e[1] += cycles(ins)
                prev = ins
        return h

    # Walk one iteration as a wave with NO lane at a bucket boundary and no lane in a special case executes it: conditional branches around the flush
    # blocks, the low-limb-filter hits (P == +-Q candidates: 2^-26 per addition), the doubling and the "accumulator was infinity" block are taken.
    def target(ins):
        return ins.split()[-1]

    def walk(n, stop):
        path = []
        while n is not None and n not in path:
            path.append(n)
            if n == stop:
                break
            nxt = names[names.index(n) + 1] if names.index(n) + 1 < len(names) else None
            idx = names.index(n)
            for ins in blocks[n]:
                if ins.startswith("s_branch"):
                    nxt = target(ins)
                elif ins.startswith("s_cbranch_execz") and (idx < i_load or idx > i_main):
                    nxt = target(ins)  # skip: flush blocks (before the loads), special cases (after the general case)
                    break
                elif ins.startswith("s_cbranch_execz") and i_first < idx < i_main and len(blocks[n]) <= 8:
                    nxt = target(ins)  # low-limb filter did not hit: skip the full comparison
                    break
            n = nxt
        return path

    back = next(n for n in names if any(i.startswith("s_cbranch_execz") for i in blocks[n]) and any(i.startswith("v_add_u32_e32") and " 1, " in i for i in blocks[n]))  # e++, exit test
    always = [n for n in walk(loop_head, back) if n != dbl_b]
    # what the walk skipped on the way that a wave with a lane at a boundary (or a lane whose accumulator is infinity) executes in addition
    after_join = names[names.index(always[always.index(main_b) + 1]):names.index(back)]
    cond_tail = [n for n in after_join if n not in always and names.index(n) > (names.index(dbl_b) if dbl_b else i_main) and not any(i.startswith("v_cmp_eq") for i in blocks[n])]
    latch_moves = [n for n in latch if n not in always]
    print("# Instruction budget of one iteration of k_msm_accumulate<BlsG1>'s chunk loop = one mixed addition (XYZZ += affine), from the ISA.")
    print("# " + cmd)
    print("# kernel: %s" % ", ".join(f"{k} {v}" for k, v in meta.items()))
    print("# VALU issue cycles per wave64 instruction: 4 = v_mad_u64_u32, v_mul_lo/hi_u32, 64-bit shift / add / move, v_mad_i64_i32; 2 = 32-bit VALU; 0 = scalar, s_nop, s_waitcnt, memory issue")
    print()
    print("## Basic blocks (layout order)")
    print(f"{'block':10s} {'instr':>6s} {'mads':>6s} {'VALU cyc':>9s}  role")
    role = {loop_head: "loop head: does any lane cross into its next bucket?", first: "u2 = qx zz, s2 = qy zzz; P - X1, R - Y1 (carried); weak reduction + low-limb filter of the P == +-Q test",
            main_b: "general case: PP, PPP, Q, R^2, X3, Y3 (one dual scan), ZZ3, ZZZ3", load_b: "entry word, then the 128-B base line (8 x dwordx4), s_waitcnt"}
    if dbl_b:
        role[dbl_b] = "P == Q: dbl_affine (not on the hot path)"
    if inf_case:
        role[inf_case] = "accumulator was infinity: acc = (qx, +-qy, 1, 1) (first addition after a flush)"
    for n in flush:
        role.setdefault(n, "bucket boundary: store the finished sum (bucket_sums / partials), next offsets word, acc = infinity")
    for n in latch:
        role.setdefault(n, "loop latch (e++, exit test)" if n in always else "join after the infinity case: the loop-carried accumulator moves into its registers")
    for n in names:
        cyc = sum(cycles(i) for i in blocks[n])
        print(f"{n:10s} {len(blocks[n]):6d} {mads[n]:6d} {cyc:9d}  {role.get(n, '')}")
    print()
    h = histo(always)
    tot_i = sum(v[0] for v in h.values())
    tot_c = sum(v[1] for v in h.values())
    print("## Blocks every iteration executes: " + " ".join(always))
    print(f"{'category':92s} {'instr':>6s} {'VALU cyc':>9s} {'% cyc':>6s}")
    for c, (ni, nc) in sorted(h.items(), key=lambda kv: -kv[1][1]):
        print(f"{c:92s} {ni:6d} {nc:9d} {100.0 * nc / tot_c:6.2f}")
    print(f"{'total':92s} {tot_i:6d} {tot_c:9d} {100.0:6.2f}")
    nm = h["mad (v_mad_u64_u32)"]
    valu_nonmad = sum(v[0] for k, v in h.items() if v[1] > 0) - nm[0]
    print()
    print(f"mads {nm[0]} = 6 mul x 392 + 2 sqr x 301 + 1 dual scan x 588; non-mad VALU instructions {valu_nonmad} ({tot_c - nm[1]} cycles = {100.0 * (tot_c - nm[1]) / tot_c:.1f} % of the VALU cycles);")
    print(f"scalar / s_nop / s_waitcnt / memory-issue instructions {tot_i - nm[0] - valu_nonmad} (no VALU cycles; they share the wave's issue slot, hidden behind the other two waves of the SIMD)")
    print()
    print("## Inside the nine product scans (generator openzl_amd/csrc/gen_mul28.py; per scan: 14 m_k, 28 column shifts, 28 masks)")
    per = {"Montgomery factor m_k = lo * INV (v_mul_lo_u32)": (9 * 14, 4), "column shift acc >>= 28 (v_lshrrev_b64)": (9 * 28, 4), "28-bit masks of m_k and of the result limbs": (9 * 28, 2)}
    s_i = s_c = 0
    for k, (ni, w) in per.items():
        print(f"{k:92s} {ni:6d} {ni * w:9d} {100.0 * ni * w / tot_c:6.2f}")
        s_i += ni
        s_c += ni * w
    print(f"{'multiplier bookkeeping, total':92s} {s_i:6d} {s_c:9d} {100.0 * s_c / tot_c:6.2f}")
    print(f"{'everything else that is not a mad (differences, X3, sign, zero test, loop, moves)':92s} {valu_nonmad - s_i:6d} {tot_c - nm[1] - s_c:9d} {100.0 * (tot_c - nm[1] - s_c) / tot_c:6.2f}")
    print()
    cond = [n for n in flush if n not in always] + [n for n in cond_tail if n not in flush] + [n for n in latch_moves if n not in cond_tail]
    hc = histo(cond)
    ci = sum(v[0] for v in hc.values())
    cc = sum(v[1] for v in hc.values())
    print("## Blocks a wave executes when ANY of its 64 lanes is at a bucket boundary (1 - (1 - 1/64)^64 = 63 % of the iterations at 64 entries per bucket): " + " ".join(cond))
    for c, (ni, nc) in sorted(hc.items(), key=lambda kv: -kv[1][1]):
        print(f"{c:92s} {ni:6d} {nc:9d}")
    print(f"{'total (x 0.63 per iteration)':92s} {ci:6d} {cc:9d}   -> {0.63 * cc:.0f} cycles per iteration = {100.0 * 0.63 * cc / tot_c:.2f} %")
    print()
    print(f"## One iteration = {tot_c} VALU issue cycles per wave (+ {0.63 * cc:.0f} for the boundary blocks) on a SIMD that issues one wave's VALU instruction at a time.")
    print("# With E entries, 1024 SIMDs and clock f: t >= E / 64 / 1024 * cycles / f (profiles/README.md compares it with the measured kernel time and clock).")


if __name__ == "__main__":
    main()
