#!/usr/bin/env python3
"""Itemised instruction budget of k_ntt_pass28 (the middle pass of a 2^24 transform: BLS12-381 Fr on nine 29-bit limbs) from the ISA hipcc emits
(VERDICT r5 "next" item 5).
    python tools/isa_budget_ntt.py > profiles/r06_ntt_instruction_budget.txt
Compiles openzl_amd/csrc/zl_ntt.hip as build.py does (device only, -S), cuts the kernel into basic blocks, finds the three phases of a pass by their
structure -- load (global load + inter-pass twiddle product + LDS store), the two-stage butterfly quad (LDS load, <= 4 products, LDS store), store (LDS
load + global store) -- and classifies every instruction.  VALU issue cycles per wave64 instruction as in tools/isa_budget.py (4 = v_mad_u64_u32 /
v_mul_lo/hi_u32 / 64-bit shifts, adds, moves / v_mad_i64_i32; 2 = the 32-bit rest); LDS and global instructions are counted, not costed."""
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
KERNEL = sys.argv[1] if len(sys.argv) > 1 else "_Z12k_ntt_pass28I12BLS12_381_FrLb0ELi2EE"


def compile_s():
    defs = next(d for n, s, d in zb._units() if n == "zl_ntt")
    out = os.path.join(tempfile.mkdtemp(prefix="isa_ntt_"), "ntt.s")
    cmd = [zb._hipcc()] + zb.FLAGS + defs + ["--cuda-device-only", "-S", os.path.join(zb.CSRC, "zl_ntt.hip"), "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out, " ".join(cmd)


def blocks_of(path):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and ": " in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
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
        blocks[cur].append(s)
    meta = {}
    for l in lines[end:end + 120]:
        for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize", "codeLenInByte"):
            m = re.match(r"^; %s: (\d+)" % key, l.strip())
            if m and key not in meta:
                meta[key] = int(m.group(1))
    return blocks, meta


def classify(ins):
    op = ins.split()[0]
    if op == "v_mad_u64_u32":
        return "mad (v_mad_u64_u32): the product scans"
    if op == "s_nop":
        return "s_nop (pad behind an inline-asm statement)"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op == "s_barrier":
        return "s_barrier"
    if op.startswith("s_"):
        return "scalar ALU / branch / exec mask"
    if op.startswith("ds_"):
        return "LDS instruction (ds_read2st64_b32 / ds_write2st64_b32: two limbs each)"
    if op.startswith("global_load"):
        return "global load"
    if op.startswith("global_store"):
        return "global store"
    if op == "v_mul_lo_u32":
        return "Montgomery factor m_k = lo * INV (v_mul_lo_u32)"
    if op == "v_lshrrev_b64":
        return "column shift of a scan (v_lshrrev_b64)"
    if op in ("v_mad_i64_i32", "v_ashrrev_i64", "v_mul_hi_u32", "v_lshl_add_u64", "v_lshlrev_b64"):
        return "weak reduction wred (quotient estimate, signed 64-bit chain) / 64-bit address arithmetic"
    if op.startswith("v_and_b32"):
        return "29-bit masks (m_k, result limbs, carry passes)"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "register moves"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"):
        return "selects / compares"
    if op.startswith("v_lshrrev_b32") or op.startswith("v_add3_u32") or op.startswith("v_alignbit"):
        return "carry passes (shift, add3, alignbit)"
    if op.startswith("v_sub") or op.startswith("v_add") or op.startswith("v_lshl_add_u32") or op.startswith("v_lshlrev_b32") or op.startswith("v_and_or") or op.startswith("v_or"):
        return "limb additions / biased subtractions / LDS and global address arithmetic"
    return "other VALU (" + op + ")"


def cycles(ins):
    op = ins.split()[0]
    if not op.startswith("v_"):
        return 0
    return 4 if op in FOUR else 2


def histo(blocks, names):
    h = collections.OrderedDict()
    for n in names:
        for ins in blocks[n]:
            c = classify(ins)
            e = h.setdefault(c, [0, 0])
            e[0] += 1
            e[1] += cycles(ins)
    return h


def show(title, h, per):
    tot_i = sum(v[0] for v in h.values())
    tot_c = sum(v[1] for v in h.values())
    print(f"## {title}")
    print(f"{'category':100s} {'instr':>6s} {'VALU cyc':>9s} {'% cyc':>6s}")
    for c, (ni, nc) in sorted(h.items(), key=lambda kv: (-kv[1][1], -kv[1][0])):
        print(f"{c:100s} {ni:6d} {nc:9d} {100.0 * nc / max(tot_c, 1):6.2f}")
    print(f"{'total':100s} {tot_i:6d} {tot_c:9d} {100.0:6.2f}   = {tot_c / per:.0f} VALU issue cycles per element")
    print()
    return tot_i, tot_c


def main():
    path, cmd = compile_s()
    blocks, meta = blocks_of(path)
    names = list(blocks)
    cnt = lambda n, p: sum(1 for i in blocks[n] if i.split()[0].startswith(p))  # noqa: E731
    mads = {n: cnt(n, "v_mad_u64_u32") for n in names}
    lds = {n: cnt(n, "ds_") for n in names}
    bars = [n for n in names if cnt(n, "s_barrier")]
    # quad loop: the blocks between the barrier that ends the load phase and the barrier that closes a round (the one whose block branches back)
    i_q0 = names.index(bars[1])
    i_q1 = next(i for i in range(i_q0 + 1, len(names)) if names[i] in bars)
    quad = names[i_q0 + 1:i_q1]
    q_ld = next(n for n in quad if lds[n] >= 18 and cnt(n, "s_waitcnt") >= 4)
    q_st = next(n for n in quad if lds[n] >= 18 and n != q_ld)
    q_mul = [n for n in quad if mads[n] >= 150]
    q_triv = next(n for n in quad if mads[n] == 0 and len(blocks[n]) > 120 and n not in (q_ld, q_st))
    q_other = [n for n in quad if n not in q_mul and n not in (q_ld, q_st, q_triv) and len(blocks[n]) >= 8 and names.index(n) > names.index(q_ld)]
    general = [q_ld] + q_mul + q_other + [q_st]
    # load phase of a middle pass with a row table: the loop that loads 4 x dwordx4 (nine dwords of the element + the 32-byte twiddle), multiplies, stores to LDS
    # The region between the first two barriers holds every variant of load_elem (first pass: Montgomery entry / coset scaling; middle pass with or without a row
    # table; last pass).  A middle pass with its row table executes: the small address / branch blocks, the nine-dword element load, the 32-byte row twiddle load,
    # ONE product block (no global loads of its own) and the LDS store -- taken here as every block of the region with < 60 instructions plus the first
    # one-product block without global loads (the other product blocks belong to the variants).
    region = names[names.index(bars[0]) + 1:names.index(bars[1])]
    one_prod = next(n for n in region if 150 <= mads[n] < 200 and cnt(n, "global_load") == 0)
    ld_phase = [n for n in region if (len(blocks[n]) < 60 and len(blocks[n]) > 0) or n == one_prod]
    st_phase = [n for n in names[i_q1:] if cnt(n, "global_store") >= 1 and any(i.startswith("s_cbranch_execnz") for i in blocks[n])]
    print("# Instruction budget of k_ntt_pass28<BLS12_381_Fr, LAST = false> (a middle pass; nine 29-bit limbs, 1024-element tiles, 256 lanes), from the ISA.")
    print("# " + cmd)
    print("# kernel: %s" % ", ".join(f"{k} {v}" for k, v in meta.items()))
    print("# A pass of 2^8 points runs: load phase x 4 elements per lane | 4 rounds x one quad (4 elements, two butterfly stages) per lane | store phase x 4 elements per lane;")
    print("# per element that is 1 load iteration + 1 quad + 1 store iteration.")
    print()
    print("## Basic blocks (layout order)")
    role = {q_ld: "quad: LDS load of 4 elements (18 x ds_read2st64_b32 + 2 x ds_read_b32), waits", q_st: "quad: LDS store of 4 elements",
            q_triv: "quad, k2 = 0 (trivial twiddles): carry passes + three weak reductions instead of two products"}
    for n in q_mul:
        role[n] = "quad: %d product(s) by a butterfly root (+ the additions / biased subtractions / x0's weak reduction around them)" % round(mads[n] / 162.5)
    for n in ld_phase:
        role[n] = "load phase (middle pass, row table): addresses, element (nine dwords) + row twiddle from global memory, one product, LDS store"
    for n in st_phase:
        role[n] = "store phase: LDS load (bit-reversed row), nine-dword global store"
    for n in bars:
        role.setdefault(n, "s_barrier")
    print(f"{'block':10s} {'instr':>6s} {'mads':>6s} {'LDS':>4s} {'VALU cyc':>9s}  role")
    for n in names:
        if len(blocks[n]) == 0:
            continue
        print(f"{n:10s} {len(blocks[n]):6d} {mads[n]:6d} {lds[n]:4d} {sum(cycles(i) for i in blocks[n]):9d}  {role.get(n, '')}")
    print()
    iq, cq = show("One quad, general case (k2 != 0: four products) = 4 elements x 2 stages: " + " ".join(general), histo(blocks, general), 4.0)
    hl = histo(blocks, ld_phase)
    il, cl = show("Load phase, one element: " + " ".join(ld_phase), hl, 1.0)
    hs = histo(blocks, st_phase)
    is_, cs = show("Store phase, one element: " + " ".join(st_phase), hs, 1.0)
    per_elem = cl + 4 * cq / 4.0 + cs
    hq = histo(blocks, general)
    mad_c = hq["mad (v_mad_u64_u32): the product scans"][1] + hl.get("mad (v_mad_u64_u32): the product scans", [0, 0])[1]
    lds_q = hq[next(k for k in hq if k.startswith("LDS"))][0]
    print("## Per element and pass (s = 8: four two-stage rounds)")
    print(f"VALU issue cycles: load {cl} + 4 rounds x {cq / 4.0:.0f} + store {cs} = {per_elem:.0f}; of these the mads: {hl.get('mad (v_mad_u64_u32): the product scans', [0, 0])[1]} + 4 x {hq['mad (v_mad_u64_u32): the product scans'][1] / 4.0:.0f} "
          f"= {hl.get('mad (v_mad_u64_u32): the product scans', [0, 0])[1] + hq['mad (v_mad_u64_u32): the product scans'][1]:.0f} ({100.0 * (hl.get('mad (v_mad_u64_u32): the product scans', [0, 0])[1] + hq['mad (v_mad_u64_u32): the product scans'][1]) / per_elem:.1f} %)")
    print(f"LDS instructions per element: 4 rounds x {lds_q / 4.0:.1f} (quad load + store) + {hl[next(k for k in hl if k.startswith('LDS'))][0]} (load phase) + {hs[next(k for k in hs if k.startswith('LDS'))][0]} (store phase)"
          f" = {lds_q + hl[next(k for k in hl if k.startswith('LDS'))][0] + hs[next(k for k in hs if k.startswith('LDS'))][0]:.0f}, two dwords each: one LDS instruction per {per_elem / (lds_q + hl[next(k for k in hl if k.startswith('LDS'))][0] + hs[next(k for k in hs if k.startswith('LDS'))][0]):.0f} VALU issue cycles")
    waves = meta.get("Occupancy", 4)
    print(f"At 2^24: 2^24 elements / 64 lanes / 1024 SIMDs = 256 wave-elements per SIMD and pass -> {256 * per_elem:.0f} VALU issue cycles per SIMD and pass = "
          f"{256 * per_elem / 2.05e9 * 1e3:.3f} ms at the measured 2.05 GHz (ntt.roofline.effective_clock_ghz); measured 0.68-0.75 ms per middle pass "
          f"(profiles/r05_kernel_stats_ntt_2_24.txt: 743 us under rocprofv3; 2.05-2.07 ms per three-pass transform in bench.py) -> the VALU issue stream alone is "
          f"{100 * 256 * per_elem / 2.05e9 * 1e3 / 0.70:.0f} % of a 0.70-ms pass.")


if __name__ == "__main__":
    main()
