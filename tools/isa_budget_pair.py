#!/usr/bin/env python3
"""Instruction budget of one mixed addition of k_msm_accumulate_pair<BlsG2> (two lanes per Fq2 point, zl_fq2pair.h) from the ISA, beside the one-lane kernel it replaces
("the ISA budget of the pair kernel beside it").
    python tools/isa_budget_pair.py > profiles/r06_g2_pair_instruction_budget.txt
The general case of an iteration is the block with the two dual scans u2, s2 (1176 mads) + the block with PP, PPP, Q, R^2, the four-product scan of Y3, ZZ3, ZZZ3 (4116 mads);
the block with 3332 mads is dbl_affine (P == Q, never on random inputs).  VALU issue cycles as in tools/isa_budget.py (4 / 2)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openzl_amd import build as zb  # noqa: E402
FOUR = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_lshl_add_u64", "v_mad_i64_i32", "v_mov_b64_e32", "v_mov_b64"}


def blocks_of(path, prefix, must):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and must in l.split(":")[0] and ": " in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s) or re.match(r"^; (%bb\.\d+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if s and not s.startswith(";") and not s.startswith("."):
            blocks[cur].append(s)
    meta = {}
    for l in lines[end:end + 120]:
        for key in ("NumVgprs", "ScratchSize", "Occupancy", "codeLenInByte"):
            m = re.match(r"^; %s: (\d+)" % key, l.strip())
            if m and key not in meta:
                meta[key] = int(m.group(1))
    return blocks, meta


def cls(ins):
    op = ins.split()[0]
    if op == "v_mad_u64_u32": return "mad (v_mad_u64_u32)"
    if "row_ror" in ins or "quad_perm" in ins: return "DPP moves (v_mov_b32_dpp row_ror:8: the partner lane's component)"
    if op == "s_nop": return "s_nop (pads behind inline-asm statements)"
    if op.startswith("s_"): return "scalar / branch / waitcnt"
    if op.startswith("global_") or op.startswith("scratch_"): return "global / scratch memory"
    if op == "v_mul_lo_u32": return "Montgomery factor m_k (v_mul_lo_u32)"
    if op == "v_lshrrev_b64": return "column shift (v_lshrrev_b64)"
    if op.startswith("v_and_b32"): return "28-bit masks"
    if op.startswith("v_mov"): return "register moves (incl. the `old` operand copies of update_dpp)"
    if op.startswith("v_cndmask"): return "selects (v_cndmask: the squaring's operand choice, infinity, sign)"
    if op in ("v_mad_i64_i32", "v_ashrrev_i64", "v_lshl_add_u64", "v_mul_hi_u32"): return "weak reduction of the zero tests"
    if op.startswith("v_lshrrev_b32") or op.startswith("v_add3_u32") or op.startswith("v_alignbit"): return "carry passes"
    if op.startswith("v_sub") or op.startswith("v_add") or op.startswith("v_lshl"): return "limb additions / subtractions / negations"
    return "other VALU"


def cyc(ins):
    op = ins.split()[0]
    return 0 if not op.startswith("v_") else (4 if op in FOUR else 2)


def budget(blocks, names, title):
    h = collections.OrderedDict()
    for n in names:
        for ins in blocks[n]:
            e = h.setdefault(cls(ins), [0, 0]); e[0] += 1; e[1] += cyc(ins)
    ti, tc = sum(v[0] for v in h.values()), sum(v[1] for v in h.values())
    print(f"## {title}")
    print(f"{'category':86s} {'instr':>6s} {'VALU cyc':>9s} {'% cyc':>6s}")
    for c, (ni, nc) in sorted(h.items(), key=lambda kv: (-kv[1][1], -kv[1][0])):
        print(f"{c:86s} {ni:6d} {nc:9d} {100.0 * nc / tc:6.2f}")
    print(f"{'total':86s} {ti:6d} {tc:9d} {100.0:6.2f}")
    print()
    return ti, tc, h


defs = next(d for n, s, d in zb._units() if n == "zl_msm_acc_BlsG2")
out = os.path.join(tempfile.mkdtemp(prefix="isa_pair_"), "acc.s")
cmd = [zb._hipcc()] + zb.FLAGS + defs + ["--cuda-device-only", "-S", os.path.join(zb.CSRC, "zl_msm_acc.hip"), "-o", out]
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
print("# " + " ".join(cmd))
res = {}
for label, prefix, must, first_m, main_m in (("k_msm_accumulate_pair<BlsG2, false> (two lanes per chunk)", "_Z21k_msm_accumulate_pair", "Lb0EEvPKj", 1176, 4116),
                                             ("k_msm_accumulate<BlsG2> (one lane per chunk, rounds 2-5)", "_Z16k_msm_accumulateI5G2Cfg", "EEvPKj", 2352, 8232)):
    blocks, meta = blocks_of(out, prefix, must)
    mads = {n: sum(1 for i in blocks[n] if i.startswith("v_mad_u64_u32")) for n in blocks}
    hot = [n for n in blocks if mads[n] in (first_m, main_m)]
    small = [n for n in blocks if 0 < len(blocks[n]) <= 160 and mads[n] == 0]
    print(f"# {label}: {', '.join(f'{k} {v}' for k, v in meta.items())}; blocks with mads: " + ", ".join(f"{n} {mads[n]}" for n in blocks if mads[n] > 1))
    ti, tc, h = budget(blocks, hot, f"{label}: the two product blocks of the general case ({' + '.join(hot)})")
    si = sum(len(blocks[n]) for n in small)
    sc = sum(cyc(i) for n in small for i in blocks[n])
    print(f"(all small blocks of the kernel together -- loop head, bucket-boundary flushes, loads, zero tests, infinity case, latch: {si} instructions, {sc} VALU cycles; an iteration executes about half of them)")
    print()
    res[label] = (tc, h["mad (v_mad_u64_u32)"][0])
(p_c, p_m), (o_c, o_m) = res.values()
print("## Per Fq2 mixed addition")
print(f"pair kernel: 2 lanes x {p_m} mads = {2 * p_m}; one-lane kernel: {o_m} mads (the same: each Fq2 product is two dual scans either way, split over the lanes or run back to back)")
print(f"VALU issue cycles of the product blocks per lane: pair {p_c}, one lane {o_c}: the pair's two lanes issue {2 * p_c} = {2.0 * p_c / o_c:.3f} x the one-lane stream "
      f"(the routing of the partner's components: 14 subtractions + 42 DPP moves per dual scan, + the copies LLVM makes for update_dpp's `old` operand)")
print("Instruction stream of an iteration: pair ~%.0f KB, one lane ~%.0f KB (8 bytes per VOP3 instruction) against a 64-KB instruction cache" % (res[list(res)[0]][1] * 8 * 1.45 / 1024, res[list(res)[1]][1] * 8 * 1.45 / 1024))
