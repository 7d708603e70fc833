#!/usr/bin/env python3
"""FP64-FMA Montgomery multiplier (VERDICT r3 item 1 / SURVEY.md 7.2): generator + exact model.

A field element is n doubles holding signed integers |x_i| <= 2^(w-1) ("signed-normalised" limbs), value = sum x_i 2^(w i), R = 2^(n w).
One limb product x*y (|x y| <= 2^(2w-2)) is split by the FMA's own rounding, with round-to-nearest-even (the default mode; no s_setreg):

    H' = fma(x, y, H)          H = B + (sum of earlier high parts), B = 1.5 * 2^(w + 52): every value of the chain lies in one binade with
                               ulp 2^w, so the FMA adds RN_{2^w}(x y) exactly -- the high parts of a whole column accumulate for free
    d  = H - H'                = -hi(x y), exact
    l  = fma(x, y, d)          = x y - hi(x y) in [-2^(w-1), 2^(w-1)], exact
    L  = L + l                 the low parts of a column (<= 2 n of them) + the carry stay below 2^53: exact

i.e. FOUR FP64 instructions per limb product (the high-part accumulation is the only thing that is free); the schedule is a product scan
(column by column) with m_k = (L_k q') mod+- 2^w from the low parts alone (high parts are multiples of 2^w) and signed m.
Result: r = (a b + m q) / R with |m| < R / 2, so |r| < |a||b| / R + q / 2; limbs signed-normalised again (top limb carries the sign).

The generator emits an op list; `emit_cpp` prints it as a HIP device function, `run_model` executes the SAME list with exact integer
arithmetic and IEEE-754 round-to-nearest-even emulated on Python integers (tests/test_f64_model.py: bit-exact vs a b R^-1 mod q).
"""
import os
import sys

Q_BLS381 = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R_BLS381 = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def signed_limbs(x, n, w):
    """x (any integer) as n signed limbs in [-2^(w-1), 2^(w-1)) ... top limb takes the rest"""
    out = []
    for i in range(n):
        if i == n - 1:
            out.append(x)
            break
        d = x & ((1 << w) - 1)
        if d >= 1 << (w - 1):
            d -= 1 << w
        out.append(d)
        x = (x - d) >> w
    return out


def gen_ops(n, w, q, sqr=False):
    """op list of one Montgomery product r = a b / 2^(n w) (mod q).  ops: (kind, dst, srcs...) with kinds fma / add / sub;
    sources are variable names or ('c', python_int_or_float) constants."""
    S = w
    B = 3 * (1 << (S + 51))            # 1.5 * 2^(S + 52)
    ql = signed_limbs(q, n, w)
    assert all(abs(v) <= 1 << (w - 1) for v in ql), "modulus limbs must be signed-normalised"
    qinv = (-pow(q, -1, 1 << w)) % (1 << w)
    if qinv >= 1 << (w - 1):
        qinv -= 1 << w
    ops, uid = [], [0]

    def C(v):
        return ("c", v)

    def new(prefix):
        uid[0] += 1
        return f"{prefix}{uid[0]}"

    def op(kind, *srcs, prefix="t"):
        d = new(prefix)
        ops.append((kind, d) + srcs)
        return d

    scale = ("c2", -w)                  # 2^-w
    Bs = B >> w                         # B 2^-w
    carry = None
    m = [None] * n
    out = [None] * n
    if sqr:
        a2 = {j: op("add", f"a{j}", f"a{j}", prefix="e") for j in range(1, n)}   # 2 a_j, exact
    for k in range(2 * n):
        H, L = C(B), carry
        if sqr:
            prods = [(f"a{i}", a2[k - i]) for i in range(max(0, k - n + 1), min(k, n - 1) + 1) if i < k - i]
            if k % 2 == 0 and k // 2 < n:
                prods.append((f"a{k // 2}", f"a{k // 2}"))
        else:
            prods = [(f"a{i}", f"b{k - i}") for i in range(max(0, k - n + 1), min(k, n - 1) + 1)]
        prods += [(m[i], C(ql[k - i])) for i in (range(0, k) if k < n else range(k - n + 1, n))]

        def accumulate(x, y):
            nonlocal H, L
            Hn = op("fma", x, y, H, prefix="h")
            d = op("sub", H, Hn, prefix="d")
            if L is None:
                L = op("fma", x, y, d, prefix="l")
            else:
                l = op("fma", x, y, d, prefix="l")
                L = op("add", L, l, prefix="L")
            H = Hn

        for x, y in prods:
            accumulate(x, y)
        M = C(B)                        # adding then subtracting 1.5 * 2^(w + 52) rounds to a multiple of 2^w
        if k < n:
            t = op("add", L, M, prefix="n")   # the one addition that is MEANT to round
            t = op("sub", t, M)
            l0 = op("sub", L, t)
            ph = op("fma", l0, C(qinv), C(B), prefix="p")
            d = op("sub", C(B), ph)
            m[k] = op("fma", l0, C(qinv), d, prefix="m")
            accumulate(m[k], C(ql[0]))
            c = op("fma", H, scale, C(-Bs))
            carry = op("fma", L, scale, c, prefix="c")
        elif k < 2 * n - 1:
            t = op("add", L, M, prefix="n")
            t = op("sub", t, M)
            out[k - n] = op("sub", L, t, prefix="r")
            c = op("fma", H, scale, C(-Bs))
            carry = op("fma", t, scale, c, prefix="c")
        else:
            hs = op("sub", H, C(B))
            out[k - n] = op("add", L, hs, prefix="r")
    return ops, out


# ---------------------------------------------------------------------------------------------- exact model
def rne53(x):
    """IEEE-754 binary64 round-to-nearest-even of the integer x (no overflow / subnormals at these magnitudes)"""
    if x == 0:
        return 0
    s, a = (1, x) if x > 0 else (-1, -x)
    e = a.bit_length() - 53
    if e <= 0:
        return x
    lo = a & ((1 << e) - 1)
    a >>= e
    half = 1 << (e - 1)
    if lo > half or (lo == half and (a & 1)):
        a += 1
    return s * (a << e)


class Inexact(Exception):
    pass


def run_model(ops, out, env, w, strict=True):
    """env: variable -> integer value (scaled values: the 2^-w scalings are tracked as exact rationals via a fixed shift)."""
    SH = 2 * w  # every value is kept multiplied by 2^SH so that the 2^-w scalings stay integers
    val = {k: v << SH for k, v in env.items()}

    def get(s):
        if isinstance(s, tuple):
            if s[0] == "c":
                return s[1] << SH
            if s[0] == "c2":
                return 1 << (SH + s[1])
        return val[s]

    for o in ops:
        kind, d = o[0], o[1]
        if kind == "fma":
            x, y, z = get(o[2]), get(o[3]), get(o[4])
            exact = x * y + (z << SH)            # times 2^(2 SH)
            r = rne53(exact)
            # only the chain FMA (dst h*) and the quotient's high part (the op whose addend is the bias constant) may round
            if strict and r != exact and not (d.startswith("h") or d.startswith("p")):
                raise Inexact(f"fma {d}")
            assert r % (1 << SH) == 0 or not strict, f"sub-unit bits survive in {d}"
            val[d] = r >> SH
        else:
            x, y = get(o[2]), get(o[3])
            exact = x + y if kind == "add" else x - y
            r = rne53(exact)
            if strict and r != exact and not d.startswith("n"):
                raise Inexact(f"{kind} {d}")
            val[d] = r
    res = []
    for v in out:
        assert val[v] % (1 << SH) == 0
        res.append(val[v] >> SH)
    return res


# ---------------------------------------------------------------------------------------------- C++ emission
def cpp_const(s):
    if s[0] == "c":
        v = s[1]
        return (float(v).hex() if abs(v) < 1 << 1023 else None)
    if s[0] == "c2":
        return f"0x1p{s[1]}"


def emit_cpp(name, n, w, q, sqr=False):
    ops, out = gen_ops(n, w, q, sqr)

    def src(s):
        if isinstance(s, tuple):
            return cpp_const(s)
        if s[0] in "ab" and s[1:].isdigit():
            return f"{s[0]}[{s[1:]}]"
        return s

    nf = sum(1 for o in ops if o[0] == "fma")
    na = len(ops) - nf
    lines = [f"// {name}: {n} x {w}-bit signed limbs in doubles, R = 2^{n * w}; {nf} v_fma_f64 + {na} v_add_f64 = {len(ops)} FP64 instructions per product",
             f"__device__ __forceinline__ void {name}(double* __restrict__ r, const double* __restrict__ a" + ("" if sqr else ", const double* __restrict__ b") + ") {"]
    for o in ops:
        if o[0] == "fma":
            lines.append(f"    const double {o[1]} = __builtin_fma({src(o[2])}, {src(o[3])}, {src(o[4])});")
        else:
            lines.append(f"    const double {o[1]} = {src(o[2])} {'+' if o[0] == 'add' else '-'} {src(o[3])};")
    for i, v in enumerate(out):
        lines.append(f"    r[{i}] = {v};")
    lines.append("}")
    return "\n".join(lines), len(ops), nf


def main():
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mul_f64_gen.h")
    txt = ["// GENERATED by tools/gen_mul_f64.py -- do not edit.  FP64-FMA Montgomery products (micro-benchmark only; not in the product).",
           "#pragma once"]
    for name, n, w, q, sq in (("mul_f64_fq", 8, 49, Q_BLS381, False), ("sqr_f64_fq", 8, 49, Q_BLS381, True), ("mul_f64_fr", 6, 44, R_BLS381, False)):
        code, total, nf = emit_cpp(name, n, w, q, sq)
        txt.append(code)
        txt.append(f"#define {name.upper()}_OPS {total}")
        txt.append(f"#define {name.upper()}_FMAS {nf}")
    open(dst, "w").write("\n".join(txt) + "\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
