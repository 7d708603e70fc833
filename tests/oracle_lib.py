"""ctypes loader for the CPU oracle (oracle/libzl_oracle.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
sys.path.insert(0, ORACLE_DIR)
import pyoracle as po  # noqa: E402

_LIB = None
_NATIVE = False
CFLAGS_PORTABLE = "-O3 -march=x86-64-v3 -fopenmp"
CFLAGS_NATIVE = "-O3 -march=native -fopenmp"
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


def build_oracle(force: bool = False) -> str:
    so = os.path.join(ORACLE_DIR, "libzl_oracle.so")
    if force:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "clean"])
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "all"])  # make is incremental: a stale oracle cannot be loaded by accident
    return so


def select_native() -> str:
    """bench.py's cpu_baseline legs: (re)build the oracle ON THIS BOX with -march=native (BASELINE.md's flags) and make lib() load that copy; returns the
    compiler flags in use.  Must be called before the first lib() of the process.  Falls back to the portable build when the box has no compiler.  The
    native object is rebuilt whenever it was made on another machine (it is never trusted across boxes)."""
    global _NATIVE
    assert _LIB is None, "select_native() after the oracle was loaded"
    so = os.path.join(ORACLE_DIR, "libzl_oracle_native.so")
    stamp = so + ".host"
    here = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0] if os.path.exists("/proc/cpuinfo") and "model name" in open("/proc/cpuinfo").read() else "?"
    try:
        if not (os.path.exists(stamp) and open(stamp).read() == here):
            if os.path.exists(so):
                os.remove(so)
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "native"])
        with open(stamp, "w") as f:
            f.write(here)
        C.CDLL(so)  # loadable here?
        _NATIVE = True
        return CFLAGS_NATIVE
    except (OSError, subprocess.CalledProcessError):
        return CFLAGS_PORTABLE + " (native build failed on this box)"


def lib():
    global _LIB
    if _LIB is None:
        so = build_oracle()
        if os.environ.get("ZL_ORACLE_LIB"):  # tests/test_sanitizers.py: the -fsanitize=address,undefined build (make -C oracle asan)
            so = os.environ["ZL_ORACLE_LIB"]
        try:
            _LIB = C.CDLL(os.path.join(ORACLE_DIR, "libzl_oracle_native.so") if _NATIVE else so)
        except OSError:
            so = build_oracle(force=True)
            _LIB = C.CDLL(so)
        L = _LIB
        L.zlo_field_op.argtypes = [C.c_int, C.c_int, u64p, u64p, u64p]
        L.zlo_field_to_mont.argtypes = [C.c_int, u64p, u64p, C.c_size_t]
        L.zlo_field_from_mont.argtypes = [C.c_int, u64p, u64p, C.c_size_t]
        for f in (L.zlo_msm_g1, L.zlo_msm_g2):
            f.argtypes = [C.c_int, u64p, C.c_int, u64p, C.c_size_t, C.c_int, C.c_int, u64p, u8p]
        L.zlo_msm_g1_ex.argtypes = [C.c_int, u64p, C.c_int, u64p, C.c_size_t, C.c_int, C.c_int, C.c_int, u64p, u8p, C.POINTER(C.c_double)]
        for f in (L.zlo_g1_mul_gen, L.zlo_g2_mul_gen):
            f.argtypes = [C.c_int, u64p, C.c_size_t, u64p]
        L.zlo_g1_mul.argtypes = [C.c_int, u64p, u64p, u64p, u8p]
        L.zlo_ntt.argtypes = [C.c_int, u64p, C.c_uint, C.c_int, C.c_int, C.c_int]
        L.zlo_ntt_ex.argtypes = [C.c_int, u64p, C.c_uint, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.zlo_poseidon3.argtypes = [u64p, u64p, C.c_int, C.c_int, u64p]
    return _LIB


def p64(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def ints_to_limbs(vals, nlimbs: int) -> np.ndarray:
    """list of python ints -> (len, nlimbs) uint64 little-endian limb array"""
    out = np.zeros((len(vals), nlimbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(nlimbs):
            out[i, j] = (v >> (64 * j)) & po.MASK64
    return out


def limbs_to_ints(a: np.ndarray):
    a = np.asarray(a, dtype=np.uint64)
    a2 = a.reshape(-1, a.shape[-1])
    return [sum(int(a2[i, j]) << (64 * j) for j in range(a2.shape[1])) for i in range(a2.shape[0])]


def nlq(curve: po.CurveParams) -> int:
    return curve.fq.limbs64


def points_to_limbs(curve: po.CurveParams, pts) -> np.ndarray:
    """affine big-int points (None = infinity -> all zero) -> (n, 2*nlq) canonical limbs"""
    n = nlq(curve)
    flat = []
    for P in pts:
        if P is None:
            flat.append(0)
        else:
            flat.append(P[0] | (P[1] << (64 * n)))
    return ints_to_limbs(flat, 2 * n)


def limbs_to_point(curve: po.CurveParams, xy: np.ndarray, inf: int):
    if inf:
        return None
    n = nlq(curve)
    v = limbs_to_ints(xy.reshape(1, -1))[0]
    return (v & ((1 << (64 * n)) - 1), v >> (64 * n))


def random_scalars(curve: po.CurveParams, n: int, seed: int) -> np.ndarray:
    """n uniform scalars < r as (n,4) uint64 (vectorised rejection sampling, numpy PCG64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((n, 4), dtype=np.uint64)
    r_l = ints_to_limbs([curve.fr.p], 4)[0]
    top_mask = np.uint64((1 << (curve.fr.bits - 192)) - 1)
    todo = np.arange(n)
    while todo.size:
        cand = rng.integers(0, 1 << 64, size=(todo.size, 4), dtype=np.uint64)
        cand[:, 3] &= top_mask
        lt = np.zeros(todo.size, dtype=bool)
        decided = np.zeros(todo.size, dtype=bool)
        for j in (3, 2, 1, 0):
            less = (cand[:, j] < r_l[j]) & ~decided
            more = (cand[:, j] > r_l[j]) & ~decided
            lt |= less
            decided |= less | more
        out[todo[lt]] = cand[lt]
        todo = todo[~lt]
    return out


def oracle_msm_g1(curve, bases: np.ndarray, scalars: np.ndarray, algo=0, threads=1, mont=False):
    n = scalars.shape[0]
    out = np.zeros(2 * nlq(curve), dtype=np.uint64)
    inf = C.c_uint8(0)
    rc = lib().zlo_msm_g1(curve.cid, p64(bases), int(mont), p64(scalars), n, algo, threads, p64(out), C.byref(inf))
    assert rc == 0
    return out, inf.value


def oracle_msm_g1_timed(curve, bases: np.ndarray, scalars: np.ndarray, algo=0, threads=1, c_override=0, mont=False):
    """-> (xy, inf, seconds of the MSM alone); algo 0 = ark window-parallel, 2 = point-chunked all-core"""
    n = scalars.shape[0]
    out = np.zeros(2 * nlq(curve), dtype=np.uint64)
    inf = C.c_uint8(0)
    sec = C.c_double(0.0)
    rc = lib().zlo_msm_g1_ex(curve.cid, p64(bases), int(mont), p64(scalars), n, algo, c_override, threads, p64(out), C.byref(inf), C.byref(sec))
    assert rc == 0
    return out, inf.value, sec.value


def oracle_ntt_timed(curve, data_mont: np.ndarray, inverse=False, coset=False, threads=1):
    """Montgomery limbs in / out -> (result, seconds of the transform alone)"""
    d = np.ascontiguousarray(data_mont.copy())
    log_n = int(d.shape[0]).bit_length() - 1
    sec = C.c_double(0.0)
    assert lib().zlo_ntt_ex(curve.cid, p64(d), log_n, int(inverse), int(coset), threads, C.byref(sec)) == 0
    return d, sec.value


def oracle_g1_mul_gen(curve, k: np.ndarray) -> np.ndarray:
    n = k.shape[0]
    out = np.zeros((n, 2 * nlq(curve)), dtype=np.uint64)
    assert lib().zlo_g1_mul_gen(curve.cid, p64(k), n, p64(out)) == 0
    return out


def oracle_ntt(curve, data: np.ndarray, inverse=False, coset=False, mont=False) -> np.ndarray:
    d = np.ascontiguousarray(data.copy())
    log_n = int(d.shape[0]).bit_length() - 1
    assert 1 << log_n == d.shape[0]
    assert lib().zlo_ntt(curve.cid, p64(d), log_n, int(inverse), int(coset), int(mont)) == 0
    return d
