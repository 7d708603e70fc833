"""Shared Groth16 test/bench scaffolding (test infrastructure): Poseidon-chain R1CS (config 5), trapdoor setup via the
oracle, marshalling to the C ABI structs, oracle prove."""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle_lib as ol
from oracle_lib import po


class ZloR1cs(C.Structure):
    _fields_ = [("n_constraints", C.c_uint32), ("n_instance", C.c_uint32), ("n_witness", C.c_uint32),
                ("row_ptr", C.POINTER(C.c_uint32) * 3), ("col", C.POINTER(C.c_uint32) * 3), ("val", ol.u64p * 3)]


class ZloPk(C.Structure):
    _fields_ = [(k, ol.u64p) for k in ("a_query", "b_g1_query", "h_query", "l_query", "b_g2_query", "alpha_g1", "beta_g1", "delta_g1",
                                       "beta_g2", "delta_g2")]


class ZloProof(C.Structure):
    _fields_ = [("a", C.c_uint64 * 12), ("b", C.c_uint64 * 24), ("c", C.c_uint64 * 12), ("a_inf", C.c_uint8), ("b_inf", C.c_uint8),
                ("c_inf", C.c_uint8), ("h_out", ol.u64p)]


def r1cs_arrays(cs: po.R1CS) -> dict:
    out = {"n_constraints": cs.n_constraints, "n_instance": cs.n_instance, "n_witness": cs.n_witness}
    for key in "ABC":
        ptr, col, val = cs.csr(key)
        out[key] = (np.array(ptr, dtype=np.uint32), np.array(col, dtype=np.uint32), ol.ints_to_limbs(val, 4))
    return out


def g2_mul_gen(curve, ks) -> np.ndarray:
    k = ol.ints_to_limbs(ks, 4)
    out = np.zeros((len(ks), 4 * ol.nlq(curve)), dtype=np.uint64)
    assert ol.lib().zlo_g2_mul_gen(curve.cid, ol.p64(k), len(ks), ol.p64(out)) == 0
    return out


def setup_with_trapdoor(curve, cs: po.R1CS, td: po.Groth16Trapdoor) -> dict:
    """proving key as host arrays of canonical affine points (exponent * generator), plus the exponents"""
    ex = po.groth16_setup_exponents(curve, cs, td)
    g1 = lambda ks: ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs(ks, 4))
    return {
        "ex": ex,
        "a_query": g1(ex["a_query"]), "b_g1_query": g1(ex["b_query"]), "h_query": g1(ex["h_query"]), "l_query": g1(ex["l_query"]),
        "b_g2_query": g2_mul_gen(curve, ex["b_query"]),
        "alpha_g1": g1([td.alpha])[0], "beta_g1": g1([td.beta])[0], "delta_g1": g1([td.delta])[0],
        "beta_g2": g2_mul_gen(curve, [td.beta])[0], "delta_g2": g2_mul_gen(curve, [td.delta])[0],
    }


def oracle_prove(curve, arrays: dict, z: np.ndarray, pk: dict, r: np.ndarray, s: np.ndarray, threads: int = 8, want_h: int = 0):
    L = ol.lib()
    L.zlo_groth16_prove.argtypes = [C.c_int, C.POINTER(ZloR1cs), ol.u64p, C.POINTER(ZloPk), ol.u64p, ol.u64p, C.c_int, C.POINTER(ZloProof)]
    cs = ZloR1cs()
    cs.n_constraints, cs.n_instance, cs.n_witness = arrays["n_constraints"], arrays["n_instance"], arrays["n_witness"]
    keep = []
    for m, key in enumerate("ABC"):
        ptr, col, val = arrays[key]
        keep += [ptr, col, val]
        cs.row_ptr[m] = ptr.ctypes.data_as(C.POINTER(C.c_uint32))
        cs.col[m] = col.ctypes.data_as(C.POINTER(C.c_uint32))
        cs.val[m] = ol.p64(val)
    pkc = ZloPk()
    for k, _ in ZloPk._fields_:
        arr = np.ascontiguousarray(pk[k], dtype=np.uint64)
        keep.append(arr)
        setattr(pkc, k, ol.p64(arr))
    proof = ZloProof()
    h = None
    if want_h:
        h = np.zeros((want_h, 4), dtype=np.uint64)
        proof.h_out = ol.p64(h)
    rc = L.zlo_groth16_prove(curve.cid, C.byref(cs), ol.p64(np.ascontiguousarray(z)), C.byref(pkc), ol.p64(r), ol.p64(s), threads, C.byref(proof))
    assert rc == 0
    nq = ol.nlq(curve)
    return (np.array(proof.a[: 2 * nq], dtype=np.uint64), proof.a_inf, np.array(proof.b[: 4 * nq], dtype=np.uint64), proof.b_inf,
            np.array(proof.c[: 2 * nq], dtype=np.uint64), proof.c_inf), h


def upload_pk(backend, curve, pk: dict) -> dict:
    from openzl_amd import ZL_G1, ZL_G2

    out = {k: pk[k] for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2")}
    for k in ("a_query", "b_g1_query", "h_query", "l_query"):
        out[k] = backend.bases_upload(curve.cid, pk[k], group=ZL_G1)
    out["b_g2_query"] = backend.bases_upload(curve.cid, pk["b_g2_query"], group=ZL_G2)
    return out


def free_pk(backend, dpk: dict):
    for k in ("a_query", "b_g1_query", "h_query", "l_query", "b_g2_query"):
        backend.bases_free(dpk[k])
