"""ctypes binding of include/zl_backend.h.  No compute happens in Python and there is no CPU fallback."""
from __future__ import annotations

import collections
import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libzl_backend.so")

ZL_BLS12_381, ZL_BN254 = 1, 2
ZL_G1, ZL_G2 = 1, 2
ZL_MONT, ZL_COSET, ZL_INVERSE, ZL_CHECK, ZL_MONT_IN, ZL_MONT_OUT = 1, 2, 4, 8, 16, 32
ZL_PARTIAL_WORDS = 64
CURVES = {"bls12_381": ZL_BLS12_381, "bn254": ZL_BN254}
FQ_LIMBS = {ZL_BLS12_381: 6, ZL_BN254: 4}

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)

# every symbol include/zl_backend.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "zl_ctx_create", "zl_ctx_destroy", "zl_ctx_fork", "zl_ctx_set_stream", "zl_ctx_sync", "zl_ctx_set_msm_window", "zl_ctx_last_hip_error",
    "zl_strerror", "zl_describe", "zl_bases_upload", "zl_bases_generate", "zl_bases_download", "zl_bases_precompute", "zl_bases_free", "zl_msm",
    "zl_msm_dev", "zl_msm_partial_dev", "zl_msm_batch_partial_dev", "zl_partials_sum", "zl_partial_from_affine", "zl_ntt", "zl_ntt_dev", "zl_ntt_batch_dev", "zl_ntt_cross_dev", "zl_ctx_enable_timing", "zl_last_timing", "zl_groth16_prove", "zl_groth16_last_h", "zl_r1cs_upload", "zl_r1cs_free", "zl_groth16_prove_resident", "zl_groth16_prove_sharded", "zl_circuit_poseidon_chain", "zl_circuit_poseidon_chain_witness", "zl_circuit_free", "zl_circuit_export",
    "zl_circuit_is_satisfied", "zl_poseidon_permute", "zl_groth16_compile", "zl_groth16_keys_free", "zl_groth16_keys_pk",
    "zl_groth16_keys_trapdoor", "zl_groth16_prove_circuit", "zl_groth16_prove_circuits", "zl_ctx_drop_lanes", "zl_groth16_verify", "zl_pairing",
    "zl_ctx_create_multi", "zl_mctx_destroy", "zl_mctx_size", "zl_mctx_ctx", "zl_mctx_uses_rccl", "zl_mctx_last_rccl_error", "zl_msm_sharded", "zl_ntt_sharded",
    "zl_point_bytes", "zl_point_to_bytes", "zl_point_from_bytes", "zl_groth16_proof_bytes", "zl_groth16_proof_to_bytes", "zl_groth16_proof_from_bytes",
    "zl_point_bytes_uncompressed", "zl_point_to_bytes_uncompressed", "zl_point_from_bytes_uncompressed", "zl_groth16_keys_to_bytes", "zl_groth16_keys_from_bytes", "zl_groth16_keys_parse",
    "zl_groth16_vk_to_bytes",
]


class BackendError(RuntimeError):
    def __init__(self, code: int, what: str, msg: str = ""):
        super().__init__(f"{what} failed: {code} ({msg})")
        self.code = code


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("dominant_ms", C.c_float), ("launches", C.c_uint32), ("window_bits", C.c_uint32),
                ("entries", C.c_uint64)]


class R1csC(C.Structure):
    _fields_ = [("n_constraints", C.c_uint32), ("n_instance", C.c_uint32), ("n_witness", C.c_uint32),
                ("row_ptr", C.POINTER(C.c_uint32) * 3), ("col", C.POINTER(C.c_uint32) * 3), ("val", u64p * 3)]


class G16PkC(C.Structure):
    _fields_ = [("curve", C.c_int), ("a_query", C.c_uint64), ("b_g1_query", C.c_uint64), ("h_query", C.c_uint64), ("l_query", C.c_uint64),
                ("b_g2_query", C.c_uint64), ("alpha_g1", u64p), ("beta_g1", u64p), ("delta_g1", u64p), ("beta_g2", u64p), ("delta_g2", u64p)]


class G16ShardC(C.Structure):
    _fields_ = [("a_query", C.c_uint64), ("b_g1_query", C.c_uint64), ("h_query", C.c_uint64), ("l_query", C.c_uint64), ("b_g2_query", C.c_uint64),
                ("var_first", C.c_size_t), ("var_count", C.c_size_t), ("wit_first", C.c_size_t), ("wit_count", C.c_size_t), ("h_first", C.c_size_t),
                ("h_count", C.c_size_t)]


class G16ProofC(C.Structure):
    _fields_ = [("a", C.c_uint64 * 12), ("b", C.c_uint64 * 24), ("c", C.c_uint64 * 12), ("a_inf", C.c_uint8), ("b_inf", C.c_uint8),
                ("c_inf", C.c_uint8)]


_lib = None


def load_library(path: Optional[str] = None):
    """Load libzl_backend.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("ZL_BACKEND_LIB") or LIB_PATH  # ZL_BACKEND_LIB: developer A/B of two builds in one gpurun call
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(p)
    vp = C.c_void_p
    L.zl_ctx_create.argtypes = [C.POINTER(vp), C.c_int]
    L.zl_ctx_destroy.argtypes = [vp]
    L.zl_ctx_fork.argtypes = [vp, C.POINTER(vp)]
    L.zl_ctx_destroy.restype = None
    L.zl_ctx_set_stream.argtypes = [vp, vp]
    L.zl_ctx_sync.argtypes = [vp]
    L.zl_ctx_set_msm_window.argtypes = [vp, C.c_int]
    L.zl_ctx_last_hip_error.argtypes = [vp]
    L.zl_strerror.argtypes = [C.c_int]
    L.zl_strerror.restype = C.c_char_p
    L.zl_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.zl_bases_upload.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_long, C.c_uint, u64p]
    L.zl_bases_generate.argtypes = [vp, C.c_int, C.c_int, u64p, C.c_size_t, u64p]
    L.zl_bases_download.argtypes = [vp, C.c_uint64, C.c_size_t, C.c_size_t, u64p]
    L.zl_bases_free.argtypes = [vp, C.c_uint64]
    L.zl_bases_precompute.argtypes = [vp, C.c_uint64, C.c_int]
    L.zl_msm.argtypes = [vp, C.c_uint64, C.c_size_t, u64p, C.c_size_t, u64p, u8p]
    L.zl_msm_dev.argtypes = [vp, C.c_uint64, C.c_size_t, vp, C.c_size_t, u64p, u8p]
    L.zl_msm_partial_dev.argtypes = [vp, C.c_uint64, C.c_size_t, vp, C.c_size_t, u64p]
    L.zl_partials_sum.argtypes = [C.c_int, C.c_int, u64p, C.c_size_t, u64p, u8p]
    L.zl_partial_from_affine.argtypes = [C.c_int, C.c_int, u64p, u64p]
    L.zl_msm_batch_partial_dev.argtypes = [vp, C.c_uint64, C.c_size_t, C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, u64p]
    L.zl_ntt.argtypes = [vp, C.c_int, u64p, C.c_uint, C.c_uint]
    L.zl_ntt_dev.argtypes = [vp, C.c_int, vp, C.c_uint, C.c_uint]
    L.zl_ntt_cross_dev.argtypes = [vp, C.c_int, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    L.zl_ntt_batch_dev.argtypes = [vp, C.c_int, vp, C.c_uint, C.c_uint, C.c_uint, C.c_size_t]
    L.zl_ctx_enable_timing.argtypes = [vp, C.c_int]
    L.zl_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.zl_groth16_prove.argtypes = [vp, C.POINTER(G16PkC), C.POINTER(R1csC), u64p, u64p, u64p, C.POINTER(G16ProofC)]
    L.zl_groth16_prove_sharded.argtypes = [vp, C.POINTER(G16PkC), C.POINTER(G16ShardC), C.c_uint64, u64p, C.c_uint, u64p, u64p, C.POINTER(G16ProofC)]
    L.zl_groth16_last_h.argtypes = [vp, u64p, C.c_size_t]
    L.zl_circuit_poseidon_chain.argtypes = [C.c_int, C.c_uint32, u64p, u64p, C.POINTER(vp)]
    L.zl_circuit_free.argtypes = [vp]
    L.zl_circuit_free.restype = None
    L.zl_circuit_export.argtypes = [vp, C.POINTER(R1csC), C.POINTER(u64p)]
    L.zl_circuit_poseidon_chain_witness.argtypes = L.zl_circuit_poseidon_chain.argtypes
    L.zl_circuit_is_satisfied.argtypes = [vp]
    L.zl_poseidon_permute.argtypes = [C.c_int, u64p]
    L.zl_groth16_compile.argtypes = [vp, vp, C.c_uint64, C.POINTER(vp)]
    L.zl_groth16_keys_free.argtypes = [vp]
    L.zl_groth16_keys_free.restype = None
    L.zl_groth16_keys_pk.argtypes = [vp, C.POINTER(G16PkC)]
    L.zl_groth16_keys_trapdoor.argtypes = [vp, u64p]
    L.zl_groth16_prove_circuit.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(G16ProofC), u64p, u64p]
    L.zl_groth16_prove_circuits.argtypes = [vp, vp, C.POINTER(vp), u64p, C.c_size_t, C.POINTER(G16ProofC)]
    L.zl_ctx_drop_lanes.argtypes = [vp]
    L.zl_groth16_verify.argtypes = [vp, u64p, C.c_size_t, C.POINTER(G16ProofC), C.POINTER(C.c_int)]
    L.zl_pairing.argtypes = [C.c_int, u64p, u64p, u64p]
    L.zl_point_bytes.argtypes = [C.c_int, C.c_int]
    L.zl_point_bytes.restype = C.c_size_t
    L.zl_point_to_bytes.argtypes = [C.c_int, C.c_int, u64p, C.c_uint8, u8p]
    L.zl_point_from_bytes.argtypes = [C.c_int, C.c_int, u8p, u64p, u8p]
    L.zl_groth16_proof_bytes.argtypes = [C.c_int]
    L.zl_groth16_proof_bytes.restype = C.c_size_t
    L.zl_groth16_proof_to_bytes.argtypes = [C.c_int, C.POINTER(G16ProofC), u8p]
    L.zl_groth16_proof_from_bytes.argtypes = [C.c_int, u8p, C.c_size_t, C.POINTER(G16ProofC)]
    L.zl_point_bytes_uncompressed.argtypes = [C.c_int, C.c_int]
    L.zl_point_bytes_uncompressed.restype = C.c_size_t
    L.zl_point_to_bytes_uncompressed.argtypes = [C.c_int, C.c_int, u64p, C.c_uint8, u8p]
    L.zl_point_from_bytes_uncompressed.argtypes = [C.c_int, C.c_int, u8p, C.c_int, u64p, u8p]
    L.zl_groth16_keys_to_bytes.argtypes = [vp, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zl_groth16_keys_from_bytes.argtypes = [vp, C.c_int, u8p, C.c_size_t, C.c_uint, C.POINTER(vp)]
    L.zl_groth16_keys_parse.argtypes = [C.c_int, u8p, C.c_size_t, C.c_uint]
    L.zl_groth16_vk_to_bytes.argtypes = [vp, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    # multi-GPU in one process
    L.zl_ctx_create_multi.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    L.zl_mctx_destroy.argtypes = [vp]
    L.zl_mctx_destroy.restype = None
    L.zl_mctx_size.argtypes = [vp]
    L.zl_mctx_ctx.argtypes = [vp, C.c_int]
    L.zl_mctx_ctx.restype = vp
    L.zl_mctx_uses_rccl.argtypes = [vp]
    L.zl_mctx_last_rccl_error.argtypes = [vp]
    L.zl_msm_sharded.argtypes = [vp, u64p, C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t), u64p, u8p]
    L.zl_ntt_sharded.argtypes = [vp, C.c_int, C.POINTER(vp), C.c_uint, C.c_uint]
    # test-only hooks (include/zl_backend_test.h)
    u32p = C.POINTER(C.c_uint32)
    L.zl_test_poseidon_permute_dev.argtypes = [vp, C.c_int, u64p]
    L.zl_test_fp28_op.argtypes = [vp, C.c_int, u32p, C.c_size_t, u32p]
    L.zl_test_fp28_bn_op.argtypes = [vp, C.c_int, u32p, C.c_size_t, u32p]
    L.zl_test_pairing_product.argtypes = [C.c_int, C.c_size_t, u64p, u64p, u64p]
    L.zl_test_point_op.argtypes = [vp, C.c_int, C.c_int, C.c_int, u32p, C.c_size_t, u32p]
    L.zl_test_circuit_tweak.argtypes = [vp]
    L.zl_test_fq_mul_rate.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.zl_test_fr28_op.argtypes = [vp, C.c_int, C.c_int, C.c_int, u32p, C.c_size_t, u32p]
    L.zl_test_fr29_op.argtypes = [vp, C.c_int, C.c_int, C.c_int, u32p, C.c_size_t, u32p]
    L.zl_test_poseidon_permute_dev28r.argtypes = [vp, C.c_int, u64p]
    L.zl_test_fq_mul_clock.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.zl_test_acc_clock.argtypes = [vp, C.c_int]
    L.zl_test_acc_clock_read.argtypes = [vp, C.POINTER(C.c_double)]
    L.zl_test_clock_probe_launch.argtypes = [vp, C.c_uint]
    L.zl_test_clock_probe_read.argtypes = [vp, C.POINTER(C.c_double)]
    if path is None:
        _lib = L
    return L


def _p64(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "need contiguous uint64"
    return a.ctypes.data_as(u64p)


class Backend:
    """One zl_ctx (one GPU, one stream).  Mirrors the two upstream entry points the arkworks plugin surfaces:
    VariableBaseMSM::multi_scalar_mul -> msm(); Radix2EvaluationDomain::{fft,ifft,coset_*} -> ntt()."""

    def __init__(self, device: int = 0, _borrowed_ctx=None):
        self.L = load_library()
        self._bases = {}
        self._owned = _borrowed_ctx is None
        if _borrowed_ctx is not None:  # a rank of a MultiBackend: the zl_mctx owns the ctx
            self._ctx = C.c_void_p(_borrowed_ctx)
            return
        self._ctx = C.c_void_p()
        self._check(self.L.zl_ctx_create(C.byref(self._ctx), device), "zl_ctx_create")

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise BackendError(rc, what, self.L.zl_strerror(rc).decode())

    def fork(self) -> "Backend":
        """zl_ctx_fork: a second prover lane on this device that reads this backend's device-resident keys (one lane per host thread; close it before this one)"""
        child = Backend.__new__(Backend)
        child.L, child._bases, child._owned, child._parent = self.L, collections.ChainMap({}, self._bases), True, self
        child._ctx = C.c_void_p()
        self._check(self.L.zl_ctx_fork(self._ctx, C.byref(child._ctx)), "zl_ctx_fork")
        self.__dict__.setdefault("_forks", []).append(child)
        return child

    def close(self):
        for f in self.__dict__.pop("_forks", []):  # the lanes go first: they read this ctx's objects
            f.close()
        if self._ctx:
            if self._owned:
                self.L.zl_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self) -> str:
        buf = C.create_string_buffer(512)
        self.L.zl_describe(self._ctx, buf, 512)
        return buf.value.decode()

    def set_stream(self, stream_handle: int):
        self._check(self.L.zl_ctx_set_stream(self._ctx, C.c_void_p(stream_handle)), "zl_ctx_set_stream")

    def set_msm_window(self, c: int):
        self._check(self.L.zl_ctx_set_msm_window(self._ctx, c), "zl_ctx_set_msm_window")

    def enable_timing(self, on: bool = True):
        self._check(self.L.zl_ctx_enable_timing(self._ctx, int(on)), "zl_ctx_enable_timing")

    def last_timing(self) -> Timing:
        t = Timing()
        self._check(self.L.zl_last_timing(self._ctx, C.byref(t)), "zl_last_timing")
        return t

    def sync(self):
        self._check(self.L.zl_ctx_sync(self._ctx), "zl_ctx_sync")

    # ---- bases ------------------------------------------------------------------------------------------------
    def bases_upload(self, curve: int, xy: np.ndarray, group: int = ZL_G1, flags: int = 0, stride: int = 0, inf_offset: int = -1,
                     n: Optional[int] = None) -> int:
        h = C.c_uint64()
        if n is None:
            n = xy.shape[0]
        ptr = xy.ctypes.data_as(C.c_void_p) if xy.size else None
        self._check(self.L.zl_bases_upload(self._ctx, curve, group, ptr, n, stride, inf_offset, flags, C.byref(h)), "zl_bases_upload")
        self._bases[h.value] = (curve, group, n)
        return h.value

    def bases_generate(self, curve: int, k: np.ndarray, group: int = ZL_G1) -> int:
        h = C.c_uint64()
        self._check(self.L.zl_bases_generate(self._ctx, curve, group, _p64(k), k.shape[0], C.byref(h)), "zl_bases_generate")
        self._bases[h.value] = (curve, group, k.shape[0])
        return h.value

    def bases_download(self, handle: int, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        curve, group, n = self._bases[handle]
        if count is None:
            count = n - first
        out = np.zeros((count, 2 * group * FQ_LIMBS[curve]), dtype=np.uint64)
        self._check(self.L.zl_bases_download(self._ctx, handle, first, count, _p64(out)), "zl_bases_download")
        return out

    def bases_precompute(self, handle: int, c: int = 0):
        """build the table of 2^(c w) P_i (W x memory) so that all windows share one bucket set"""
        self._check(self.L.zl_bases_precompute(self._ctx, handle, c), "zl_bases_precompute")

    def bases_free(self, handle: int):
        self._check(self.L.zl_bases_free(self._ctx, handle), "zl_bases_free")
        self._bases.pop(handle, None)

    # ---- MSM --------------------------------------------------------------------------------------------------
    def _out(self, handle: int):
        curve, group, _ = self._bases[handle]
        return np.zeros(2 * group * FQ_LIMBS[curve], dtype=np.uint64)

    def msm(self, handle: int, scalars: np.ndarray, first: int = 0) -> Tuple[np.ndarray, int]:
        """scalars: (n,4) uint64 canonical, host memory."""
        out, inf = self._out(handle), C.c_uint8(0)
        n = scalars.shape[0]
        self._check(self.L.zl_msm(self._ctx, handle, first, _p64(scalars) if n else None, n, _p64(out), C.byref(inf)), "zl_msm")
        return out, inf.value

    def msm_dev(self, handle: int, d_scalars: int, n: int, first: int = 0) -> Tuple[np.ndarray, int]:
        """d_scalars: device pointer (e.g. torch tensor .data_ptr()) to n x 4 u64 canonical scalars in HBM."""
        out, inf = self._out(handle), C.c_uint8(0)
        self._check(self.L.zl_msm_dev(self._ctx, handle, first, C.c_void_p(d_scalars), n, _p64(out), C.byref(inf)), "zl_msm_dev")
        return out, inf.value

    def msm_partial_dev(self, handle: int, d_scalars: int, n: int, first: int = 0) -> np.ndarray:
        out = np.zeros(ZL_PARTIAL_WORDS, dtype=np.uint64)
        self._check(self.L.zl_msm_partial_dev(self._ctx, handle, first, C.c_void_p(d_scalars), n, _p64(out)), "zl_msm_partial_dev")
        return out

    def msm_batch_partial_dev(self, handle: int, d_scalars, n: int, first: int = 0) -> np.ndarray:
        """d_scalars: list of device pointers (one scalar vector per MSM) -> (count, ZL_PARTIAL_WORDS) partial sums; pipelined."""
        count = len(d_scalars)
        ptrs = (C.c_void_p * max(1, count))(*[int(p) for p in d_scalars])
        out = np.zeros((count, ZL_PARTIAL_WORDS), dtype=np.uint64)
        self._check(self.L.zl_msm_batch_partial_dev(self._ctx, handle, first, ptrs, n, count, _p64(out)), "zl_msm_batch_partial_dev")
        return out

    def partials_sum(self, curve: int, partials: np.ndarray, group: int = ZL_G1) -> Tuple[np.ndarray, int]:
        partials = np.ascontiguousarray(partials.reshape(-1, ZL_PARTIAL_WORDS))
        out, inf = np.zeros(2 * group * FQ_LIMBS[curve], dtype=np.uint64), C.c_uint8(0)
        self._check(self.L.zl_partials_sum(curve, group, _p64(partials), partials.shape[0], _p64(out), C.byref(inf)), "zl_partials_sum")
        return out, inf.value

    # ---- NTT --------------------------------------------------------------------------------------------------
    def ntt(self, curve: int, data: np.ndarray, inverse: bool = False, coset: bool = False, mont: bool = False) -> np.ndarray:
        """data: (2^k, 4) uint64, returns the transformed copy (natural order)."""
        d = np.ascontiguousarray(data.copy())
        n = d.shape[0]
        log_n = n.bit_length() - 1
        if n == 0 or (1 << log_n) != n:
            raise BackendError(-1, "zl_ntt", "length must be a power of two")
        flags = (ZL_INVERSE if inverse else 0) | (ZL_COSET if coset else 0) | (ZL_MONT if mont else 0)
        self._check(self.L.zl_ntt(self._ctx, curve, _p64(d), log_n, flags), "zl_ntt")
        return d

    def ntt_dev(self, curve: int, d_data: int, log_n: int, inverse: bool = False, coset: bool = False, mont: bool = True):
        flags = (ZL_INVERSE if inverse else 0) | (ZL_COSET if coset else 0) | (ZL_MONT if mont else 0)
        self._check(self.L.zl_ntt_dev(self._ctx, curve, C.c_void_p(d_data), log_n, flags), "zl_ntt_dev")

    def ntt_dev_flags(self, curve: int, d_data: int, log_n: int, flags: int):
        """zl_ntt_dev with raw flags (ZL_MONT_IN / ZL_MONT_OUT legs of the distributed transform)."""
        self._check(self.L.zl_ntt_dev(self._ctx, curve, C.c_void_p(d_data), log_n, flags), "zl_ntt_dev")

    def ntt_batch_dev(self, curve: int, d_data: int, log_n: int, flags: int, count: int, stride_elems: int):
        """`count` equal transforms, one launch per pass (include/zl_backend_ext.h: zl_ntt_batch_dev)."""
        self._check(self.L.zl_ntt_batch_dev(self._ctx, curve, C.c_void_p(d_data), log_n, flags, count, stride_elems), "zl_ntt_batch_dev")

    def ntt_cross_dev(self, curve: int, d_data: int, log_n: int, log_g: int, rank: int, flags: int):
        """Cross-rank step of the distributed transform (include/zl_backend.h: zl_ntt_cross_dev)."""
        self._check(self.L.zl_ntt_cross_dev(self._ctx, curve, C.c_void_p(d_data), log_n, log_g, rank, flags), "zl_ntt_cross_dev")

    # ---- Groth16 ----------------------------------------------------------------------------------------------
    def groth16_prove(self, curve: int, pk: dict, r1cs: dict, assignment: np.ndarray, r: np.ndarray, s: np.ndarray):
        """pk: {'a_query','b_g1_query','h_query','l_query','b_g2_query': handles, 'alpha_g1','beta_g1','delta_g1','beta_g2',
        'delta_g2': uint64 arrays}; r1cs: {'n_constraints','n_instance','n_witness', 'A'/'B'/'C': (ptr u32, col u32, val (nnz,4) u64)}.
        Returns (a, a_inf, b, b_inf, c, c_inf) as canonical affine limb arrays."""
        cs = R1csC()
        cs.n_constraints, cs.n_instance, cs.n_witness = r1cs["n_constraints"], r1cs["n_instance"], r1cs["n_witness"]
        keep = []
        for m, key in enumerate("ABC"):
            ptr, col, val = r1cs[key]
            ptr = np.ascontiguousarray(ptr, dtype=np.uint32)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            val = np.ascontiguousarray(val, dtype=np.uint64)
            keep += [ptr, col, val]
            cs.row_ptr[m] = ptr.ctypes.data_as(C.POINTER(C.c_uint32))
            cs.col[m] = col.ctypes.data_as(C.POINTER(C.c_uint32))
            cs.val[m] = val.ctypes.data_as(u64p)
        pkc = G16PkC()
        pkc.curve = curve
        for k in ("a_query", "b_g1_query", "h_query", "l_query", "b_g2_query"):
            setattr(pkc, k, pk[k])
        for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2"):
            arr = np.ascontiguousarray(pk[k], dtype=np.uint64)
            keep.append(arr)
            setattr(pkc, k, arr.ctypes.data_as(u64p))
        proof = G16ProofC()
        z = np.ascontiguousarray(assignment, dtype=np.uint64)
        self._check(self.L.zl_groth16_prove(self._ctx, C.byref(pkc), C.byref(cs), _p64(z), _p64(np.ascontiguousarray(r)),
                                            _p64(np.ascontiguousarray(s)), C.byref(proof)), "zl_groth16_prove")
        nq = FQ_LIMBS[curve]
        a = np.array(proof.a[: 2 * nq], dtype=np.uint64)
        b = np.array(proof.b[: 4 * nq], dtype=np.uint64)
        c = np.array(proof.c[: 2 * nq], dtype=np.uint64)
        return a, proof.a_inf, b, proof.b_inf, c, proof.c_inf

    def groth16_last_h(self, n: int) -> np.ndarray:
        out = np.zeros((n, 4), dtype=np.uint64)
        self._check(self.L.zl_groth16_last_h(self._ctx, _p64(out), n), "zl_groth16_last_h")
        return out


class ShardedGroth16Keys:
    """a proving key spread over the ranks of a MultiBackend (zl_g16_shard per rank) + the constraint matrices on rank 0"""

    def __init__(self, mb: "MultiBackend", curve: int, pk: dict, r1cs: dict, cuts=None):
        self.mb, self.curve, self.L = mb, curve, mb.L
        G = mb.size
        ni, nw = r1cs["n_instance"], r1cs["n_witness"]
        nv = ni + nw
        n_h = pk["h_query"].shape[0]

        def bounds(total):
            if cuts is not None:
                edges = [0] + [int(round(total * f)) for f in cuts] + [total]
            else:
                edges = [total * g // G for g in range(G + 1)]
            return [(edges[g], max(0, edges[g + 1] - edges[g])) for g in range(G)]

        vb, wb, hb = bounds(nv), bounds(nw), bounds(n_h)
        self.shards = (G16ShardC * G)()
        self._made, self._keep, self._r1cs = [], [], 0
        try:
            for g in range(G):
                be = mb.ranks[g]
                sh = self.shards[g]
                (sh.var_first, sh.var_count), (sh.wit_first, sh.wit_count), (sh.h_first, sh.h_count) = vb[g], wb[g], hb[g]
                for name, (f, c), grp in (("a_query", vb[g], ZL_G1), ("b_g1_query", vb[g], ZL_G1), ("h_query", hb[g], ZL_G1), ("l_query", wb[g], ZL_G1),
                                          ("b_g2_query", vb[g], ZL_G2)):
                    if c == 0:
                        continue
                    hnd = be.bases_upload(curve, np.ascontiguousarray(pk[name][f:f + c]), group=grp)
                    self._made.append((be, hnd))
                    setattr(sh, name, hnd)
            cs = R1csC()
            cs.n_constraints, cs.n_instance, cs.n_witness = r1cs["n_constraints"], ni, nw
            keep = []
            for m, key in enumerate("ABC"):
                ptr, col, val = (np.ascontiguousarray(x, dtype=t) for x, t in zip(r1cs[key], (np.uint32, np.uint32, np.uint64)))
                keep += [ptr, col, val]
                cs.row_ptr[m] = ptr.ctypes.data_as(C.POINTER(C.c_uint32))
                cs.col[m] = col.ctypes.data_as(C.POINTER(C.c_uint32))
                cs.val[m] = val.ctypes.data_as(u64p)
            hr = C.c_uint64()
            be0 = mb.ranks[0]
            be0._check(self.L.zl_r1cs_upload(be0._ctx, curve, C.byref(cs), C.byref(hr)), "zl_r1cs_upload")
            self._r1cs = hr.value
            self.pkc = G16PkC()
            self.pkc.curve = curve
            for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2"):
                arr = np.ascontiguousarray(pk[k], dtype=np.uint64)
                self._keep.append(arr)
                setattr(self.pkc, k, arr.ctypes.data_as(u64p))
        except Exception:
            self.close()
            raise

    def prove(self, assignment: np.ndarray, r: np.ndarray, s: np.ndarray, mont: bool = False):
        proof = G16ProofC()
        z = np.ascontiguousarray(assignment, dtype=np.uint64)
        rc = self.L.zl_groth16_prove_sharded(self.mb._m, C.byref(self.pkc), self.shards, self._r1cs, _p64(z), ZL_MONT if mont else 0, _p64(np.ascontiguousarray(r)),
                                             _p64(np.ascontiguousarray(s)), C.byref(proof))
        if rc:
            raise BackendError(rc, "zl_groth16_prove_sharded", self.L.zl_strerror(rc).decode())
        nq = FQ_LIMBS[self.curve]
        return (np.array(proof.a[: 2 * nq], dtype=np.uint64), proof.a_inf, np.array(proof.b[: 4 * nq], dtype=np.uint64), proof.b_inf,
                np.array(proof.c[: 2 * nq], dtype=np.uint64), proof.c_inf)

    def close(self):
        for be, hnd in self._made:
            try:
                be.bases_free(hnd)
            except Exception:
                pass
        self._made = []
        if self._r1cs:
            self.L.zl_r1cs_free(self.mb.ranks[0]._ctx, self._r1cs)
            self._r1cs = 0


class MultiBackend:
    """zl_mctx: G devices driven from one process (include/zl_backend.h, multi-GPU section).  ranks[g] is a Backend bound to rank g's
    ctx (upload / generate that rank's shard of the bases there); device ids may repeat (virtual ranks on one GPU, test mode)."""

    def __init__(self, device_ids):
        self.L = load_library()
        self._m = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        rc = self.L.zl_ctx_create_multi(C.byref(self._m), ids, len(device_ids))
        if rc:
            raise BackendError(rc, "zl_ctx_create_multi", self.L.zl_strerror(rc).decode())
        self.size = self.L.zl_mctx_size(self._m)
        self.ranks = [Backend(_borrowed_ctx=self.L.zl_mctx_ctx(self._m, g)) for g in range(self.size)]
        self.uses_rccl = bool(self.L.zl_mctx_uses_rccl(self._m))

    def msm_sharded(self, handles, d_scalars, counts, firsts=None) -> Tuple[np.ndarray, int]:
        """handles[g]: bases handle on rank g; d_scalars[g]: device pointer (on rank g's device) to counts[g] x 4 u64 canonical scalars"""
        G = self.size
        curve, group, _ = self.ranks[0]._bases[handles[0]]
        hs = np.array(handles, dtype=np.uint64)
        ptrs = (C.c_void_p * G)(*[int(p) for p in d_scalars])
        ns = (C.c_size_t * G)(*[int(c) for c in counts])
        fs = (C.c_size_t * G)(*[int(f) for f in (firsts or [0] * G)])
        out, inf = np.zeros(2 * group * FQ_LIMBS[curve], dtype=np.uint64), C.c_uint8(0)
        rc = self.L.zl_msm_sharded(self._m, _p64(hs), fs, ptrs, ns, _p64(out), C.byref(inf))
        if rc:
            raise BackendError(rc, "zl_msm_sharded", self.L.zl_strerror(rc).decode())
        return out, inf.value

    def groth16_shard_keys(self, curve: int, pk: dict, r1cs: dict, cuts=None) -> "ShardedGroth16Keys":
        """Spread a proving key over the ranks of this mctx for zl_groth16_prove_sharded.  pk: the key as HOST arrays of canonical affine points ('a_query' ... as
        (n, words) uint64, all-zero row = infinity) + the five single points; every query is cut into contiguous slices (cuts: optional fractions at which to
        cut, default equal parts), rank g uploads its slices to its own ctx, rank 0 uploads the matrices."""
        return ShardedGroth16Keys(self, curve, pk, r1cs, cuts)

    def groth16_prove_sharded(self, curve: int, pk: dict, r1cs: dict, assignment: np.ndarray, r: np.ndarray, s: np.ndarray, cuts=None):
        """upload + one proof + free (tests); returns (a, a_inf, b, b_inf, c, c_inf)"""
        keys = self.groth16_shard_keys(curve, pk, r1cs, cuts)
        try:
            return keys.prove(assignment, r, s)
        finally:
            keys.close()

    def ntt_sharded(self, curve: int, d_data, log_n: int, inverse: bool = False, coset: bool = False, mont: bool = False):
        G = self.size
        ptrs = (C.c_void_p * G)(*[int(p) for p in d_data])
        flags = (ZL_INVERSE if inverse else 0) | (ZL_COSET if coset else 0) | (ZL_MONT if mont else 0)
        rc = self.L.zl_ntt_sharded(self._m, curve, ptrs, log_n, flags)
        if rc:
            raise BackendError(rc, "zl_ntt_sharded", self.L.zl_strerror(rc).decode())

    def close(self):
        if self._m:
            for r in self.ranks:
                r.close()
            self.L.zl_mctx_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- test-only hooks (include/zl_backend_test.h): device Poseidon KAT, raw-limb field / point access ---------------------------------
TEST_ABI_SYMBOLS = ["zl_test_poseidon_permute_dev", "zl_test_fp28_op", "zl_test_fp28_bn_op", "zl_test_pairing_product", "zl_test_point_op", "zl_test_circuit_tweak", "zl_test_fq_mul_rate", "zl_test_fr28_op", "zl_test_fr29_op",
                    "zl_test_poseidon_permute_dev28r", "zl_test_fq_mul_clock", "zl_test_acc_clock", "zl_test_acc_clock_read", "zl_test_clock_probe_launch", "zl_test_clock_probe_read"]


def _p32(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def groth16_keys_parse(curve: int, data: bytes, check: bool = False) -> int:
    """host-only validation of ProvingContext bytes (zl_groth16_keys_parse): the C error code (0 = acceptable framing)"""
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    return int(load_library().zl_groth16_keys_parse(curve, buf, len(data), ZL_CHECK if check else 0))


def hook_poseidon_permute_dev(be: "Backend", curve: int, state: np.ndarray) -> np.ndarray:
    """width-3 Poseidon permutation computed on the device with the device Fr arithmetic (canonical (3,4) uint64 in / out)"""
    st = np.ascontiguousarray(state.copy(), dtype=np.uint64)
    be._check(be.L.zl_test_poseidon_permute_dev(be._ctx, curve, _p64(st)), "zl_test_poseidon_permute_dev")
    return st


def hook_poseidon_permute_dev28r(be: "Backend", curve: int, state: np.ndarray) -> np.ndarray:
    """the same permutation through the lazily reduced 10 x 28-bit Fr multiplier of the NTT passes (zl_field28r.h)"""
    st = np.ascontiguousarray(state.copy(), dtype=np.uint64)
    be._check(be.L.zl_test_poseidon_permute_dev28r(be._ctx, curve, _p64(st)), "zl_test_poseidon_permute_dev28r")
    return st


def hook_fp28_op(be: Optional["Backend"], op: int, operands: np.ndarray, bn254: bool = False) -> np.ndarray:
    """operands: (n, 4, L) uint32 raw limbs -> (n, L), L = 14 (BLS12-381 Fq) or 10 (bn254=True: BN254 Fq); be = None runs the host code path"""
    a = np.ascontiguousarray(operands, dtype=np.uint32)
    n, limbs = a.shape[0], (10 if bn254 else 14)
    assert a.shape[1:] == (4, limbs)
    out = np.zeros((n, limbs), dtype=np.uint32)
    L = load_library()
    fn = L.zl_test_fp28_bn_op if bn254 else L.zl_test_fp28_op
    rc = fn(be._ctx if be is not None else None, op, _p32(a), n, _p32(out))
    if rc:
        raise BackendError(rc, "zl_test_fp28_bn_op" if bn254 else "zl_test_fp28_op")
    return out


def hook_pairing_product(curve: int, ps: np.ndarray, qs: np.ndarray) -> np.ndarray:
    """prod_i e(P_i, Q_i) through the lock-step Miller loops (host only): ps (n, 2 FQ64) / qs (n, 4 FQ64) canonical u64 words -> 12 x FQ64 words"""
    p = np.ascontiguousarray(ps, dtype=np.uint64)
    q = np.ascontiguousarray(qs, dtype=np.uint64)
    n = p.shape[0]
    out = np.zeros((12, p.shape[1] // 2), dtype=np.uint64)
    rc = load_library().zl_test_pairing_product(curve, n, _p64(p), _p64(q), _p64(out))
    if rc:
        raise BackendError(rc, "zl_test_pairing_product")
    return out


def hook_point_op(be: Optional["Backend"], group: int, hot: bool, op: int, pq: np.ndarray) -> np.ndarray:
    """pq: (n, 8, W) uint32 raw limbs of two XYZZ points (W = 14 for G1, 28 for G2) -> (n, 4, W)"""
    a = np.ascontiguousarray(pq, dtype=np.uint32)
    n, W = a.shape[0], a.shape[2]
    out = np.zeros((n, 4, W), dtype=np.uint32)
    L = load_library()
    rc = L.zl_test_point_op(be._ctx if be is not None else None, group, int(hot), op, _p32(a), n, _p32(out))
    if rc:
        raise BackendError(rc, "zl_test_point_op")
    return out


def hook_fr28_op(be: Optional["Backend"], curve: int, op: int, ab: np.ndarray, j: int = 2, bits: int = 28) -> np.ndarray:
    """ab: (n, 2, 8) uint32 words of two values < 2^256 -> (n, 8) canonical result words; be = None runs the host code path;
    bits = 28: the 10 x 28-bit instance (R' = 2^280), 29: the 9 x 29-bit instance (R' = 2^261)"""
    a = np.ascontiguousarray(ab, dtype=np.uint32)
    n = a.shape[0]
    out = np.zeros((n, 8), dtype=np.uint32)
    L = load_library()
    fn = L.zl_test_fr28_op if bits == 28 else L.zl_test_fr29_op
    rc = fn(be._ctx if be is not None else None, curve, op, j, _p32(a), n, _p32(out))
    if rc:
        raise BackendError(rc, "zl_test_fr28_op" if bits == 28 else "zl_test_fr29_op")
    return out


def hook_fq_mul_rate(be: "Backend", waves_per_simd: int = 3, iters: int = 3000) -> float:
    """10^9 Montgomery products per second of the accumulation kernel's multiplier on per-lane pseudo-random operands (measurement hook)"""
    v = C.c_double(0.0)
    be._check(be.L.zl_test_fq_mul_rate(be._ctx, waves_per_simd, iters, C.byref(v)), "zl_test_fq_mul_rate")
    return v.value


_CLOCK_KEYS = ("effective_clock_ghz", "span_ms_by_100mhz_counter", "waves", "mean_wave_life_ms", "min_wave_ghz", "max_wave_ghz")


def hook_fq_mul_clock(be: "Backend", waves_per_simd: int = 3, iters: int = 50000) -> dict:
    """effective shader clock of the multiplier chain (s_memtime / s_memrealtime per wave) + its rate in the same launch (measurement hook)"""
    v = (C.c_double * 8)()
    be._check(be.L.zl_test_fq_mul_clock(be._ctx, waves_per_simd, iters, v), "zl_test_fq_mul_clock")
    d = {k: float(v[i]) for i, k in enumerate(_CLOCK_KEYS)}
    d["g_products_per_s"] = float(v[6])
    d["ms_by_hip_events"] = float(v[7])
    return d


def hook_clock_probe_launch(be: "Backend", spin_us: int) -> None:
    """start the sleeping clock probe (one wave per XCD, own stream) for spin_us microseconds; launch the work to observe right after"""
    be._check(be.L.zl_test_clock_probe_launch(be._ctx, int(spin_us)), "zl_test_clock_probe_launch")


def hook_clock_probe_read(be: "Backend") -> dict:
    v = (C.c_double * 6)()
    be._check(be.L.zl_test_clock_probe_read(be._ctx, v), "zl_test_clock_probe_read")
    return {k: float(v[i]) for i, k in enumerate(_CLOCK_KEYS)}


def hook_acc_clock(be: "Backend", on: bool) -> None:
    """arm / disarm the clock-reading build of the large G1 accumulation kernel on this ctx (measurement hook)"""
    be._check(be.L.zl_test_acc_clock(be._ctx, 1 if on else 0), "zl_test_acc_clock")


def hook_acc_clock_read(be: "Backend") -> dict:
    v = (C.c_double * 6)()
    be._check(be.L.zl_test_acc_clock_read(be._ctx, v), "zl_test_acc_clock_read")
    return {k: float(v[i]) for i, k in enumerate(_CLOCK_KEYS)}


# ---- host mirror (openzl::R1CS / poseidon / Groth16<E>, csrc/zl_host.h) through its C hooks ---------------------------
def poseidon_permute(curve: int, state: np.ndarray) -> np.ndarray:
    """native width-3 Poseidon permutation on canonical (3,4) uint64 limbs (host code, no GPU)."""
    st = np.ascontiguousarray(state.copy(), dtype=np.uint64)
    rc = load_library().zl_poseidon_permute(curve, _p64(st))
    if rc:
        raise BackendError(rc, "zl_poseidon_permute")
    return st


def pairing(curve: int, p_xy: np.ndarray, q_xy: np.ndarray) -> np.ndarray:
    """e(P, Q) as 12 canonical Fq coefficients (host pairing, no GPU)"""
    out = np.zeros((12, FQ_LIMBS[curve]), dtype=np.uint64)
    rc = load_library().zl_pairing(curve, _p64(np.ascontiguousarray(p_xy, dtype=np.uint64)), _p64(np.ascontiguousarray(q_xy, dtype=np.uint64)), _p64(out))
    if rc:
        raise BackendError(rc, "zl_pairing")
    return out


def _proof_struct(proof) -> "G16ProofC":
    pc = G16ProofC()
    a, ai, b, bi, c, ci = proof
    for i, v in enumerate(a):
        pc.a[i] = int(v)
    for i, v in enumerate(b):
        pc.b[i] = int(v)
    for i, v in enumerate(c):
        pc.c[i] = int(v)
    pc.a_inf, pc.b_inf, pc.c_inf = int(ai), int(bi), int(ci)
    return pc


def point_to_bytes(curve: int, group: int, xy: np.ndarray, inf: int = 0) -> bytes:
    """arkworks CanonicalSerialize (compressed) of one affine point given as canonical limbs (host code, no GPU)."""
    L = load_library()
    out = (C.c_uint8 * L.zl_point_bytes(curve, group))()
    rc = L.zl_point_to_bytes(curve, group, _p64(np.ascontiguousarray(xy, dtype=np.uint64)), int(inf), out)
    if rc:
        raise BackendError(rc, "zl_point_to_bytes")
    return bytes(out)


def point_from_bytes(curve: int, group: int, data: bytes):
    L = load_library()
    if len(data) != L.zl_point_bytes(curve, group):
        raise BackendError(-1, "zl_point_from_bytes", "wrong length")
    xy = np.zeros(2 * group * FQ_LIMBS[curve], dtype=np.uint64)
    inf = C.c_uint8(0)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = L.zl_point_from_bytes(curve, group, buf, _p64(xy), C.byref(inf))
    if rc:
        raise BackendError(rc, "zl_point_from_bytes")
    return xy, inf.value


def point_to_bytes_uncompressed(curve: int, group: int, xy: np.ndarray, inf: int = 0) -> bytes:
    """arkworks serialize_uncompressed of one affine point given as canonical limbs (host code, no GPU)."""
    L = load_library()
    out = (C.c_uint8 * L.zl_point_bytes_uncompressed(curve, group))()
    rc = L.zl_point_to_bytes_uncompressed(curve, group, _p64(np.ascontiguousarray(xy, dtype=np.uint64)), int(inf), out)
    if rc:
        raise BackendError(rc, "zl_point_to_bytes_uncompressed")
    return bytes(out)


def point_from_bytes_uncompressed(curve: int, group: int, data: bytes, check: bool = False):
    L = load_library()
    if len(data) != L.zl_point_bytes_uncompressed(curve, group):
        raise BackendError(-1, "zl_point_from_bytes_uncompressed", "wrong length")
    xy = np.zeros(2 * group * FQ_LIMBS[curve], dtype=np.uint64)
    inf = C.c_uint8(0)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = L.zl_point_from_bytes_uncompressed(curve, group, buf, int(check), _p64(xy), C.byref(inf))
    if rc:
        raise BackendError(rc, "zl_point_from_bytes_uncompressed")
    return xy, inf.value


def proof_to_bytes(curve: int, proof) -> bytes:
    """proof = (a, a_inf, b, b_inf, c, c_inf) as returned by Groth16Keys.prove -> 192 (BLS12-381) / 128 (BN254) bytes"""
    L = load_library()
    out = (C.c_uint8 * L.zl_groth16_proof_bytes(curve))()
    pc = _proof_struct(proof)
    rc = L.zl_groth16_proof_to_bytes(curve, C.byref(pc), out)
    if rc:
        raise BackendError(rc, "zl_groth16_proof_to_bytes")
    return bytes(out)


def proof_from_bytes(curve: int, data: bytes):
    L = load_library()
    pc = G16ProofC()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    rc = L.zl_groth16_proof_from_bytes(curve, buf, len(data), C.byref(pc))
    if rc:
        raise BackendError(rc, "zl_groth16_proof_from_bytes")
    nq = FQ_LIMBS[curve]
    return (np.array(pc.a[: 2 * nq], dtype=np.uint64), pc.a_inf, np.array(pc.b[: 4 * nq], dtype=np.uint64), pc.b_inf,
            np.array(pc.c[: 2 * nq], dtype=np.uint64), pc.c_inf)


class Circuit:
    """R1CS<F> compiler in proof mode holding the config-5 Poseidon-chain circuit (host code, no GPU)."""

    def __init__(self, curve: int, k: int, x0: int = 1, x1: int = 2, witness_only: bool = False):
        """witness_only: run the circuit in a witness-only compiler (R1CS::for_witness: values, no rows) -- for proofs against keys that already hold the matrices"""
        self.L = load_library()
        self.curve = curve
        self.witness_only = witness_only
        self._c = C.c_void_p()
        a = np.array([[(x0 >> (64 * j)) & (2**64 - 1) for j in range(4)]], dtype=np.uint64)
        b = np.array([[(x1 >> (64 * j)) & (2**64 - 1) for j in range(4)]], dtype=np.uint64)
        fn = self.L.zl_circuit_poseidon_chain_witness if witness_only else self.L.zl_circuit_poseidon_chain
        rc = fn(curve, k, _p64(a), _p64(b), C.byref(self._c))
        if rc:
            raise BackendError(rc, "zl_circuit_poseidon_chain")
        self._view = R1csC()
        zp = u64p()
        self.L.zl_circuit_export(self._c, C.byref(self._view), C.byref(zp))
        self._zp = zp

    @property
    def shape(self):
        return self._view.n_constraints, self._view.n_instance, self._view.n_witness

    def is_satisfied(self) -> bool:
        return self.L.zl_circuit_is_satisfied(self._c) == 1

    def arrays(self) -> dict:
        """copy the CSR view + assignment into numpy (same dict layout Backend.groth16_prove takes)"""
        v = self._view
        out = {"n_constraints": v.n_constraints, "n_instance": v.n_instance, "n_witness": v.n_witness}
        for m, key in enumerate("ABC"):
            ptr = np.ctypeslib.as_array(v.row_ptr[m], shape=(v.n_constraints + 1,)).copy()
            nnz = int(ptr[-1])
            if nnz == 0:  # (a witness-only circuit exports no rows)
                col, val = np.zeros(0, dtype=np.uint32), np.zeros((0, 4), dtype=np.uint64)
            else:
                col = np.ctypeslib.as_array(v.col[m], shape=(nnz,)).copy()
                val = np.ctypeslib.as_array(v.val[m], shape=(nnz * 4,)).copy().reshape(nnz, 4)
            out[key] = (ptr, col, val)
        nv = v.n_instance + v.n_witness
        out["assignment"] = np.ctypeslib.as_array(self._zp, shape=(nv * 4,)).copy().reshape(nv, 4)
        return out

    def close(self):
        if self._c:
            self.L.zl_circuit_free(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Groth16Keys:
    """Groth16<E>::compile result (ProvingContext on the device), or -- from_bytes -- a ProvingContext decoded from its wire format."""

    def __init__(self, backend: Backend, circuit: Circuit, seed: int, _encoded: bytes = None, _flags: int = 0):
        self.L = backend.L
        self.backend = backend
        self.circuit = circuit
        self._k = C.c_void_p()
        if _encoded is None:
            backend._check(self.L.zl_groth16_compile(backend._ctx, circuit._c, seed, C.byref(self._k)), "zl_groth16_compile")
        else:
            buf = (C.c_uint8 * max(1, len(_encoded))).from_buffer_copy(_encoded if _encoded else b"\0")
            backend._check(self.L.zl_groth16_keys_from_bytes(backend._ctx, circuit.curve, buf, len(_encoded), _flags, C.byref(self._k)),
                           "zl_groth16_keys_from_bytes")
        self.pk = G16PkC()
        self.L.zl_groth16_keys_pk(self._k, C.byref(self.pk))
        n_c, n_i, n_w = circuit.shape
        nv = n_i + n_w
        log_n = max(1, (n_c + n_i - 1).bit_length())
        for name, cnt, grp in (("a_query", nv, ZL_G1), ("b_g1_query", nv, ZL_G1), ("h_query", (1 << log_n) - 1, ZL_G1), ("l_query", n_w, ZL_G1),
                               ("b_g2_query", nv, ZL_G2)):
            backend._bases[getattr(self.pk, name)] = (circuit.curve, grp, cnt)  # so Backend.bases_download works on them

    @classmethod
    def from_bytes(cls, backend: Backend, circuit: Circuit, data: bytes, check: bool = False) -> "Groth16Keys":
        """codec::Decode for ProvingContext (zl_groth16_keys_from_bytes); `circuit` is the circuit the key belongs to (proofs need it)"""
        return cls(backend, circuit, 0, _encoded=data, _flags=ZL_CHECK if check else 0)

    def _bytes_of(self, fn, what: str) -> bytes:
        n = C.c_size_t(0)
        self.backend._check(fn(self._k, None, 0, C.byref(n)), what)
        out = (C.c_uint8 * max(1, n.value))()
        self.backend._check(fn(self._k, out, n.value, C.byref(n)), what)
        return C.string_at(C.addressof(out), n.value)

    def to_bytes(self) -> bytes:
        """codec::Encode for ProvingContext: ark_groth16::ProvingKey serialize_unchecked (uncompressed)"""
        return self._bytes_of(self.L.zl_groth16_keys_to_bytes, "zl_groth16_keys_to_bytes")

    def vk_to_bytes(self) -> bytes:
        """ark_groth16::VerifyingKey serialize (compressed)"""
        return self._bytes_of(self.L.zl_groth16_vk_to_bytes, "zl_groth16_vk_to_bytes")

    def trapdoor(self):
        out = np.zeros((5, 4), dtype=np.uint64)
        self.backend._check(self.L.zl_groth16_keys_trapdoor(self._k, _p64(out)), "zl_groth16_keys_trapdoor")
        return [sum(int(v) << (64 * j) for j, v in enumerate(row)) for row in out]

    def pk_dict(self) -> dict:
        nq = FQ_LIMBS[self.circuit.curve]
        d = {k: getattr(self.pk, k) for k in ("a_query", "b_g1_query", "h_query", "l_query", "b_g2_query")}
        for k, n in (("alpha_g1", 2 * nq), ("beta_g1", 2 * nq), ("delta_g1", 2 * nq), ("beta_g2", 4 * nq), ("delta_g2", 4 * nq)):
            d[k] = np.ctypeslib.as_array(getattr(self.pk, k), shape=(n,)).copy()
        return d

    def prove(self, seed: int, circuit: Optional["Circuit"] = None, lane: Optional[Backend] = None):
        """Groth16::prove(context, compiler, rng=SplitMix64(seed)) -> ((a, a_inf, b, b_inf, c, c_inf), r, s); circuit: another compiler of the SAME circuit
        (e.g. a witness-only one with a new witness) instead of the one the keys were built with; lane: a fork() of the keys' backend to run on (one
        per host thread: concurrent proofs over one device-resident key)"""
        proof = G16ProofC()
        r = np.zeros(4, dtype=np.uint64)
        s = np.zeros(4, dtype=np.uint64)
        self.backend._check(self.L.zl_groth16_prove_circuit((lane or self.backend)._ctx, self._k, (circuit or self.circuit)._c, seed, C.byref(proof), _p64(r), _p64(s)),
                            "zl_groth16_prove_circuit")
        nq = FQ_LIMBS[self.circuit.curve]
        return (np.array(proof.a[: 2 * nq], dtype=np.uint64), proof.a_inf, np.array(proof.b[: 4 * nq], dtype=np.uint64), proof.b_inf,
                np.array(proof.c[: 2 * nq], dtype=np.uint64), proof.c_inf), r, s

    def prove_many(self, seeds, circuits=None):
        """zl_groth16_prove_circuits: a stream of proofs over this key on two prover lanes (two host threads inside the library); proofs[i] is what
        prove(seeds[i], circuits[i]) returns.  circuits: None (the keys' own compiler for every proof) or one compiler per seed."""
        n = len(seeds)
        circuits = list(circuits) if circuits is not None else [self.circuit] * n
        assert len(circuits) == n
        cs = (C.c_void_p * max(1, n))(*[c._c for c in circuits])
        sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
        out = (G16ProofC * max(1, n))()
        self.backend._check(self.L.zl_groth16_prove_circuits(self.backend._ctx, self._k, cs, _p64(sd), n, out), "zl_groth16_prove_circuits")
        nq = FQ_LIMBS[self.circuit.curve]
        return [(np.array(p.a[: 2 * nq], dtype=np.uint64), p.a_inf, np.array(p.b[: 4 * nq], dtype=np.uint64), p.b_inf, np.array(p.c[: 2 * nq], dtype=np.uint64), p.c_inf)
                for p in out[:n]]

    def verify(self, proof, public_inputs: np.ndarray) -> bool:
        """Groth16::verify(vk, input, proof) with the host pairing; proof = (a, a_inf, b, b_inf, c, c_inf)"""
        pc = _proof_struct(proof)
        pub = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
        ok = C.c_int(0)
        self.backend._check(self.L.zl_groth16_verify(self._k, _p64(pub), pub.shape[0], C.byref(pc), C.byref(ok)), "zl_groth16_verify")
        return bool(ok.value)

    def close(self):
        if self._k:
            self.L.zl_groth16_keys_free(self._k)
            self._k = C.c_void_p()
