"""CPU tests of the C-ABI library: it loads, exports every symbol include/zl_backend.h declares, reports errors by
code, and its host tail (zl_partials_sum: XYZZ fold + affine normalisation) matches the oracle.  No GPU compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po
from openzl_amd.backend import ABI_SYMBOLS, TEST_ABI_SYMBOLS, ZL_PARTIAL_WORDS, load_library
from openzl_amd.sharded import fold_partials

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = load_library()
    pat = r"^\s*(?:const char\*|zl_ctx\*|int|void|size_t)\s+(zl_[a-z0-9_]+)\s*\("
    boundary = set(re.findall(pat, open(os.path.join(ROOT, "include", "zl_backend.h")).read(), flags=re.M))
    ext = set(re.findall(pat, open(os.path.join(ROOT, "include", "zl_backend_ext.h")).read(), flags=re.M))
    declared = boundary | ext
    assert boundary and ext and not (boundary & ext), "header parse failed"
    # round 6: zl_backend.h is the drop-in boundary -- what a binding of plugins/arkworks/src/groth16.rs needs -- and stays small
    assert len(boundary) <= 25, sorted(boundary)
    assert {"zl_msm", "zl_ntt", "zl_groth16_prove", "zl_bases_upload", "zl_groth16_keys_from_bytes", "zl_groth16_proof_to_bytes"} <= boundary
    assert declared == set(ABI_SYMBOLS)
    for sym in declared:
        assert getattr(L, sym) is not None


def test_library_exports_every_declared_test_hook():
    """include/zl_backend_test.h: test-only hooks (device field KAT, raw-limb field / point access); host paths run without a GPU"""
    L = load_library()
    hdr = open(os.path.join(ROOT, "include", "zl_backend_test.h")).read()
    declared = set(re.findall(r"^\s*(?:const char\*|zl_ctx\*|int|void|size_t)\s+(zl_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(TEST_ABI_SYMBOLS)
    for sym in declared:
        assert getattr(L, sym) is not None
    assert L.zl_test_fp28_op(None, 99, None, 0, None) == -1
    assert L.zl_test_poseidon_permute_dev(None, 1, None) == -1


def test_strerror_and_argument_checks():
    L = load_library()
    assert L.zl_strerror(0) == b"ok"
    assert L.zl_strerror(-5) == b"unknown bases handle"
    assert L.zl_ctx_create(None, 0) == -1  # ZL_EINVAL
    assert L.zl_ctx_fork(None, None) == -1
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    assert L.zl_partials_sum(99, 1, None, 0, ol.p64(out), C.byref(inf)) == -1


def _partial_from_affine(curve, pt):
    """wrap an affine point as the backend's opaque partial (zl_partial_from_affine)"""
    xy = ol.points_to_limbs(curve, [pt])[0]
    out = np.zeros(ZL_PARTIAL_WORDS, dtype=np.uint64)
    assert load_library().zl_partial_from_affine(curve.cid, 1, ol.p64(xy), ol.p64(out)) == 0
    return out


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_partials_sum_host_tail_matches_oracle(curve):
    G = po.g1_generator(curve)
    pts = [po.g1_mul(curve, k, G) for k in (5, 7, 11, 5)] + [None, po.g1_neg(curve, po.g1_mul(curve, 7, G))]
    parts = np.stack([_partial_from_affine(curve, p) for p in pts])
    xy, inf = fold_partials(curve.cid, parts)
    exp = None
    for p in pts:
        exp = po.g1_add(curve, exp, p)
    assert ol.limbs_to_point(curve, xy, inf) == exp == po.g1_mul(curve, 21, G)
    # doubling inside the fold, cancellation to infinity, empty input
    xy, inf = fold_partials(curve.cid, parts[[0, 3]])
    assert ol.limbs_to_point(curve, xy, inf) == po.g1_mul(curve, 10, G)
    xy, inf = fold_partials(curve.cid, parts[[1, 5]])
    assert inf == 1 and not xy.any()
    xy, inf = fold_partials(curve.cid, parts[:0])
    assert inf == 1


def test_headers_are_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/*.h must compile as strict C99 (no C++-isms outside the extern "C" guards), and a C program
    must link against the library with nothing but -lzl_backend."""
    import subprocess

    src = tmp_path / "hdr.c"
    src.write_text('#include "zl_backend.h"\n#include "zl_backend_ext.h"\n#include "zl_backend_test.h"\n'
                   "int main(void) { zl_ctx* c = 0; zl_mctx* m = 0; zl_timing t; (void)c; (void)m; (void)t;\n"
                   "  return zl_strerror(ZL_OK) == 0; }\n")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "hdr.o")])
    libdir = os.path.join(ROOT, "openzl_amd")
    subprocess.check_call(["gcc", str(tmp_path / "hdr.o"), "-L", libdir, "-l:libzl_backend.so", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined",
                           "-o", str(tmp_path / "hdr")])


def test_rust_ffi_matches_header():
    """plugins/arkworks-mi355x/src/ffi.rs (the source-only Rust shim's extern "C" block) is generated from include/zl_backend.h by
    tools/gen_rust_ffi.py: the committed file must be what the generator produces now, name every exported symbol exactly once and give
    every function the arity of its C declaration.  (The Rust cannot be compiled here: no cargo / rustc; this keeps it from drifting.)"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(ROOT, "plugins", "arkworks-mi355x", "src", "ffi.rs")).read()
    assert committed == gen.generate(), "stale ffi.rs: run python tools/gen_rust_ffi.py"
    rust_fns = dict(re.findall(r"pub fn (zl_[a-z0-9_]+)\(([^)]*)\)", committed))
    assert set(rust_fns) == set(ABI_SYMBOLS)
    hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "zl_backend.h")).read() + open(os.path.join(ROOT, "include", "zl_backend_ext.h")).read(), flags=re.S)
    for name, args in rust_fns.items():
        m = re.search(r"\b" + name + r"\s*\(([^;{}]*?)\)\s*;", hdr, flags=re.S)
        assert m, name
        c_args = " ".join(m.group(1).split())
        n_c = 0 if c_args in ("", "void") else c_args.count(",") + 1
        n_rust = 0 if not args.strip() else args.count(",") + 1
        assert n_c == n_rust, (name, c_args, args)
    # the hand-written part of the shim only calls functions the generated block declares
    lib = open(os.path.join(ROOT, "plugins", "arkworks-mi355x", "src", "lib.rs")).read()
    for called in set(re.findall(r"ffi::(zl_[a-z0-9_]+)\(", lib)):
        assert called in rust_fns, called


def test_host_pool_serves_concurrent_callers(tmp_path):
    """csrc/zl_pool.h: one job per concurrent caller (round 6; it had one job slot, a second caller displaced the first one's job)"""
    import subprocess

    exe = tmp_path / "pool_stress"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "openzl_amd", "csrc"), os.path.join(ROOT, "tests", "c", "pool_stress.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "bad = 0" in out.stdout, (out.stdout, out.stderr)


def _build_inmemory_key(tmp_path):
    import subprocess

    exe = tmp_path / "inmemory_key"
    libdir = os.path.join(ROOT, "openzl_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "inmemory_key.c"), "-o", str(exe),
                           "-L", libdir, "-l:libzl_backend.so", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"])
    return exe


def test_inmemory_key_program_builds(tmp_path):
    """tests/c/inmemory_key.c (the C stand-in for the Rust shim's layout-independent key upload) compiles against the two public headers and links with -lzl_backend"""
    assert os.path.exists(_build_inmemory_key(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("curve,k", [("bls12_381", 3), ("bn254", 3), ("bls12_381", 40)])
def test_inmemory_key_upload_proves_like_the_library_key(tmp_path, curve, k):
    """a caller that holds arkworks' in-memory ProvingKey uploads the five queries with zl_bases_upload(stride = size_of::<GroupAffine>,
    inf_offset = offset_of!(infinity), ZL_MONT) -- G1 and G2 -- and assembles zl_g16_pk from the handles; the proof through that key equals the proof through
    the library's own key byte for byte and verifies.  Driven from C (no rustc here): tests/c/inmemory_key.c."""
    import subprocess

    out = subprocess.run([str(_build_inmemory_key(tmp_path)), curve, str(k)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), (out.stdout, out.stderr)
