"""openzl_amd -- MI355X (gfx950) MSM / NTT backend for OpenZL's arkworks Groth16 plugin path.

The product is the C-ABI shared library `libzl_backend.so` (include/zl_backend.h).  This package holds its HIP
sources (csrc/), the build driver (build.py) and a thin ctypes binding (backend.py) used by the tests and the
bench; it never falls back to a CPU implementation: without the HIP library or without a GPU it raises.
"""
from .backend import Backend, MultiBackend, BackendError, load_library, CURVES, ZL_BLS12_381, ZL_BN254, ZL_G1, ZL_G2  # noqa: F401
from .backend import ZL_MONT, ZL_COSET, ZL_INVERSE, ZL_CHECK  # noqa: F401
from .backend import Circuit, Groth16Keys, poseidon_permute, pairing  # noqa: F401
