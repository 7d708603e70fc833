import os
import sys

import pytest


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The C-ABI library and the CPU oracle are build products (git-ignored): build them when missing so that a clean
    checkout can run `pytest -m "not gpu"` (hipcc cross-compiles gfx950 without a GPU)."""
    from openzl_amd import build as zb

    # always: build() is incremental (sha stamps over sources + headers), so an up-to-date tree costs milliseconds and a
    # stale library after a source edit cannot be tested by accident
    zb.build(verbose=False)
    import oracle_lib

    oracle_lib.build_oracle()


def _gpu_present() -> bool:
    try:
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


@pytest.fixture(scope="session")
def backend():
    """The HIP backend through the C ABI.  No fallback: a missing library or GPU is a hard failure under -m gpu."""
    from openzl_amd import Backend

    # torch ships its own HIP runtime: when both live in one process torch must initialise first (bench.py does the same),
    # otherwise torch.cuda sees no device.  Only the tests that shuffle device tensors (virtual-rank NTT) need torch at all.
    import torch

    if torch.cuda.is_available():
        torch.cuda.init()
    b = Backend(0)
    yield b
    b.close()
