"""Build libzl_backend.so (HIP, gfx950) in-tree with hipcc.  Incremental: objects are rebuilt only when a source or
header they include changed.  Used by __graft_entry__.build(); `python -m openzl_amd.build` works too."""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libzl_backend.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
FLAGS += [f for f in os.environ.get("ZL_EXTRA_FLAGS", "").split() if f]
GROUPS = ["BlsG1", "BnG1", "BlsG2", "BnG2"]
_INC = re.compile(r'^[ \t]*#[ \t]*include[ \t]*"([^"]+)"', re.M)


def _deps(src: str) -> list[str]:
    """Every file `src` reaches through quoted #include lines (recursively, whatever the preprocessor conditions say: device-only
    includes count too).  Derived from the sources on every build, so a new header can never be forgotten in a hand-kept list
    (round 3's zl_quad.h / zl_pool.h were)."""
    seen, todo = [], [os.path.normpath(src)]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        with open(f, encoding="utf-8", errors="replace") as fh:
            for inc in _INC.findall(fh.read()):
                todo.append(os.path.normpath(os.path.join(os.path.dirname(f), inc)))
    return sorted(seen)


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(paths, extra="") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _units():
    units = [("zl_capi", "zl_capi.hip", []), ("zl_ntt", "zl_ntt.hip", ["-DZL_INLINE_MUL"]), ("zl_groth16", "zl_groth16.hip", []), ("zl_host", "zl_host.hip", []), ("zl_testhooks", "zl_testhooks.hip", []), ("zl_multi", "zl_multi.hip", []),
             ("zl_msm_sort", "zl_msm_sort.hip", [])]  # the curve-independent sort kernels of the MSM: once, not per group
    for g in GROUPS:
        # Fq2 accumulators: 1 wave/SIMD register budget avoids scratch spills
        extra = ["-DZL_ACC_WAVES=1"] if g.endswith("G2") else []
        # G1 device code inlines the multiplier: -4.5 % on the accumulate kernel vs the out-of-line call (host code keeps the call).
        if g.endswith("G1"):
            extra = extra + ["-DZL_INLINE_MUL_DEVICE"]
        if g == "BlsG1":
            # three waves per SIMD (<= 168 registers): what the accumulation kernel needs anyway (162); without the cap the compiler spreads to 185 = two waves
            extra = extra + ["-DZL_ACC_WAVES=3"]
        # three units per group: the host side + light kernels, the accumulation kernels, the merge / reduction kernels (zl_msm.hip's header)
        for part in ("zl_msm", "zl_msm_acc", "zl_msm_tail"):
            units.append((f"{part}_{g}", part + ".hip", [f"-DZL_G={g}"] + extra))
    return [u for u in units if os.path.exists(os.path.join(CSRC, u[1]))]


def build(verbose: bool = True, jobs: int | None = None) -> str:
    params = os.path.join(CSRC, "zl_params.h")
    gen = os.path.join(CSRC, "gen_params.py")
    if not os.path.exists(params) or os.path.getmtime(params) < os.path.getmtime(gen):
        subprocess.check_call([sys.executable, gen])
    todo, objs = [], []
    for name, src, defs in _units():
        obj = os.path.join(CSRC, name + ".o")
        stamp = obj + ".sha"
        d = _digest(_deps(os.path.join(CSRC, src)), " ".join(FLAGS + defs))
        objs.append(obj)
        if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == d:
            continue
        todo.append((name, src, defs, obj, stamp, d))

    def compile_one(t):
        name, src, defs, obj, stamp, d = t
        cmd = [_hipcc()] + FLAGS + defs + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(d)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, todo))
    if todo or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]  # RCCL is dlopen'ed on first multi-GPU use
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build())
