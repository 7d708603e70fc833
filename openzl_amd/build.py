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
# Developer A/B builds: ZL_BUILD_TAG=x writes objects *.x.o and libzl_backend.x.so beside the product build (load it with ZL_BACKEND_LIB, tools/ab/*.sh)
TAG = os.environ.get("ZL_BUILD_TAG", "")
if TAG:
    LIB = os.path.join(HERE, f"libzl_backend.{TAG}.so")
# Round 5 experiment, OFF by default (ZL_STRIP_ASM_NOPS=1 turns it on for an A/B): hipcc pads EVERY inline-asm statement whose result the next instruction reads
# with `s_nop 0` (GCNHazardRecognizer treats an inline-asm def as a possible dst_sel / opsel partial write on gfx940+).  The product scans (zl_mul28*_gfx950.h) are
# chains of such statements holding only v_mad_u64_u32: 515 of the 5 150 instructions of one mixed addition are those pads.  With the switch on, the accumulation
# units are compiled to device assembly, the pads between one of OUR asm statements and an instruction that cannot carry that hazard are removed, and the result is
# assembled, linked, bundled and embedded exactly as hipcc does itself (hipcc -### shows the same five steps).  Measured (profiles/r05_nop_ab.log, interleaved on
# one box, all MSM / Groth16 parity tests green on the stripped build): 2^24 accumulation 31.7-32.1 ms with and without the pads, 2^16 / 2^20 / G2 / BN254 equal too --
# at three waves per SIMD the pads issue beside the other waves' VALU instructions.  The product build is hipcc's own output.
STRIP_NOPS = os.environ.get("ZL_STRIP_ASM_NOPS", "") == "1"
STRIP_VERSION = "strip-nops-v1"
_LLVM = "/opt/rocm/lib/llvm/bin"
STRIP_UNITS = tuple(u for u in os.environ.get("ZL_STRIP_UNITS", "zl_msm_acc_").split(",") if u)  # unit-name prefixes compiled that way
_SAFE_NEXT = re.compile(r"^(;;#ASMSTART|v_mul_lo_u32|v_lshrrev_b64|v_and_b32_e32|v_and_b32_e64)\b")
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


def strip_asm_nops(text: str) -> tuple[str, int]:
    """Remove `s_nop 0` lines that directly follow `;;#ASMEND` and directly precede another of our asm statements or one of the plain VALU instructions
    the product scans place there (m_k = lo * INV, the 28-bit mask, the column shift).  Every asm statement of this code base holds v_mad_u64_u32 only
    (gen_mul*.py; DPP moves of zl_quad.h are builtins, not asm), which writes full 32-bit registers: the forwarding hazard hipcc guards against cannot occur."""
    lines = text.split("\n")
    out, n = [], 0
    for i, l in enumerate(lines):
        if l.strip() == "s_nop 0" and i > 0 and lines[i - 1].strip() == ";;#ASMEND" and i + 1 < len(lines) and _SAFE_NEXT.match(lines[i + 1].strip()):
            n += 1
            continue
        out.append(l)
    return "\n".join(out), n


def _compile_stripped(name: str, src: str, defs: list[str], obj: str, verbose: bool) -> None:
    """hipcc's own pipeline for one HIP unit with the device assembly edited in between."""
    work = os.path.join(CSRC, ".asm")
    os.makedirs(work, exist_ok=True)
    base = os.path.join(work, name + (f".{TAG}" if TAG else ""))
    cuid = "-cuid=zl" + hashlib.sha256(name.encode()).hexdigest()[:14]
    common = [_hipcc()] + FLAGS + defs + [cuid]
    run = lambda cmd: (print("[build]", " ".join(cmd), flush=True) if verbose else None, subprocess.check_call(cmd, stderr=subprocess.DEVNULL if "-S" in cmd else None))
    run(common + ["--cuda-device-only", "-S", src, "-o", base + ".raw.s"])
    text, n = strip_asm_nops(open(base + ".raw.s").read())
    open(base + ".s", "w").write(text)
    os.remove(base + ".raw.s")
    if verbose:
        print(f"[build] {name}: removed {n} inline-asm pads", flush=True)
    run([os.path.join(_LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + ARCH, "-c", base + ".s", "-o", base + ".dev.o"])
    run([os.path.join(_LLVM, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", base + ".out", base + ".dev.o"])
    run([os.path.join(_LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096", f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{ARCH}",
         "-input=/dev/null", "-input=" + base + ".out", "-output=" + base + ".hipfb"])
    run(common + ["--cuda-host-only", "-c", src, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", base + ".hipfb", "-o", obj])
    for ext in (".dev.o", ".out", ".hipfb"):
        os.remove(base + ext)


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
    # the generated product scans (committed, like zl_params.h): regenerate when their generator is newer, so that an edit of gen_mul*.py can never
    # leave a stale header behind; the object digests then see the new text
    for g, outs in (("gen_mul28.py", ("zl_mul28_gfx950.h", "zl_mul28r_gfx950.h", "zl_mul29r_gfx950.h")), ("gen_mul.py", ("zl_mul_gfx950.h",))):
        gp = os.path.join(CSRC, g)
        if os.path.exists(gp) and any(not os.path.exists(os.path.join(CSRC, o)) or os.path.getmtime(os.path.join(CSRC, o)) < os.path.getmtime(gp) for o in outs):
            subprocess.check_call([sys.executable, gp], stdout=subprocess.DEVNULL)
    todo, objs = [], []
    for name, src, defs in _units():
        strip = STRIP_NOPS and name.startswith(STRIP_UNITS)
        obj = os.path.join(CSRC, name + (f".{TAG}" if TAG else "") + ".o")
        stamp = obj + ".sha"
        d = _digest(_deps(os.path.join(CSRC, src)), " ".join(FLAGS + defs) + (STRIP_VERSION if strip else ""))
        objs.append(obj)
        if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == d:
            continue
        todo.append((name, src, defs, obj, stamp, d, strip))

    def compile_one(t):
        name, src, defs, obj, stamp, d, strip = t
        if strip:
            _compile_stripped(name, os.path.join(CSRC, src), defs, obj, verbose)
        else:
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


# ---- sanitizer build of the HOST side (VERDICT r4 "next" item 7) -----------------------------------------------------------------------------
# zl_host.hip holds every parser of untrusted bytes (zl_groth16_keys_from_bytes, zl_point_from_bytes*: zl_serialize.h), the host pairing (zl_pairing.h) and the
# R1CS / Poseidon / Groth16 mirror; zl_capi.hip the handle and argument checks.  `python -m openzl_amd.build --host-asan` compiles the HOST half of those two
# units with -fsanitize=address,undefined (device code and every other unit: the product objects) and links libzl_backend.asan.so against the shared ASan
# runtime; tests/test_sanitizers.py runs the serialisation / pairing / host-mirror tests (and, on a GPU box, the key-wire corruption fuzz) against it with
# the runtime preloaded.  Any report aborts the process (-fno-sanitize-recover=all).
ASAN_UNITS = ("zl_host", "zl_capi")
ASAN_FLAGS = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-sanitize-recover=all", "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-g"]
ASAN_LIB = os.path.join(HERE, "libzl_backend.asan.so")


def asan_runtime() -> str:
    out = subprocess.check_output([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    if not os.path.isabs(out) or not os.path.exists(out):
        cands = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(_LLVM, "..", "lib", "clang")) for f in fs if f == "libclang_rt.asan-x86_64.so"]
        if not cands:
            raise RuntimeError("no shared ASan runtime (libclang_rt.asan-x86_64.so) beside hipcc")
        out = os.path.normpath(cands[0])
    return out


def build_host_asan(verbose: bool = True) -> str:
    build(verbose=verbose)  # the product objects of every other unit
    objs, todo = [], []
    for name, src, defs in _units():
        if name not in ASAN_UNITS:
            objs.append(os.path.join(CSRC, name + ".o"))
            continue
        obj = os.path.join(CSRC, name + ".asan.o")
        stamp = obj + ".sha"
        d = _digest(_deps(os.path.join(CSRC, src)), " ".join(FLAGS + defs + ASAN_FLAGS))
        objs.append(obj)
        if not (os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == d):
            todo.append((name, src, defs, obj, stamp, d))

    def one(t):
        name, src, defs, obj, stamp, d = t
        cmd = [_hipcc()] + FLAGS + defs + ASAN_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        open(stamp, "w").write(d)

    if todo:
        with ThreadPoolExecutor(max_workers=2) as ex:
            list(ex.map(one, todo))
    if todo or not os.path.exists(ASAN_LIB) or os.path.getmtime(ASAN_LIB) < os.path.getmtime(LIB):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libasan", "-o", ASAN_LIB] + objs + ["-ldl"]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return ASAN_LIB


if __name__ == "__main__":
    print(build_host_asan() if "--host-asan" in sys.argv else build())
