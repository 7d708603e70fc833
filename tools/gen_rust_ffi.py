#!/usr/bin/env python3
"""Generate plugins/arkworks-mi355x/src/ffi.rs -- the complete `extern "C"` block of include/zl_backend.h + include/zl_backend_ext.h (every function, struct,
enum constant and flag) -- so that the Rust shim can never drift from the header: tests/test_abi.py regenerates it and compares.
    python tools/gen_rust_ffi.py [--check]
The Rust itself cannot be compiled in this image (no cargo / rustc); the generator guarantees names, arity and pointer shapes only."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "zl_backend.h")          # the drop-in boundary
HDR_EXT = os.path.join(ROOT, "include", "zl_backend_ext.h")  # everything beyond it (same library)
OUT = os.path.join(ROOT, "plugins", "arkworks-mi355x", "src", "ffi.rs")

SCALAR = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "size_t": "usize", "long": "core::ffi::c_long", "uint64_t": "u64", "uint32_t": "u32",
          "uint8_t": "u8", "float": "f32", "char": "core::ffi::c_char", "void": "core::ffi::c_void", "zl_curve_t": "i32", "zl_group_t": "i32"}
OPAQUE = ["zl_ctx", "zl_mctx", "zl_circuit", "zl_g16_keys"]
STRUCTS = ["zl_r1cs", "zl_g16_pk", "zl_g16_shard", "zl_g16_proof", "zl_timing"]


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def rust_type(ctype: str) -> str:
    """C type (already without the parameter name) -> Rust type.  Pointers are read right to left: `const void* const*` is a const pointer
    to const pointers to const void."""
    t = " ".join(ctype.replace("*", " * ").split())
    toks = t.split(" ")
    # split into base (up to the first '*') and pointer suffixes
    if "*" in toks:
        i = toks.index("*")
        base, rest = toks[:i], toks[i:]
    else:
        base, rest = toks, []
    base_const = "const" in base
    name = " ".join(x for x in base if x not in ("const", "struct"))
    rt = SCALAR.get(name, name)
    assert rt in SCALAR.values() or rt in OPAQUE or rt in STRUCTS, f"unknown C type {ctype!r}"
    const = base_const
    k = 0
    while k < len(rest):
        assert rest[k] == "*"
        rt = ("*const " if const else "*mut ") + rt
        const = k + 1 < len(rest) and rest[k + 1] == "const"
        k += 2 if const else 1
    return rt


def split_param(p: str):
    p = p.strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?$", p)
    ctype, name, arr = m.group(1).strip(), m.group(2), m.group(3)
    if arr:  # array parameter decays to a pointer
        ctype += "*"
    return ctype, name


RUST_KEYWORDS = {"in": "input", "type": "kind", "ref": "reference", "match": "matched", "fn": "func", "mod": "modulus"}


def functions(text: str):
    out = []
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*(zl_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        ps = []
        if params and params != "void":
            for p in params.split(","):
                ctype, pname = split_param(p)
                ps.append((RUST_KEYWORDS.get(pname, pname), rust_type(ctype)))
        rret = None if ret == "void" else rust_type(ret)
        out.append((name, ps, rret))
    return out


def structs(text: str):
    out = []
    for m in re.finditer(r"typedef\s+struct\s+(zl_[a-z0-9_]+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # "const uint64_t *a, *b, *c" / "uint64_t a[12], b[24]" / "uint32_t n, m"
            m2 = re.match(r"^((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*)\s*(.*)$", decl)
            base, rest = m2.group(1), m2.group(2)
            for item in rest.split(","):
                item = item.strip()
                stars = len(item) - len(item.lstrip("* "))
                nstar = item[:stars].count("*")
                core = item[stars:].strip()
                am = re.match(r"^([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[(\d+)\])?$", core)
                fname, alen = am.group(1), am.group(2)
                rt = rust_type(base + " " + "*" * nstar)
                if alen:
                    rt = f"[{rt}; {alen}]"
                fields.append((fname, rt))
        out.append((name, fields))
    return out


def constants(text_raw: str, text: str):
    out = []
    for m in re.finditer(r"typedef\s+enum\s*(?:ZL_ENUM_INT\s*)?\{([^}]*)\}\s*(zl_[a-z_]+)\s*;", text):
        for item in m.group(1).split(","):
            k, v = [x.strip() for x in item.split("=")]
            out.append((k, "i32", v))
    for m in re.finditer(r"(?<!typedef )enum\s*\{([^}]*)\}\s*;", text):
        for item in m.group(1).split(","):
            if "=" in item:
                k, v = [x.strip() for x in item.split("=")]
                out.append((k, "i32", v))
    for m in re.finditer(r"(?m)^#define\s+(ZL_[A-Z_0-9]+)\s+(\d+)(u?)\s*(?:/\*.*)?$", text_raw):
        out.append((m.group(1), "u32" if m.group(3) else "usize", m.group(2)))
    return out


def generate() -> str:
    raw = open(HDR).read() + "\n" + open(HDR_EXT).read()
    text = strip_comments(raw)
    boundary = {n for n, _, _ in functions(strip_comments(open(HDR).read()))}
    lines = ["// GENERATED by tools/gen_rust_ffi.py from include/zl_backend.h (the boundary) and include/zl_backend_ext.h -- do not edit; tests/test_abi.py::test_rust_ffi_matches_header",
             "// regenerates this file and fails on any difference.  UNTESTED as Rust: this image has no cargo / rustc.",
             "#![allow(non_camel_case_types, dead_code)]", ""]
    for o in OPAQUE:
        lines.append(f"#[repr(C)] pub struct {o} {{ _private: [u8; 0] }}")
    lines.append("")
    for name, fields in structs(text):
        lines.append("#[repr(C)]")
        lines.append(f"pub struct {name} {{")
        for f, t in fields:
            lines.append(f"    pub {f}: {t},")
        lines.append("}")
    lines.append("")
    for k, t, v in constants(raw, text):
        lines.append(f"pub const {k}: {t} = {v};")
    lines.append("")
    lines.append('#[link(name = "zl_backend")]')
    lines.append('extern "C" {')
    fns = functions(text)
    for part, title in ((True, "include/zl_backend.h: the drop-in boundary"), (False, "include/zl_backend_ext.h: pipelines, shards, lanes, host-mirror hooks, codecs, timing")):
        lines.append(f"    // ---- {title}")
        for name, ps, ret in fns:
            if (name in boundary) != part:
                continue
            args = ", ".join(f"{n}: {t}" for n, t in ps)
            lines.append(f"    pub fn {name}({args})" + (f" -> {ret};" if ret else ";"))
    lines.append("}")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    src = generate()
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == src
        print("ffi.rs is up to date" if ok else "ffi.rs is STALE: run python tools/gen_rust_ffi.py")
        sys.exit(0 if ok else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(src)
    print("wrote", OUT)
