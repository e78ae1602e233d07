"""CPU: static cross-check of the three descriptions of the C ABI — include/gpmi355.h (the contract), abstractgps.jl_amd/_lib.py
(the ctypes prototypes the tests call through) and abstractgps.jl_amd/julia/HipGPs.jl (the reference-side binding, which cannot be
executed in the build image: Julia is not installed).  Every `ccall((:sym, libgpmi355), Ret, (ArgTypes...), args...)` of the shim
must name a declared symbol, pass as many arguments as it declares types, and agree with the header's prototype in arity, return
class and the class of every argument (pointer / int32 / int64 / double); the same for every ctypes prototype."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "gpmi355.h"
SHIM = ROOT / "abstractgps.jl_amd" / "julia" / "HipGPs.jl"


def _c_class(decl: str) -> str:
    decl = decl.strip()
    if "*" in decl:
        return "ptr"
    base = decl.replace("const", " ").split()
    base = base[0] if base else ""
    return {"int32_t": "i32", "int64_t": "i64", "double": "f64", "void": "void"}[base]


def header_prototypes() -> dict:
    txt = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)\b(gpd?_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argl = [] if args in ("", "void") else [a for a in args.split(",")]
        out[name] = (_c_class(ret), [_c_class(a) for a in argl])
    return out


def _split_top(s: str) -> list:
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return [p for p in parts if p]


def _balanced(s: str, i: int) -> int:
    """index just past the bracket group opening at s[i]"""
    depth = 0
    for j in range(i, len(s)):
        if s[j] in "({[":
            depth += 1
        elif s[j] in ")}]":
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced")


def _jl_class(t: str) -> str:
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "ptr"
    return {"Int32": "i32", "Int64": "i64", "Float64": "f64", "Cvoid": "void"}[t]


def shim_ccalls() -> list:
    src = SHIM.read_text()
    src = "\n".join(line.split("#")[0] if "ccall" not in line.split("#")[0] and line.lstrip().startswith("#") else line for line in src.splitlines())
    calls = []
    for m in re.finditer(r"ccall\(\(:([A-Za-z0-9_]+),\s*libgpmi355\),", src):
        start = m.start() + len("ccall")
        end = _balanced(src, start)
        inner = src[start + 1:end - 1]
        parts = _split_top(inner)  # [(:sym, lib)] [Ret] [(ArgTypes)] args...
        assert parts[0].startswith("(:"), parts[0]
        ret, tup = parts[1], parts[2]
        assert tup.startswith("(") and tup.endswith(")"), (m.group(1), tup)
        types = _split_top(tup[1:-1])
        calls.append({"sym": m.group(1), "ret": ret, "types": types, "nargs": len(parts) - 3, "line": src.count("\n", 0, m.start()) + 1})
    return calls


def test_header_parses_every_declared_function(agp):
    protos = header_prototypes()
    assert set(protos) == set(agp._lib.header_functions())


def test_ctypes_prototypes_match_the_header(agp):
    protos = header_prototypes()

    def cls(t):
        if t in (C.c_int32,):
            return "i32"
        if t in (C.c_int64,):
            return "i64"
        if t in (C.c_double,):
            return "f64"
        return "ptr"  # c_void_p, c_char_p, POINTER(...)

    for name, (res, args) in agp._lib.PROTOTYPES.items():
        hret, hargs = protos[name]
        assert cls(res) == hret, name
        assert [cls(a) for a in args] == hargs, (name, [cls(a) for a in args], hargs)


def test_every_shim_ccall_matches_the_header():
    protos = header_prototypes()
    calls = shim_ccalls()
    assert len(calls) >= 30, len(calls)
    for c in calls:
        where = f"HipGPs.jl:{c['line']} {c['sym']}"
        assert c["sym"] in protos, f"{where}: not declared in include/gpmi355.h"
        hret, hargs = protos[c["sym"]]
        assert c["nargs"] == len(c["types"]), f"{where}: {len(c['types'])} argument types but {c['nargs']} arguments"
        assert len(c["types"]) == len(hargs), f"{where}: arity {len(c['types'])} vs header {len(hargs)}"
        assert _jl_class(c["ret"]) == hret, f"{where}: return {c['ret']} vs header {hret}"
        got = [_jl_class(t) for t in c["types"]]
        assert got == hargs, f"{where}: argument classes {got} vs header {hargs}"


def test_shim_binds_what_integration_md_says_it_binds():
    """Every `gp_*` entry point INTEGRATION.md lists in the shim table is really called by the shim (round 3 listed three that only
    the Python mirror bound)."""
    bound = {c["sym"] for c in shim_ccalls()}
    txt = (ROOT / "INTEGRATION.md").read_text()
    table = [ln for ln in txt.splitlines() if ln.startswith("|") and "gp_" in ln]
    listed = set()
    for ln in table:
        listed |= set(re.findall(r"`(gp_[a-z0-9_]+)`", ln.split("|")[-2]))
    assert listed, "no entry points found in the INTEGRATION.md table"
    missing = sorted(s for s in listed if s not in bound and s not in ("gp_kernelmatrix", "gp_ctx_multi_info"))
    assert not missing, f"listed as shim-bound but never ccall'ed in HipGPs.jl: {missing}"


def test_shim_data_property_mirrors_the_reference_cache_fields():
    """src/sparse_approximations.jl:73: cache = (m_ε, Λ_ε, U, α, b_y, B_εf, x, Σy); test/sparse_approximations.jl:48-55 reads
    m_ε, Λ_ε.U, U, α, b_y.  B_εf (M×N) is deliberately not materialised."""
    src = SHIM.read_text()
    m = re.search(r"Base\.propertynames\(::HipVfeCache\) = \((.*?)\)\n", src)
    assert m, "the lazy cache view HipVfeCache is not there"
    assert [f.strip() for f in m.group(1).split(",")] == [":m_ε", ":Λ_ε", ":U", ":α", ":b_y", ":x", ":Σy"]
    body = src[src.index("function Base.getproperty(c::HipVfeCache"):src.index("Base.getproperty(f::HipApproxPosteriorGP")]
    for field in (":α", ":m_ε", ":U", ":Λ_ε", ":b_y", ":x", ":Σy"):  # every advertised field is served, each by the call that owns it
        assert f"s === {field}" in body, field
    assert "Cholesky(A, 'U', 0)" in body and "UpperTriangular(A)" in body
    assert "s === :data ? HipVfeCache(f)" in src  # `post.data` itself moves nothing
    assert "LinearAlgebra.logdet(C::DeviceCholesky)" in src
