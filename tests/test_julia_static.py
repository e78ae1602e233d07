"""CPU: static checks of the two Julia artefacts that cannot be executed in the build image (Julia is not installed) — the reference-side golden
generator tests/golden/make_golden.jl and the shim abstractgps.jl_amd/julia/HipGPs.jl — so that the first maintainer run does not die on a typo:
block / `end` balance and bracket balance (a lexer that knows strings, comments, `a[end]`, symbols and generator forms), every AbstractGPs name the
generator calls against the reference's export list (src/AbstractGPs.jl:19-35; KernelFunctions is re-exported, :8), every `data.<field>` it reads
against the NamedTuples the reference builds (src/exact_gpr_posterior.jl:34, src/sparse_approximations.jl:73)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GEN = ROOT / "tests" / "golden" / "make_golden.jl"
SHIM = ROOT / "abstractgps.jl_amd" / "julia" / "HipGPs.jl"
REF = Path("/root/reference/src")

# src/AbstractGPs.jl:19-35 (AbstractGPs 0.5.24)
EXPORTS = {"GP", "LatentGP", "VFE", "DTC", "ZeroMean", "ConstMean", "CustomMean", "rand!", "mean", "cov", "var", "std", "mean_and_cov", "mean_and_var",
           "mean_vector", "marginals", "logpdf", "approx_log_evidence", "elbo", "dtc", "posterior", "update_posterior", "ColVecs", "RowVecs"}
# KernelFunctions names the generator uses (re-exported by `@reexport using KernelFunctions`, src/AbstractGPs.jl:8)
KERNELFUNCTIONS = {"SqExponentialKernel", "Matern12Kernel", "Matern32Kernel", "Matern52Kernel", "ScaleTransform", "ARDTransform"}
EXACT_FIELDS = {"α", "C", "x", "δ"}                                   # src/exact_gpr_posterior.jl:34
VFE_FIELDS = {"m_ε", "Λ_ε", "U", "α", "b_y", "B_εf", "x", "Σy"}       # src/sparse_approximations.jl:73

OPENERS = {"function", "if", "for", "while", "try", "begin", "let", "do", "struct", "module", "baremodule", "quote", "macro", "type"}


def lex(src: str):
    """(kind, text, line) tokens: identifiers / keywords, brackets; strings, chars, comments dropped.  Julia specifics handled: `#= =#` and `#`
    comments, triple and single quoted strings with escapes and $(...) interpolation left opaque, the transpose quote after an identifier / closer,
    `:sym` symbols (so `:end` / `:if` are not keywords), field access `.end`."""
    toks, i, n, line = [], 0, len(src), 1
    prev_sig = ""  # last significant character class: decides whether ' is a transpose
    while i < n:
        ch = src[i]
        if ch == "\n":
            line += 1
            i += 1
            continue
        if ch in " \t\r":
            i += 1
            continue
        if src.startswith("#=", i):
            j = src.index("=#", i + 2)
            line += src.count("\n", i, j)
            i = j + 2
            continue
        if ch == "#":
            while i < n and src[i] != "\n":
                i += 1
            continue
        if src.startswith('"""', i):
            j = src.index('"""', i + 3)
            line += src.count("\n", i, j)
            i = j + 3
            prev_sig = "s"
            continue
        if ch == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            line += src.count("\n", i, j)
            i = j + 1
            prev_sig = "s"
            continue
        if ch == "'":
            if prev_sig in ("w", ")", "]", "}", "'"):  # transpose
                i += 1
                prev_sig = "'"
                continue
            j = i + 1
            while src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            i = j + 1
            prev_sig = "s"
            continue
        m = re.compile(r"[^\W\d]\w*[!?]?", re.UNICODE).match(src, i)
        if m:
            word = m.group(0)
            before = src[i - 1] if i > 0 else ""
            before2 = src[i - 2] if i > 1 else ""
            is_symbol = before == ":" and before2 != ":" and not (before2.isalnum() or before2 in ")]}_")  # :end, (:gp_x, lib) — not a ? b : c, not A::T
            kind = "sym" if is_symbol else ("field" if before == "." else "w")
            if kind == "w" and src[m.end():m.end() + 1] == "(":
                kind = "call"  # an identifier immediately followed by "(" (block keywords cannot be: `if (` is written with a space here)
            toks.append((kind, word, line))
            i = m.end()
            prev_sig = "w"
            continue
        if ch in "([{":
            toks.append(("open", ch, line))
        elif ch in ")]}":
            toks.append(("close", ch, line))
        i += 1
        prev_sig = ch
    return toks


def check_balance(path: Path):
    toks = lex(path.read_text())
    stack = []  # brackets and block openers
    pair = {")": "(", "]": "[", "}": "{"}
    prev = None
    for kind, text, line in toks:
        if kind == "call":
            kind = "w"
        depth_br = sum(1 for s in stack if s[0] in "([{")
        if kind == "open":
            stack.append((text, line))
        elif kind == "close":
            assert stack and stack[-1][0] == pair[text], f"{path.name}:{line}: unbalanced '{text}' (innermost open: {stack[-1] if stack else None})"
            stack.pop()
        elif kind == "w":
            if text == "end":
                if stack and stack[-1][0] == "[":
                    pass  # a[end]
                elif depth_br and stack[-1][0] in "([{" and not any(s[0] in OPENERS for s in stack[len(stack) - 1:]):
                    # `end` directly inside (...) with no block opened inside it: an index-like use such as f(x[end]) is handled above;
                    # anything else here is an error
                    raise AssertionError(f"{path.name}:{line}: `end` inside {stack[-1]} closes nothing")
                else:
                    assert stack and stack[-1][0] in OPENERS, f"{path.name}:{line}: `end` without an open block (innermost: {stack[-1] if stack else None})"
                    stack.pop()
            elif text in OPENERS:
                if text in ("for", "if") and stack and stack[-1][0] in "([{" :
                    pass  # generator / comprehension / filter inside brackets: no `end`
                elif text == "type" and not (prev and prev[1] in ("abstract", "primitive")):
                    pass  # `type` is only a block keyword after abstract / primitive
                elif text == "struct" and prev and prev[1] == "mutable":
                    stack.append((text, line))
                else:
                    stack.append((text, line))
        prev = (kind, text, line)
    assert not stack, f"{path.name}: unclosed at end of file: {stack[-5:]}"
    return len(toks)


@pytest.mark.parametrize("path", [GEN, SHIM], ids=lambda p: p.name)
def test_blocks_and_brackets_balance(path):
    assert check_balance(path) > 500


def test_lexer_finds_a_dropped_end(tmp_path):
    """the checker is not vacuous: the generator with one `end` removed, one bracket removed, and one `end` too many all fail"""
    src = GEN.read_text()
    for bad in (src.replace("    return out\nend\n\nfunction main()", "    return out\n\nfunction main()", 1),
                src.replace("push!(out, \"logpdf\" => logpdf(fx, y))", "push!(out, \"logpdf\" => logpdf(fx, y)", 1),
                src.replace("main()\n", "end\nmain()\n", 1) if src.rstrip().endswith("main()") else src + "\nend\n"):
        assert bad != src
        p = tmp_path / "bad.jl"
        p.write_text(bad)
        with pytest.raises(AssertionError):
            check_balance(p)


def _called_names(src: str) -> set:
    toks = lex(src)
    defined = set()
    for a, b in zip(toks, toks[1:]):
        if a[0] == "w" and a[1] == "function" and b[0] in ("w", "call"):
            defined.add(b[1])
    defined |= set(re.findall(r"^([a-z_]+)\(.*?\) = ", src, flags=re.M))  # one-line definitions: points(X) = ...
    calls = {text for kind, text, _l in toks if kind == "call"}
    return calls, defined


def test_generator_calls_only_exported_reference_names():
    src = GEN.read_text()
    calls, defined = _called_names(src)
    base = {  # Base / LinearAlgebra / keywords-as-calls used by the generator
        "open", "String", "read", "ltoh", "htol", "ntuple", "Int", "Array", "read!", "error", "write", "UInt32", "Int64", "Float64", "sizeof", "ndims",
        "size", "isnan", "vec", "Vector", "Matrix", "min", "fill", "rethrow", "joinpath", "mkpath", "sort", "filter", "endswith", "readdir", "println",
        "first", "pkgversion", "push!", "Dict", "length", "Pair", "sin"}
    local_callables = {"f", "p1", "ft", "g"}  # GP objects called as f(x, Σ) / p1(x, Σ): src/finite_gp_projection.jl:32; g: the closure handed to central(g)
    unknown = sorted(c for c in calls if c not in defined | base | EXPORTS | KERNELFUNCTIONS | local_callables)
    assert not unknown, f"make_golden.jl calls names that are neither defined there, nor Base, nor exported by AbstractGPs / KernelFunctions: {unknown}"
    used = {c for c in calls if c in EXPORTS}
    assert {"GP", "VFE", "DTC", "logpdf", "posterior", "mean_and_var", "cov", "elbo", "approx_log_evidence", "update_posterior", "RowVecs"} <= used


def test_generator_reads_only_fields_the_reference_cache_has():
    src = GEN.read_text()
    exact = set(re.findall(r"\b(?:post|p2)\.data\.([^\s.,()\[\]]+)", src))
    sparse = set(re.findall(r"\b(?:ap|a2|b2)\.data\.([^\s.,()\[\]]+)", src))
    assert exact and exact <= EXACT_FIELDS, exact
    assert sparse and sparse <= VFE_FIELDS, sparse
    every = set(re.findall(r"\.data\.([^\s.,()\[\]]+)", src))
    assert every == exact | sparse, "a `.data.<field>` read on a variable this test does not classify"


@pytest.mark.skipif(not REF.exists(), reason="the reference tree is not on this machine (GPU box): the lists above are checked where it is")
def test_the_lists_above_are_the_reference_s():
    txt = (REF / "AbstractGPs.jl").read_text()
    blocks = re.findall(r"^export (.*?)(?=^\S|\Z)", txt, flags=re.S | re.M)
    names = set()
    for b in blocks:
        names |= {t.strip() for t in b.replace("\n", " ").split(",") if t.strip()}
    assert names == EXPORTS, (sorted(names - EXPORTS), sorted(EXPORTS - names))
    assert "@reexport using KernelFunctions" in txt
    exact = re.search(r"PosteriorGP\(fx\.f, \((.*?)\)\)", (REF / "exact_gpr_posterior.jl").read_text())
    assert exact and {kv.split("=")[0].strip() for kv in exact.group(1).split(",")} == EXACT_FIELDS
    sp = (REF / "sparse_approximations.jl").read_text()
    m = re.search(r"cache = \((.*?)\)\n", sp)
    assert m and {kv.split("=")[0].strip() for kv in m.group(1).split(",")} == VFE_FIELDS
