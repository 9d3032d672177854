"""The Julia extension (integration/HPCLinearAlgebraROCmExt.jl) cannot be executed here (no Julia in the
image), so its `@ccall`s are checked statically against the C ABI: every `@ccall LIB.name(arg::T, ...)::R`
must name a function declared in include/hpcla_rocm.h with the same arity, and every argument's Julia
type must be the C type of that parameter (pointer / int / int64_t / uint64_t / double / float); likewise the
return type.  The ctypes table of linearalgebrampi.jl_amd/_capi.py is checked against the header the same
way, so the three descriptions of the boundary cannot drift apart unnoticed."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _c_class(decl):
    d = decl.strip()
    if d in ("", "void"):
        return None
    if "*" in d:
        return "ptr"
    if re.search(r"\buint64_t\b", d):
        return "u64"
    if re.search(r"\bint64_t\b", d):
        return "i64"
    if re.search(r"\bdouble\b", d):
        return "f64"
    if re.search(r"\bfloat\b", d):
        return "f32"
    if re.search(r"\bint\b", d):
        return "i32"
    raise AssertionError(f"unclassified C parameter: {d!r}")


def header_prototypes():
    text = _strip_c_comments(open(os.path.join(ROOT, "include", "hpcla_rocm.h")).read())
    protos = {}
    for m in re.finditer(r"\b(const\s+char\s*\*|int64_t|int)\s*(hpcla_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = [c for c in (_c_class(a) for a in args.split(",")) if c is not None]
        protos[name] = ("cstr" if "char" in ret else ("i64" if "int64_t" in ret else "i32"), params)
    return protos


_JULIA = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "UInt64": "u64", "Cdouble": "f64", "Float64": "f64", "Cfloat": "f32", "Float32": "f32",
          "Cstring": "cstr"}


def _julia_class(t):
    t = t.strip()
    if t.startswith("Ptr{") or t == "Ptr":
        return "ptr"
    if t in _JULIA:
        return _JULIA[t]
    raise AssertionError(f"unclassified Julia type: {t!r}")


def _split_top_level(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def julia_ccalls():
    """(name, [arg classes], return class, line number) of every @ccall in the extension."""
    text = open(os.path.join(ROOT, "integration", "HPCLinearAlgebraROCmExt.jl")).read()
    calls = []
    for m in re.finditer(r"@ccall\s*\(?\s*LIB\.(hpcla_\w+)\(", text):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(text[j], 0)
            j += 1
        args = text[i:j - 1]
        ret = re.match(r"\s*::\s*(\w+)", text[j:])
        assert ret, f"@ccall {m.group(1)} without a return type"
        classes = []
        for a in _split_top_level(args):
            assert "::" in a, f"@ccall {m.group(1)}: argument without a type annotation: {a.strip()!r}"
            classes.append(_julia_class(a.rsplit("::", 1)[1]))
        calls.append((m.group(1), classes, _julia_class(ret.group(1)), text.count("\n", 0, m.start()) + 1))
    return calls


def test_header_parses_and_covers_the_exported_symbols():
    import hpcla_amd as hp
    protos = header_prototypes()
    assert len(protos) >= 80
    missing = [n for n in hp._capi.EXPORTED_SYMBOLS if n not in protos]
    assert not missing, f"bound in _capi.py but not declared in the header: {missing}"


def test_every_julia_ccall_matches_the_header():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 30, "the extension binds the hot path with far more than 30 @ccalls"
    problems = []
    for name, classes, ret, line in calls:
        if name not in protos:
            problems.append(f"line {line}: {name} is not declared in include/hpcla_rocm.h")
            continue
        want_ret, want = protos[name]
        if ret != want_ret:
            problems.append(f"line {line}: {name} returns {want_ret} in C, {ret} in Julia")
        if len(classes) != len(want):
            problems.append(f"line {line}: {name} takes {len(want)} arguments in C, {len(classes)} in Julia")
            continue
        for k, (a, b) in enumerate(zip(classes, want)):
            if a != b:
                problems.append(f"line {line}: {name} argument {k + 1} is {b} in C, {a} in Julia")
    assert not problems, "\n".join(problems)


def test_ctypes_table_matches_the_header():
    import ctypes
    import hpcla_amd as hp
    protos = header_prototypes()
    cls = {ctypes.c_void_p: "ptr", ctypes.c_int: "i32", ctypes.c_int64: "i64", ctypes.c_uint64: "u64",
           ctypes.c_double: "f64", ctypes.c_float: "f32"}
    problems = []
    for name, argtypes in hp._capi._SIGNATURES.items():
        want = protos[name][1]
        got = [cls[t] for t in argtypes]
        if got != want:
            problems.append(f"{name}: header {want} vs _capi.py {got}")
    assert not problems, "\n".join(problems)


def test_classifier_call_uses_one_index_base_for_both_arrays():
    """Regression for the round-1 defect: hpcla_classify_blocks_* was given the 1-based rowptr_target, the
    0-based split columns and index_base = 1 (first ghost column counted as owned)."""
    text = open(os.path.join(ROOT, "integration", "HPCLinearAlgebraROCmExt.jl")).read()
    for m in re.finditer(r"hpcla_classify_blocks_i(?:32|64)\(([^;]*?)::Cint\), \"hpcla_classify", text, flags=re.S):
        args = _split_top_level(m.group(1))
        assert "_ptr(rp0)" in args[0] and "rowptr_target" not in args[0], args[0]
        assert args[3].strip().startswith("0::Cint"), args[3]


def test_integration_md_export_tiers_add_up():
    """INTEGRATION.md section 5 sorts the header's exports by need; its three counts must be the header's: (A) the minimal
    drop-in listed there, (B) the rest of what the extension file binds, (C) everything the extension does not bind."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## 5. Which of the"):]
    total = int(re.search(r"## 5\. Which of the (\d+) exports", sec).group(1))
    n_a = int(re.search(r"(\d+) symbols with their", sec).group(1))
    n_b = int(re.search(r"beyond that \((\d+) more symbols\)", sec).group(1))
    n_c = int(re.search(r"Not bound by the extension \((\d+) symbols\)", sec).group(1))
    header = open(os.path.join(root, "include", "hpcla_rocm.h")).read()
    exported = set(re.findall(r"\b(hpcla_[a-z0-9_]+)\s*\(", header))
    ext = open(os.path.join(root, "integration", "HPCLinearAlgebraROCmExt.jl")).read()
    used = set(re.findall(r"LIB\.(hpcla_[a-z0-9_]+)", ext))
    assert used <= exported
    assert total == len(exported) == n_a + n_b + n_c, (total, len(exported), n_a, n_b, n_c)
    assert n_a + n_b == len(used) and n_c == len(exported - used)
