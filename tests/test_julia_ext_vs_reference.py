"""Static cross-check of integration/HPCLinearAlgebraROCmExt.jl against the REFERENCE's Julia sources.

The extension cannot be executed here (no Julia in the image); tests/test_julia_binding_signatures.py checks its
`@ccall`s against the C header.  This test checks the OTHER side of the file -- everything it takes from the parent
package: every name in its `using HPCLinearAlgebra: ...` list, every `HPCLinearAlgebra.<name>` it extends or calls
(with the number of positional arguments of the methods it adds), and every field it reads from the parent's structs
(`plan.<field>` of VectorPlan / VectorRepartitionPlan / AdditionPlan, `A.<field>` of HPCSparseMatrix, `x.<field>` of
HPCVector, `M.<field>` of HPCMatrix, `backend.<field>`, `comm.<field>`) must exist in /root/reference/src/*.jl.

The reference tree exists only in the build container (never on the GPU box): the test skips when it is absent.
Nothing is copied from it -- it is read as text, names and arities are extracted with regular expressions.
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
EXT = os.path.join(ROOT, "integration", "HPCLinearAlgebraROCmExt.jl")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference sources not present")

# names the PARENT PATCH of INTEGRATION.md adds (they cannot exist in the unpatched reference): checked against
# INTEGRATION.md instead
PARENT_PATCH_NAMES = {"DeviceROCm", "backend_rocm_serial", "backend_rocm_mpi"}
# names the parent does not DEFINE but brings into its namespace with `using Blake3Hash` (Project.toml dependency) and calls
# itself (src/sparse.jl:103-119): reachable as HPCLinearAlgebra.<name>; accepted only while the reference still imports the
# package and calls each of them (checked below)
THIRD_PARTY_VIA_PARENT = {"Blake3Ctx": r"Blake3Ctx\(\)", "update!": r"update!\(ctx", "digest": r"digest\(ctx"}


def _third_party_names(text):
    if not re.search(r"^\s*using\s+Blake3Hash\b", text, flags=re.M):
        return set()
    return {n for n, pat in THIRD_PARTY_VIA_PARENT.items() if re.search(pat, text)}


def _strip_comments_and_strings(text):
    text = re.sub(r'"""(?:.|\n)*?"""', lambda m: '""' + "\n" * m.group(0).count("\n"), text)   # docstrings (line numbers kept)
    text = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', text)      # string literals
    return "\n".join(line.split("#", 1)[0] for line in text.splitlines())


def _ref_text():
    out = []
    for fn in sorted(os.listdir(os.path.join(REF, "src"))):
        if fn.endswith(".jl"):
            out.append(open(os.path.join(REF, "src", fn)).read())
    return _strip_comments_and_strings("\n".join(out))


def _match_paren(text, i):
    """index just behind the parenthesis that closes the one at text[i]"""
    depth = 0
    for j in range(i, len(text)):
        if text[j] in "([{":
            depth += 1
        elif text[j] in ")]}":
            depth -= 1
            if depth == 0:
                return j + 1
    raise AssertionError("unbalanced parentheses")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch in ",;" and depth == 0:
            out.append((cur, ch))
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append((cur, ""))
    return out


def _arity(argtext):
    """(minimum, maximum) number of POSITIONAL arguments of a Julia signature's argument text (max None = varargs)."""
    lo = hi = 0
    kw = False
    for a, sep in _split_top(argtext):
        a = a.strip()
        if not a:
            if sep == ";":
                kw = True
            continue
        if not kw:
            if a.endswith("..."):
                hi = None
            else:
                if hi is not None:
                    hi += 1
                if "=" not in re.sub(r"\{[^{}]*\}", "", a).replace("==", ""):
                    lo += 1
        if sep == ";":
            kw = True
    return lo, hi


def reference_structs(text):
    """struct name -> list of field names"""
    out = {}
    for m in re.finditer(r"^\s*(?:mutable\s+)?struct\s+(\w+)[^\n]*\bend\s*$", text, flags=re.M):
        out[m.group(1)] = []                              # one-line definitions: `struct CommSerial <: AbstractComm end`
    for m in re.finditer(r"^\s*(?:mutable\s+)?struct\s+(\w+)(?:(?!\bend\s*$)[^\n])*\n(.*?)^end\b", text, flags=re.S | re.M):
        fields = []
        depth = 0
        for line in m.group(2).splitlines():
            s = line.strip()
            if re.match(r"(function\b|for\b|if\b|while\b|let\b|begin\b)", s):
                depth += 1
            if depth == 0:
                f = re.match(r"(\w+)\s*(::|$)", s)
                if f and f.group(1) not in ("end", "new"):
                    fields.append(f.group(1))
            if s == "end" or s.startswith("end "):
                depth = max(depth - 1, 0)
        out[m.group(1)] = fields
    return out


def reference_functions(text):
    """function name -> list of (min, max) positional arities over all its methods ([] for bare `function f end`)"""
    out = {}
    for m in re.finditer(r"^\s*function\s+(?:\w+\.)*([\w!]+)\s*(\(|end\b|\{)", text, flags=re.M):
        name = m.group(1)
        out.setdefault(name, [])
        if m.group(2) == "(":
            i = m.end() - 1
            out[name].append(_arity(text[i + 1:_match_paren(text, i) - 1]))
        elif m.group(2) == "{":
            i = text.index("(", m.end())
            out[name].append(_arity(text[i + 1:_match_paren(text, i) - 1]))
    for m in re.finditer(r"^(?:\w+\.)*([\w!]+)(?:\{[^\n]*?\})?\(", text, flags=re.M):          # one-line definitions
        i = m.end() - 1
        j = _match_paren(text, i)
        if re.match(r"\s*(?:where\s+[^=\n]+?)?\s*=(?!=)", text[j:j + 200]):
            out.setdefault(m.group(1), []).append(_arity(text[i + 1:j - 1]))
    return out


def reference_names(text):
    names = set(reference_structs(text)) | set(reference_functions(text))
    names |= set(re.findall(r"^\s*abstract\s+type\s+(\w+)", text, flags=re.M))
    names |= set(re.findall(r"^\s*const\s+(\w+)", text, flags=re.M))
    return names


def _ext_text():
    return _strip_comments_and_strings(open(EXT).read())


def test_reference_parser_sees_the_known_landmarks():
    """The regex parser is only trusted if it finds what SURVEY.md section 8 cites by line."""
    text = _ref_text()
    structs = reference_structs(text)
    assert {"send_rank_ids", "send_indices", "recv_rank_ids", "recv_perm", "local_src_indices", "local_dst_indices",
            "gathered", "gathered_cpu", "result_partition_hash", "result_partition"} <= set(structs["VectorPlan"])
    assert {"row_partition", "col_partition", "col_indices", "rowptr", "colval", "nzval", "nrows_local",
            "ncols_compressed", "rowptr_target", "colval_target", "backend", "cached_transpose"} <= set(structs["HPCSparseMatrix"])
    assert structs["HPCVector"][:4] == ["structural_hash", "partition", "v", "backend"]
    funcs = reference_functions(text)
    assert (2, 2) in funcs["execute_plan!"] and (2, 2) in funcs["_convert_array"] and (2, 2) in funcs["get_vector_plan"]
    assert "backend_cuda_mpi" in funcs                       # the stub pattern the parent patch copies


def test_imported_names_exist_in_the_reference():
    ext = _ext_text()
    m = re.search(r"using\s+HPCLinearAlgebra\s*:\s*((?:[\w!]+\s*,\s*)*[\w!]+)", ext)
    assert m, "no `using HPCLinearAlgebra: ...` list"
    imported = [n.strip() for n in m.group(1).split(",")]
    assert len(imported) >= 12
    names = reference_names(_ref_text())
    missing = [n for n in imported if n not in names and n not in PARENT_PATCH_NAMES]
    assert not missing, f"imported from HPCLinearAlgebra but not defined in {REF}/src: {missing}"
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in PARENT_PATCH_NAMES & set(imported):
        assert n in integ, f"{n} must be added by the parent patch shown in INTEGRATION.md"


def test_qualified_names_exist_and_extended_methods_have_a_reference_arity():
    ext = _ext_text()
    text = _ref_text()
    names = reference_names(text)
    funcs = reference_functions(text)
    problems = []
    names = names | _third_party_names(text)
    assert {"Blake3Ctx", "update!", "digest"} <= names, "the reference no longer hashes with Blake3Ctx / update! / digest"
    for m in re.finditer(r"HPCLinearAlgebra\.([\w!]+)", ext):
        n = m.group(1)
        if n not in names and n not in PARENT_PATCH_NAMES:
            problems.append(f"line {ext.count(chr(10), 0, m.start()) + 1}: HPCLinearAlgebra.{n} is not defined in the reference")
    # method extensions: `function HPCLinearAlgebra.f(args)` and `HPCLinearAlgebra.f(args) = ...`
    extended = []
    for m in re.finditer(r"^(?:function\s+)?HPCLinearAlgebra\.([\w!]+)\(", ext, flags=re.M):
        i = m.end() - 1
        j = _match_paren(ext, i)
        is_def = m.group(0).startswith("function") or re.match(r"\s*(?:where\s+[^=\n]+?)?\s*=(?!=)", ext[j:j + 200])
        if is_def:
            extended.append((m.group(1), _arity(ext[i + 1:j - 1]), ext.count("\n", 0, m.start()) + 1))
    assert len(extended) >= 12, extended
    for name, (lo, hi), line in extended:
        if name in PARENT_PATCH_NAMES:
            continue                                          # the stubs the parent patch declares (`function f end`)
        ref = list(funcs.get(name, []))
        if not ref and name in funcs:
            # a bare stub in the parent (`function _array_to_device end`): the contract is the parent's CALL sites
            for c in re.finditer(r"(?<![\w.!])%s\(" % re.escape(name), text):
                if text[max(c.start() - 9, 0):c.start()].strip().endswith("function"):
                    continue
                k = c.end() - 1
                n_args = len([a for a, _ in _split_top(text[k + 1:_match_paren(text, k) - 1]) if a.strip()])
                ref.append((n_args, n_args))
            if not ref:
                continue                                      # declared for extensions, never called by the parent: any arity
        if not ref:
            problems.append(f"line {line}: {name} has neither a method nor a stub in the reference")
            continue
        ok = any((rhi is None or lo <= rhi) and (hi is None or hi >= rlo) for rlo, rhi in ref)
        if not ok:
            problems.append(f"line {line}: {name} extended with {lo}..{hi} positional arguments, reference methods take {sorted(set(ref), key=str)}")
    assert not problems, "\n".join(problems)


# variable -> struct, from the type annotations of each definition's signature (plus the two idioms
# `plan = get_vector_plan(...)` and `d = _device_plan(...)`)
_TYPE_OF_ANNOTATION = [
    (r"HPCSparseMatrix\b", "HPCSparseMatrix"), (r"HPCVector\b", "HPCVector"), (r"HPCMatrix\b", "HPCMatrix"),
    (r"(?<![A-Za-z])VectorRepartitionPlan\b", "VectorRepartitionPlan"), (r"(?<![A-Za-z])AdditionPlan\b", "AdditionPlan"),
    (r"(?<![A-Za-z])VectorPlan\b", "VectorPlan"),             # (not ROCVectorPlan: that struct is the extension's own)
    (r"(?<![A-Za-z])MatrixPlan\b", "MatrixPlan"),
    (r"HPCBackend\b|ROCBackend\b", "HPCBackend"), (r"CommMPI\b", "CommMPI"),
]


def _chunks(ext):
    """(signature text, body text, first line) of every top-level definition of the extension"""
    starts = [m.start() for m in re.finditer(r"^(?:function\s|[\w.:!*]+(?:\{[^\n]*?\})?\([^\n]*\)\s*(?:where[^\n=]*)?=(?!=))", ext, flags=re.M)]
    starts.append(len(ext))
    for a, b in zip(starts, starts[1:]):
        chunk = ext[a:b]
        i = chunk.index("(")
        j = _match_paren(chunk, i)
        yield chunk[i + 1:j - 1], chunk[j:], ext.count("\n", 0, a) + 1


def test_every_field_read_from_a_parent_struct_exists_there():
    ext = _ext_text()
    structs = reference_structs(_ref_text())
    for need in ("VectorPlan", "VectorRepartitionPlan", "AdditionPlan", "MatrixPlan", "HPCSparseMatrix", "HPCVector", "HPCMatrix",
                 "HPCBackend", "CommMPI"):
        assert need in structs and structs[need], f"struct {need} not found in the reference"
    problems, checked = [], 0
    for sig, body, line in _chunks(ext):
        var_type = {}
        for a, _sep in _split_top(sig):
            m = re.match(r"\s*(\w+)\s*::\s*(.+)", a.strip(), flags=re.S)
            if not m:
                continue
            for pat, st in _TYPE_OF_ANNOTATION:
                if re.search(pat, m.group(2)):
                    var_type[m.group(1)] = st
                    break
        if re.search(r"\bplan\s*=\s*get_vector_plan\(", body) or ("plan" in [a.strip() for a, _ in _split_top(sig)]
                                                                  and re.search(r"plan\.(send_rank_ids|recv_perm)", body)):
            var_type.setdefault("plan", "VectorPlan")
        # sparse A * B: `plan = HPCLinearAlgebra.MatrixPlan(A, Bm)` and the helpers that take it (they read plan.AT)
        if re.search(r"\bplan\s*=\s*HPCLinearAlgebra\.MatrixPlan\(", body) or ("plan" in [a.strip() for a, _ in _split_top(sig)]
                                                                               and re.search(r"plan\.AT\b", body)):
            var_type["plan"] = "MatrixPlan"
        for var, st in var_type.items():
            for m in re.finditer(r"(?<![\w.])%s((?:\.\w+)+)" % re.escape(var), body):
                chain = m.group(1).strip(".").split(".")
                cur = st
                for f in chain:
                    if cur is None:
                        break
                    checked += 1
                    if f not in structs[cur]:
                        problems.append(f"definition at line {line}: {var}.{'.'.join(chain)} -- struct {cur} of the reference has no field `{f}` "
                                        f"(fields: {structs[cur]})")
                        break
                    cur = {("HPCSparseMatrix", "backend"): "HPCBackend", ("HPCVector", "backend"): "HPCBackend",
                           ("HPCMatrix", "backend"): "HPCBackend"}.get((cur, f))
                    if cur == "HPCBackend" and chain[chain.index(f) + 1:chain.index(f) + 2] == ["comm"] and chain[-1] == "comm" and len(chain) > chain.index(f) + 2:
                        # A.backend.comm.comm: the MPI communicator inside CommMPI (guarded by `isa CommMPI` in the file)
                        if "comm" not in structs["CommMPI"]:
                            problems.append(f"line {line}: CommMPI has no field comm")
                        cur = None
    assert checked >= 60, f"only {checked} field reads were checked: the parser lost track of the file"
    assert not problems, "\n".join(problems)


def _julia_code_tokens(text):
    """The extension's text with comments and string literals blanked out (Julia is not available to parse it)."""
    out, i, n = [], 0, len(text)
    while i < n:
        if text.startswith('"""', i):
            j = text.index('"""', i + 3)
            out.append(" " * (j + 3 - i) if "\n" not in text[i:j + 3] else "\n" * text[i:j + 3].count("\n"))
            i = j + 3
        elif text[i] == '"':
            j = i + 1
            while text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append('""')
            i = j + 1
        elif text[i] == "#":
            j = text.find("\n", i)
            j = n if j < 0 else j
            i = j
        else:
            out.append(text[i])
            i += 1
    return "".join(out)


def _julia_balance_problem(text):
    """None when every block keyword at statement level is closed by an `end` and (), [], {} nest properly; else what is wrong."""
    import re
    stack, blocks, opens = [], [], 0              # open brackets / open blocks as (token, line)
    pairs = {")": "(", "]": "[", "}": "{"}
    line = 1
    for m in re.finditer(r"\n|[()\[\]{}]|(?<![\w.:@!])(?:mutable\s+struct|function|if|for|while|let|do|struct|module|begin|try|quote|macro|end)(?![\w!?(])", text):
        tok = m.group(0)
        if tok == "\n":
            line += 1
        elif tok in "([{":
            stack.append((tok, line))
        elif tok in ")]}":
            if not stack or stack[-1][0] != pairs[tok]:
                return f"line {line}: unbalanced {tok!r}"
            stack.pop()
        elif any(b[0] == "[" for b in stack):
            continue                              # comprehension `for` / `if`, indexing `end`
        elif tok == "end":
            if not blocks:
                return f"line {line}: `end` without an open block"
            blocks.pop()
        elif stack and tok in ("for", "if"):
            continue                              # generator / filter inside parentheses: no `end`
        else:
            blocks.append((tok, line))
            opens += 1
    if stack:
        return f"unclosed bracket opened at line {stack[-1][1]}"
    if blocks:
        return f"unclosed block: `{blocks[-1][0]}` at line {blocks[-1][1]}"
    return None if opens > 50 else "the scanner saw almost no blocks"


def test_extension_blocks_and_brackets_balance():
    """No Julia here to parse the extension, so at least its block structure is checked: every function / if / for / while /
    let / do / struct / module / begin / try at statement level opens a block that an `end` closes (keywords inside square
    brackets -- comprehensions, `a[end]` -- and generator `for` / `if` inside parentheses do not count), and (), [], {} nest
    properly.  The scanner is itself checked on mutations of the file: a dropped `end`, a stray `end`, a dropped bracket."""
    text = _julia_code_tokens(open(os.path.join(ROOT, "integration", "HPCLinearAlgebraROCmExt.jl")).read())
    assert _julia_balance_problem(text) is None, _julia_balance_problem(text)
    k = text.rindex("\nend\n", 0, text.rindex("\nend"))           # the `end` of the last function
    assert _julia_balance_problem(text[:k] + text[k + 4:]) is not None
    assert _julia_balance_problem(text[:k] + "\nend" + text[k:]) is not None
    j = text.index("(", text.index("function _check"))
    assert _julia_balance_problem(text[:j] + text[j + 1:]) is not None


def test_every_private_helper_the_extension_calls_is_defined_in_it():
    """`_name(...)` calls that are not qualified with the parent module must be helpers of the extension file itself."""
    import re
    text = _julia_code_tokens(open(os.path.join(ROOT, "integration", "HPCLinearAlgebraROCmExt.jl")).read())
    defined = set(re.findall(r"\bfunction\s+(?:[\w.]+\.)?(_\w+!?)", text))
    defined |= set(re.findall(r"^(?:[\w.]+\.)?(_\w+!?)\([^=\n]*\)(?:\s*where\s*\{[^}]*\})?\s*=", text, flags=re.M))
    defined |= set(re.findall(r"^const\s+(_\w+)", text, flags=re.M))
    used = set(re.findall(r"(?<![\w.])(_\w+!?)\(", text))
    assert len(used) > 20
    assert not (used - defined), f"called but not defined in the extension: {sorted(used - defined)}"


def _function_body(text, header_regex):
    """text of the top-level `function ...` whose header matches, up to its closing `end` at column 0"""
    m = re.search(header_regex, text, flags=re.M)
    assert m, header_regex
    end = re.search(r"^end\b", text[m.start():], flags=re.M)
    return text[m.start():m.start() + end.end()]


def test_vector_plan_constructor_override_builds_the_reference_struct_with_the_reference_collectives():
    """Round 6: the extension overrides `VectorPlan(A, x)` for DeviceROCm backends (binary searches instead of one push! per
    compressed column).  Checked against the parent's constructor (src/sparse.jl:1875-1984) as text: the 18 arguments of the
    final `VectorPlan{T,Ti,AV}(...)` call are the struct's fields in declaration order (the twelve lists / buffers by name,
    six `nothing`s for the lazily filled fields), and the collectives appear in the parent's order with the parent's tag --
    Alltoall of the counts, isend / irecv of the requested indices with tag 20, waitall on the receives, waitall on the
    sends -- so that ranks running either method meet in every collective."""
    ext = _ext_text()
    ref = _ref_text()
    fields = reference_structs(ref)["VectorPlan"]
    assert len(fields) == 18, fields
    body = _function_body(ext, r"^function HPCLinearAlgebra\.VectorPlan\(A::HPCSparseMatrix\{T,Ti,B\}, x::HPCVector\{T,B\}\) where \{T,Ti,B<:ROCBackend\}")
    call = body[body.rindex("VectorPlan{T,Ti,AV}("):]
    i = call.index("(")
    args = [a.strip() for a, _ in _split_top(call[i + 1:_match_paren(call, i) - 1]) if a.strip()]
    assert len(args) == 18, args
    assert args[:12] == fields[:12], (args[:12], fields[:12])
    assert args[12:] == ["nothing"] * 6, args[12:]
    ref_body = _function_body(ref, r"^function VectorPlan\(A::HPCSparseMatrix\{T,Ti,B\}, x::HPCVector\{T,Bx\}\)")

    def collectives(text):
        out = []
        for m in re.finditer(r"comm_(alltoall|isend|irecv!|waitall)\(([^\n]*)", text):
            kind, rest = m.group(1), m.group(2)
            if kind in ("isend", "irecv!"):
                tag = re.search(r",\s*(\d+)\)", rest)
                out.append((kind, tag.group(1) if tag else None))
            elif kind == "waitall":
                out.append((kind, "recv" if "recv" in rest else "send"))
            else:
                out.append((kind, None))
        return out
    mine, theirs = collectives(body), collectives(ref_body)
    assert theirs == [("alltoall", None), ("isend", "20"), ("irecv!", "20"), ("waitall", "recv"), ("waitall", "send")], theirs
    assert mine == theirs, (mine, theirs)
    # the lazily sized buffers: only for Float64, where this file's execute_plan! / mul! never read gathered_cpu
    assert re.search(r"lazy = T === Float64", body) and "similar(x.v, lazy ? 0 : n_gathered)" in body


def test_spmm_result_passes_the_fields_of_hpcmatrix_in_order():
    """`_spmm_result` builds HPCMatrix{T,B} with the inner constructor (src/dense.jl:59-69): five arguments in field order --
    hash (lazy: nothing), row partition, column partition, the local block, the backend."""
    ref = _ref_text()
    fields = reference_structs(ref)["HPCMatrix"]
    assert fields == ["structural_hash", "row_partition", "col_partition", "A", "backend"], fields
    ext = _ext_text()
    m = re.search(r"_spmm_result\(A::HPCSparseMatrix\{T,Ti,B\}, C::ROCMatrix\{T\}, k::Int\) where \{T,Ti,B\} =\s*HPCMatrix\{T,B\}\(", ext)
    assert m
    i = m.end() - 1
    args = [a.strip() for a, _ in _split_top(ext[i + 1:_match_paren(ext, i) - 1])]
    assert len(args) == 5 and args[0] == "nothing" and args[1] == "copy(A.row_partition)" and args[3] == "C" and args[4] == "A.backend", args
    assert args[2].startswith("HPCLinearAlgebra.uniform_partition(k,"), args[2]


def _unbound_type_variables(text):
    """[(definition, line, missing names)] -- see test_type_variables_are_bound"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_julia_no_host_staging as audit
    names = ("T", "Ti", "Tk", "B", "Timat", "AV")
    problems = []
    for name, first, last, body in audit.parse_functions(text):
        code = [audit._code(l) for l in body]
        sig = re.sub(r"^function\s+", "", " ".join(code[:10]))
        i = sig.index("(")
        j = audit._match_close(sig, i)
        bound = set()
        m = re.match(r"\s*where\s*", sig[j:])
        if m:
            k = j + m.end()
            if sig[k] == "{":
                depth, e = 0, k
                while True:
                    depth += {"{": 1, "}": -1}.get(sig[e], 0)
                    e += 1
                    if depth == 0:
                        break
                inner, depth, cur, parts = sig[k + 1:e - 1], 0, "", []
                for ch in inner:
                    depth += {"{": 1, "}": -1}.get(ch, 0)
                    if ch == "," and depth == 0:
                        parts.append(cur)
                        cur = ""
                    else:
                        cur += ch
                parts.append(cur)
                bound = {re.match(r"\s*(\w+)", q).group(1) for q in parts if q.strip()}
            else:
                bound = {re.match(r"(\w+)", sig[k:]).group(1)}
        whole = "\n".join(code)
        assigned = set(re.findall(r"(?<![\w.])(%s)\s*=(?!=)" % "|".join(names), whole))
        argnames = set(re.findall(r"[(,]\s*(\w+)\s*(?:::|[,)=])", sig[i:j]))
        used = set(re.findall(r"(?<![\w.:\"])(%s)(?![\w(\"])" % "|".join(names), whole))
        missing = used - bound - assigned - argnames
        if missing:
            problems.append((name, first, sorted(missing), sorted(bound)))
    return problems


def test_type_variables_are_bound():
    """A Julia method whose signature or body names a type variable that its `where` clause does not bind is an UndefVarError
    at the first call -- invisible to every other static check of this file.  For every top-level definition of the extension:
    the conventional type-variable names it uses (T, Ti, Tk, B, Timat, AV) must be bound by the where clause (balanced braces:
    `T<:Union{Float32,Float64}`), be the name of an argument, or be assigned in the body (`Ti = indextype_backend(B)`).
    Nested closures (`function launch(...)` inside a method, `do` blocks) see the enclosing method's variables."""
    text = open(EXT).read()
    problems = _unbound_type_variables(text)
    assert not problems, "\n".join(f"{n} (line {l}): {m} used but not bound by `where` ({b}) nor assigned" for n, l, m, b in problems)
    # the check must be able to fail: drop `Tk` from one where clause, use an unbound `Ti` in a body
    old = "d::ROCVectorPlan{Tk}, xpart::Vector{Int}, width::Int, rpb::Cint) where {T,Ti,Tk,B<:ROCBackend}"
    assert text.count(old) == 1
    got = _unbound_type_variables(text.replace(old, old.replace("{T,Ti,Tk,B<:ROCBackend}", "{T,Ti,B<:ROCBackend}")))
    assert any(n == "_spmm_halo" and "Tk" in m for n, l, m, b in got), got
    old2 = "function _classify_blocks(A, rp0::ROCVector{Tk}, colval_split::ROCVector{Tk}, n_own::Int, rpb::Cint) where {Tk}\n"
    assert text.count(old2) == 1
    got = _unbound_type_variables(text.replace(old2, old2 + "    z = zero(Ti)\n"))
    assert any(n == "_classify_blocks" and "Ti" in m for n, l, m, b in got), got

def _undefined_names(text):
    """[(definition, line, names)]: identifiers a top-level definition of the extension uses that are neither its arguments,
    type variables, names assigned anywhere in it (plain / tuple / augmented assignment, `for` variables, `do` and `->`
    arguments, nested function definitions and their arguments), module-level names of the file (functions, consts, structs,
    the `using HPCLinearAlgebra:` list, the imported modules) nor one of the Base / AMDGPU names the file is known to use."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_julia_no_host_staging as t
    code_all = "\n".join(t._code(l) for l in text.split("\n"))
    funcs = t.parse_functions(text)
    KEYWORDS=set("function end if else elseif for while in return where do let begin try catch finally const mutable struct module using import export true false nothing isa break continue local global quote macro new".split())
    # module-level names
    modnames=set(re.findall(r"^const\s+(\w+)",code_all,flags=re.M))|set(re.findall(r"^(?:mutable\s+)?struct\s+(\w+)",code_all,flags=re.M))
    modnames|={f[0].split(".")[-1].lstrip(":") for f in funcs}
    m=re.search(r"using\s+HPCLinearAlgebra\s*:\s*((?:[\w!]+\s*,\s*)*[\w!]+)",code_all)
    modnames|={n.strip() for n in m.group(1).split(",")}
    modnames|={"HPCLinearAlgebra","AMDGPU","MPI","LinearAlgebra","SparseArrays","Base","LIB","ROCBackend"}
    BASE=set("""Ptr Cvoid Cint Cdouble Cstring Cfloat UInt UInt8 UInt32 UInt64 Int Int8 Int16 Int32 Int64 Float16 Float32 Float64 Bool Any Real Number Integer
    Vector Matrix Array Dict IdDict Ref Type Union Tuple NamedTuple Symbol AbstractString AbstractVector AbstractMatrix UnitRange Nothing
    ROCArray ROCVector ROCMatrix C_NULL ENV Inf
    error get get! haskey length size zeros fill similar copy push! append! empty! filter! popfirst! sum minimum maximum extrema any all isempty zip enumerate
    reduce vcat cld iseven isnan sqrt parse max min unsafe_string pointer reshape findall unique rand sortperm cumsum cumsum! reinterpret convert
    collect eltype typemax one zero values keys first last invoke MergeSort searchsortedfirst searchsortedlast undef typeof diff
    isa @ccall @inbounds ccall string print println abs Threads""".split())
    probs=[]
    for name,first,last,body in funcs:
        code=[t._code(l) for l in body]
        whole="\n".join(code)
        # strip @ccall type annotations noise: keep
        ids=set(re.findall(r"(?<![\w.:@])([A-Za-z_]\w*!?)(?![\w!])",whole))
        # remove field accesses (after a dot) -- handled by lookbehind; remove keyword-arg names `name=` inside calls? keep
        sig=re.sub(r"^function\s+","",whole)
        i=sig.index("("); j=t._match_close(sig,i)
        args=set()
        depth=0; cur=""
        for ch in sig[i+1:j-1]:
            if ch in "({[": depth+=1
            elif ch in ")}]": depth-=1
            if ch in ",;" and depth==0:
                mm=re.match(r"\s*(\w+)",cur); 
                if mm: args.add(mm.group(1))
                cur=""
            else: cur+=ch
        mm=re.match(r"\s*(\w+)",cur)
        if mm: args.add(mm.group(1))
        wherev=set(re.findall(r"\b(\w+)\s*(?:<:|[,}])",sig[j:j+200].split("\n")[0])) if "where" in sig[j:j+200].split("\n")[0] else set()
        assigned=set(re.findall(r"(?<![\w.])(\w+)\s*(?:=(?!=)|\+=|-=|\*=|\|=|&=)",whole))
        # tuple assignments a, b = ... / (a, b) = / for (a, b) in
        for mm in re.finditer(r"(?:^|\n|;)\s*((?:\w+\s*,\s*)+\w+)\s*=(?!=)",whole):
            assigned|=set(re.findall(r"\w+",mm.group(1)))
        for mm in re.finditer(r"\(\s*((?:\w+\s*,\s*)+\w+)\s*\)\s*(?:=(?!=)|in\b)",whole):
            assigned|=set(re.findall(r"\w+",mm.group(1)))
        for mm in re.finditer(r"\bfor\s+(\w+)\s+in\b|\bfor\s+(\w+)\s*=",whole):
            assigned|={g for g in mm.groups() if g}
        for mm in re.finditer(r"\bfor\s+\(([^)]*)\)\s+in\b",whole):
            assigned|=set(re.findall(r"\w+",mm.group(1)))
        for mm in re.finditer(r"\bfor\s+((?:\w+\s*,\s*)+\w+)\s+in\b",whole):
            assigned|=set(re.findall(r"\w+",mm.group(1)))
        for mm in re.finditer(r"\bdo\s+((?:\w+\s*,?\s*)*)$",whole,flags=re.M):
            assigned|=set(re.findall(r"\w+",mm.group(1)))
        for mm in re.finditer(r"(\w+)\s*->",whole): assigned.add(mm.group(1))
        for mm in re.finditer(r"\(\s*((?:\w+\s*,\s*)*\w+)\s*\)\s*->",whole): assigned|=set(re.findall(r"\w+",mm.group(1)))
        # nested function definitions and their args
        for mm in re.finditer(r"\bfunction\s+(\w+)\(([^)]*)\)",whole):
            assigned.add(mm.group(1)); assigned|=set(re.findall(r"(\w+)\s*(?:::|,|$)",mm.group(2)))
        for mm in re.finditer(r"(?:^|\n)\s*(\w+)\(([^)]*)\)\s*=(?!=)",whole):
            assigned.add(mm.group(1)); assigned|=set(re.findall(r"(\w+)\s*(?:::|,|$)",mm.group(2)))
        # keyword arguments in calls: name=value inside parentheses -> `name` appears as assigned (acceptable)
        unknown=ids-KEYWORDS-modnames-BASE-args-wherev-assigned
        unknown={u for u in unknown if not u.startswith("hpcla_") and not re.fullmatch(r"\d\w*",u)}
        if unknown: probs.append((name,first,sorted(unknown)))
    
    return probs


def test_no_undefined_names_in_the_extension():
    """No Julia here to run the file: a misspelt local (`n_ghosts` for `n_ghost`) would be an UndefVarError at the first call.
    Every identifier of every definition must resolve (see _undefined_names); two mutations check that the scan can fail."""
    text = open(EXT).read()
    probs = _undefined_names(text)
    assert not probs, "\n".join(f"{n} (line {l}): undefined {names}" for n, l, names in probs)
    old = "        n_ghost = sum(recv_counts; init=Int64(0))\n"
    assert text.count(old) == 1
    got = _undefined_names(text.replace(old, "        n_ghosts = sum(recv_counts; init=Int64(0))\n"))
    assert any(n == "_spmm_halo" and "n_ghost" in names for n, l, names in got), got
    old2 = "    nnz = length(A.nzval); nb = _len(blocks)\n    _spmm_order!(rp0, 1)"
    assert text.count(old2) == 1
    got = _undefined_names(text.replace(old2, "    nnz = length(A.nzval); nb = _len(blocks)\n    _spmm_order!(rpo, 1)"))
    assert any(n == "_spmm_split!" and "rpo" in names for n, l, names in got), got

def _helper_call_arity_problems(text):
    """Calls of the file's own helpers (`_name(...)`, `rocm_...`, `clear_rocm_plan_cache!`) with a number of positional
    arguments that none of the helper's methods accepts."""
    code = _julia_code_tokens(text)
    defs = {}
    for m in re.finditer(r"^(?:function\s+)?((?:_\w+|rocm_\w+|clear_rocm_plan_cache)!?)\(", code, flags=re.M):
        i = m.end() - 1
        j = _match_paren(code, i)
        is_def = m.group(0).startswith("function") or re.match(r"\s*(?:where\s*(?:\{.*?\}|\w+)\s*)?=(?!=)", code[j:j + 200])
        if is_def:
            defs.setdefault(m.group(1), []).append(_arity(code[i + 1:j - 1]))
    # nested closures (`function launch(ghost, blocks, nblocks)` inside a method)
    for m in re.finditer(r"^\s+function\s+(\w+!?)\(", code, flags=re.M):
        i = m.end() - 1
        defs.setdefault(m.group(1), []).append(_arity(code[i + 1:_match_paren(code, i) - 1]))
    problems = []
    for name, arities in defs.items():
        for c in re.finditer(r"(?<![\w.!])%s\(" % re.escape(name), code):
            line_start = code.rfind("\n", 0, c.start()) + 1
            if code[line_start:c.start()].strip().startswith("function") or code[line_start:c.start()].strip() == "":
                k = c.end() - 1
                after = code[_match_paren(code, k):_match_paren(code, k) + 120]
                if code[line_start:c.start()].strip().startswith("function") or re.match(r"\s*(?:where\s*(?:\{.*?\}|\w+)\s*)?=(?!=)", after):
                    continue                                  # a definition, not a call
            k = c.end() - 1
            inner = code[k + 1:_match_paren(code, k) - 1]
            pos = 0
            for a, sep in _split_top(inner):
                if sep == ";" and a.strip():
                    pos += 1
                    break
                if not a.strip():
                    if sep == ";":
                        break
                    continue
                if re.match(r"\s*\w+\s*=(?!=)", a):            # keyword argument written after a comma
                    continue
                pos += 1
                if sep == ";":
                    break
            if not any(lo <= pos and (hi is None or pos <= hi) for lo, hi in arities):
                problems.append(f"line {code.count(chr(10), 0, c.start()) + 1}: {name} called with {pos} positional arguments, methods take {sorted(set(arities), key=str)}")
    return problems, defs


def test_helper_calls_match_a_method_arity():
    """Every call of one of the file's own helpers passes a number of positional arguments some method of it accepts (round 6
    rewrote several helper signatures -- `_spmm_split!`, `_spmm_halo`, `_device_plan` -- in a file that cannot be run)."""
    text = open(EXT).read()
    problems, defs = _helper_call_arity_problems(text)
    assert len(defs) >= 40, len(defs)
    assert not problems, "\n".join(problems)
    old = "_spmm_split!(C, A, rp0, colval_split, d.n_own, Brow, kp, ghost, k, interior, true)"
    assert text.count(old) == 1
    got, _ = _helper_call_arity_problems(text.replace(old, "_spmm_split!(C, A, rp0, colval_split, d.n_own, Brow, ghost, k, interior, true)"))
    assert got and "_spmm_split!" in got[0], got



# ---- tuples built against the tuples taken apart ----------------------------------------------------------------------------------
def _count_top(expr):
    """number of top-level comma-separated items of `expr` (one enclosing pair of parentheses removed when it spans all of it)"""
    e = expr.strip()
    if e.startswith("(") and _match_paren(e, 0) == len(e):
        e = e[1:-1]
    return len([a for a, _ in _split_top(e) if a.strip()])


def _joined_statement(lines, idx):
    """the statement that starts on lines[idx], continuation lines joined while its brackets are open"""
    s = lines[idx].strip()
    depth = sum(s.count(c) for c in "([{") - sum(s.count(c) for c in ")]}")
    j = idx
    while depth > 0 and j + 1 < len(lines):
        j += 1
        t = lines[j].strip()
        depth += sum(t.count(c) for c in "([{") - sum(t.count(c) for c in ")]}")
        s += " " + t
    return s, j


def _tuple_arities(body):
    """sizes of the tuples a block of code can evaluate to: `return a, b`, `return (a, b)`, and a parenthesised tuple that is the
    last statement before an `end` (the value of a `do` block or of a function)"""
    out = set()
    lines = body.splitlines()
    idx = 0
    while idx < len(lines):
        s, last = _joined_statement(lines, idx)
        expr = None
        m = re.match(r"(?:.*?&&\s*|.*?\|\|\s*)?return\s+(.+)$", s)
        if m:
            expr = m.group(1)
        elif s.startswith("(") and _match_paren(s, 0) == len(s):
            nxt = next((l.strip() for l in lines[last + 1:] if l.strip()), "")
            if nxt == "end" or nxt.startswith("end "):
                expr = s
        if expr is not None and " ? " not in expr:
            n = _count_top(expr)
            if n >= 2:
                out.add(n)
        idx = last + 1
    return out


def _top_level_functions(code):
    """[(name, first line index, last line index)] of every `function name(...) ... end` that starts in column 0"""
    lines = code.splitlines()
    out = []
    for i, line in enumerate(lines):
        m = re.match(r"function\s+([\w.:!*+\-/]+)\(", line)
        if m:
            j = next(q for q in range(i + 1, len(lines)) if lines[q].rstrip() == "end")
            out.append((m.group(1).split(".")[-1].lstrip(":"), i, j))
    return out


def _destructuring_problems(text):
    """`a, b, c = F(...)` / `a, b, c = v` where F is one of the file's functions (or v a local bound to a `get!(...) do` block, or to
    `cond ? F(...) : (tuple)`) and the number of names differs from the size of a tuple F / the block can return.  Julia accepts
    FEWER names than elements silently and throws on more; here the counts must be equal."""
    code = _julia_code_tokens(text)
    lines = code.splitlines()
    funcs = _top_level_functions(code)
    produced = {}
    for name, i, j in funcs:
        body = "\n".join(lines[i:j + 1])
        sizes = _tuple_arities(body)
        # `return NAME[]` of a cell filled with `NAME[] = (a, b)` in the same function (the reduction scratch)
        for cell in re.findall(r"return\s+(\w+)\[\]", body):
            for mm in re.finditer(r"%s\[\]\s*=\s*\(" % cell, body):
                k = mm.end() - 1
                sizes.add(_count_top(body[k:_match_paren(body, k)]))
        produced.setdefault(name, set()).update(sizes)
    # a tuple kept in a struct field between the function that builds it and the one that takes it apart
    FIELD_PRODUCERS = {"st.map": "_spgemm_product_lists"}
    problems, checked = [], 0
    for idx, line in enumerate(lines):
        m = re.match(r"(?:.*?[;(]\s*|\s*)\(?\s*((?:\w+\s*,\s*)+\w+)\s*\)?\s*=(?!=)\s*(.*)$", line)
        if not m or re.match(r"\s*(for|function)\b", line):
            continue
        names = len(re.findall(r"\w+", m.group(1)))
        rhs = m.group(2).strip()
        if not rhs:                                           # `a, b, c =` with the call on the next line
            rhs, _ = _joined_statement(lines, idx + 1)
        else:
            stmt, _ = _joined_statement(lines, idx)
            rhs = stmt[stmt.index(m.group(1)) + len(m.group(1)):].split("=", 1)[1].strip()
        owner = next(((n, i, j) for n, i, j in funcs if i <= idx <= j), ("<top level>", idx, idx))
        sizes = set()
        call = re.match(r"([\w.]+!?)\(", rhs)
        if call and call.group(1).split(".")[-1] in produced:
            sizes = produced[call.group(1).split(".")[-1]]
        elif rhs in FIELD_PRODUCERS:
            sizes = produced[FIELD_PRODUCERS[rhs]]
        elif re.fullmatch(r"\w+", rhs):
            # a local: `v = get!(...) do` ... `end` at the same indentation, or `v = cond ? F(...) : (tuple)`
            for k in range(idx - 1, owner[1] - 1, -1):
                mm = re.match(r"(\s*)%s\s*=(?!=)\s*(.+)$" % re.escape(rhs), lines[k])
                if not mm:
                    continue
                stmt, _ = _joined_statement(lines, k)
                val = stmt.split("=", 1)[1].strip()
                if re.match(r"get!\(.*\)\s+do\b", val):
                    ind = mm.group(1)
                    stop = next(q for q in range(k + 1, len(lines)) if lines[q].rstrip() == ind + "end")
                    sizes = _tuple_arities("\n".join(lines[k + 1:stop + 1]))
                else:
                    t = re.match(r".+?\?\s*([\w.]+!?)\((?:.*)\)\s*:\s*(\(.*\))$", val)
                    if t and t.group(1).split(".")[-1] in produced:
                        sizes = set(produced[t.group(1).split(".")[-1]]) | {_count_top(t.group(2))}
                break
        if not sizes:
            continue
        checked += 1
        if sizes != {names}:
            problems.append(f"{owner[0]} (line {idx + 1}): `{line.strip()[:90]}` takes {names} names from a tuple of {sorted(sizes)}")
    return problems, checked


def test_tuples_are_taken_apart_with_as_many_names_as_they_have_elements():
    """The cached plan entries of the extension are plain tuples (`_spmm_halo`'s seven fields, the entries of `_rocm_exec` and
    `_rocm_matexec`, ...), built in one place and destructured in several.  A field added on one side only is a BoundsError -- or,
    with fewer names than fields, silently the wrong field -- at the first call, which nothing here can make.  Held statically,
    with two mutations that must be caught."""
    text = open(EXT).read()
    problems, checked = _destructuring_problems(text)
    assert checked >= 14, f"only {checked} destructuring assignments were resolved: the parser lost track of the file"
    assert not problems, "\n".join(problems)
    old = "ROCVector(perm), ROCVector(collect(Int64, 1:length(perm))), send_idx, ghost[])"
    assert text.count(old) == 1
    got, _ = _destructuring_problems(text.replace(old, "ROCVector(perm), ROCVector(collect(Int64, 1:length(perm))), send_idx)"))
    assert any("execute_plan!" in p for p in got), got
    old2 = "        (halo[], interior, boundary, send_idx, ghost[], colval_split, rp0)\n"
    assert text.count(old2) == 1
    got, _ = _destructuring_problems(text.replace(old2, "        (halo[], interior, boundary, send_idx, ghost[], colval_split)\n"))
    assert len(got) >= 3 and all("_spmm_halo" in p or "ent" in p or "*" in p for p in got), got
