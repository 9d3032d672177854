"""Static audit of integration/HPCLinearAlgebraROCmExt.jl: no bound operator may stage device data through the host in
its PER-CALL path ("Next round" item 1).

The extension cannot be executed here (no Julia), and the defect class this test exists for is invisible to the other two
static checks (names / arities / @ccall types): round 5's `A * B` returned through the parent's `HPCMatrix_local`, whose
`Matrix(A_local)` (src/dense.jl:153) copies the whole product device -> host -> device -- correct types, correct values, two
268 MB PCIe copies around a 1.5 ms kernel.  So the file is walked method by method:

* PLAN-TIME code is set apart structurally: the body of every `get!(cache, key) do ... end` block (runs once per cached plan),
  every region between `# >>> plan time: <why>` and `# <<< plan time` markers, and the functions of PLAN_TIME_FUNCTIONS (each
  with the reason it only runs at setup).
* In everything else -- the per-call path -- a line that moves data across PCIe or calls a parent function whose reference
  body does must carry a `# PCIe: <bytes> -- <why>` annotation:
    - device -> host: `Array(`, `Matrix(`, `Vector(`, `collect(`;
    - host -> device: `ROCVector(`, `ROCArray(`, `ROCMatrix(` (the `{T}(undef, ...)` forms are device allocations);
    - the parent's constructors and helpers that stage: `HPCMatrix_local(`, `HPCVector_local(`, `HPCSparseMatrix_local(`,
      `_ensure_cpu`, `_copy_range_to_cpu`, `_values_to_backend`, `_matrix_to_backend`, `_copy_to_output!`,
      `HPCLinearAlgebra.execute_plan!(` (the parent's MatrixPlan / VectorPlan executors), `HPCLinearAlgebra.repartition(`
      (host-staged unless this file's device executor takes it), `invoke(` (parent fallbacks).
    An annotation that starts with `none` records a checked line that moves nothing (e.g. a parent function that lands in this
    file's device method); it satisfies the rule and does not count as a transfer.
* `HPCMatrix_local(` / `HPCVector_local(` with a device array are forbidden outright in the per-call path (no annotation
  makes a whole-product round trip acceptable).
* INTEGRATION.md section 6 holds one row per bound operator (method name + argument types + element type) with the bytes that
  cross PCIe per call; the row must say "none" exactly when no annotated line is reachable from the method through the
  file's own helpers.

A mutation self-test re-introduces the round-5 defect and four relatives and checks that each is reported.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = os.path.join(ROOT, "integration", "HPCLinearAlgebraROCmExt.jl")

# functions that only run at setup / plan time / teardown, with the reason (everything inside them is exempt)
PLAN_TIME_FUNCTIONS = {
    "_rccl": "communicator bootstrap, once per process",
    "_attach_comm_window": "communicator bootstrap, once per process",
    "_attach_halo_window": "plan time: window attach + connection test, once per exchange plan",
    "_split_colval": "plan time: called from _device_plan / _spmm_halo inside their get! blocks",
    "_classify_blocks": "plan time: called from _device_plan / _spmm_halo inside their get! blocks",
    "_whole_slice_wishes": "host-only arithmetic on the plan's lists (no device data)",
    "_spgemm_symbolic": "plan time: symbolic state, once per reference MatrixPlan",
    "_spgemm_product_lists": "plan time: per-entry product lists, once per structure (third product)",
    "HPCLinearAlgebra.VectorPlan": "plan constructor (memoized by get_vector_plan)",
    "HPCLinearAlgebra.HPCSparseMatrix_local": "matrix CONSTRUCTOR: the caller's host CSR goes up once, the struct's host fields come back once",
    "HPCLinearAlgebra._ensure_hash": "memoized in A.structural_hash: once per matrix (4 digest words come back)",
    "HPCLinearAlgebra.backend_rocm_serial": "factory",
    "HPCLinearAlgebra.backend_rocm_mpi": "factory",
    "HPCLinearAlgebra._convert_array": "conversion hook: IS the host <-> device move the caller asked for (to_backend)",
    "HPCLinearAlgebra._to_target_device": "conversion hook",
    "HPCLinearAlgebra._array_to_device": "conversion hook",
    "HPCLinearAlgebra._convert_vector_to_device": "conversion hook",
    "clear_rocm_plan_cache!": "teardown",
    "_check_exchange_health": "error path: reads status words after a NaN reached the host",
    "_scratch": "allocated once per process",
    "_partition_hash": "host-only (hash of a partition vector)",
    "HPCLinearAlgebra._zeros_device": "allocation hook (device fill, nothing crosses PCIe)",
    "HPCLinearAlgebra._index_array_type": "type hook",
}

PULL = re.compile(r"(?<![\w.{])(Array|Matrix|Vector|collect)\(")
PUSH = re.compile(r"(?<![\w.{])(ROCVector|ROCArray|ROCMatrix)\(")
PARENT_STAGING = re.compile(r"(HPCMatrix_local|HPCVector_local|HPCSparseMatrix_local)\(|\b(_ensure_cpu|_copy_range_to_cpu|"
                            r"_values_to_backend|_matrix_to_backend|_copy_to_output!)\b|HPCLinearAlgebra\.(execute_plan!|repartition)\(|(?<![\w.])invoke\(")
FORBIDDEN = re.compile(r"(HPCMatrix_local|HPCVector_local)\(")
ANNOT = re.compile(r"#\s*PCIe:\s*(.+)$")


def _strip_strings(line):
    return re.sub(r'"(?:\\.|[^"\\])*"', '""', line)


def _code(line):
    """the line without string literals and without its comment"""
    return _strip_strings(line).split("#", 1)[0]


def parse_functions(text):
    """[(name, first line number, last line number, [lines])] of every top-level definition: `function NAME(...) ... end`
    at column 0, and one-line definitions `NAME(args) = ...` / `NAME(args) where {...} = ...` at column 0 (with their indented
    continuation lines)."""
    lines = text.split("\n")
    out = []
    i = 0
    name_re = r"((?:[A-Za-z_]\w*\.)*(?::?[^\s(]+))"
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"function\s+" + name_re + r"\(", ln)
        if m:
            j = i + 1
            while j < len(lines) and lines[j] != "end":
                j += 1
            assert j < len(lines), f"unterminated function at line {i + 1}"
            out.append((m.group(1), i + 1, j + 1, lines[i:j + 1]))
            i = j + 1
            continue
        m = None
        if ln and not ln[0].isspace() and not ln.startswith(("#", "const", "using", "module", "end", "mutable", "struct")):
            m = re.match(name_re + r"\(", ln)
        if m:
            # the signature may run over several lines: join until its parentheses balance, then an `=` (not `==`) must follow
            j, depth, sig = i, 0, ""
            while j < len(lines):
                sig += _code(lines[j]) + " "
                depth += _code(lines[j]).count("(") - _code(lines[j]).count(")")
                j += 1
                if depth <= 0:
                    break
            close = _match_close(sig, sig.index("("))
            if re.match(r"\s*(?:where\s*(?:\{[^=]*\}|\w+)\s*)?=(?!=)", sig[close:]):
                while j < len(lines) and lines[j].startswith((" ", "\t")) and lines[j].strip():
                    j += 1
                out.append((m.group(1), i + 1, j, lines[i:j]))
                i = j
                continue
        i += 1
    return out


def _match_close(text, i):
    """index just behind the parenthesis that closes the one at text[i]"""
    depth = 0
    for j in range(i, len(text)):
        depth += {"(": 1, ")": -1}.get(text[j], 0)
        if depth == 0:
            return j + 1
    return len(text)


def method_key(body):
    """Stable name of a method for the INTEGRATION.md table: `Name(ArgType, ArgType, ...) Float64|Float32` from its signature
    (argument type names without parameters and module prefixes; keyword arguments dropped)."""
    sig = " ".join(_code(l) for l in body[:6])
    sig = re.sub(r"^function\s+", "", sig)
    i = sig.index("(")
    name = sig[:i].strip()
    args = sig[i + 1:_match_close(sig, i) - 1]
    parts, depth, cur = [], 0, ""
    for ch in args:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == ";" and depth == 0:
            break
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    types = []
    for a in parts:
        t = a.split("::", 1)[1] if "::" in a else "Any"
        t = re.split(r"[{=]", t, 1)[0].strip()
        types.append(t.split(".")[-1])
    elt = "Float32" if "Float32" in sig[:_match_close(sig, i) + 80] else "Float64"
    return f"{name}({', '.join(types)}) {elt}"


def plan_time_mask(body):
    """per line of a function body: True when the line belongs to a `get!(...) do ... end` block or to a marked plan-time region"""
    mask = [False] * len(body)
    i = 0
    while i < len(body):
        ln = body[i]
        code = _code(ln)
        if re.search(r"\bget!\(.*\)\s+do\s*$", code) or re.search(r"\bget!\(.*\bdo\s*$", code):
            indent = len(ln) - len(ln.lstrip())
            j = i + 1
            while j < len(body) and not (body[j].rstrip() == " " * indent + "end"):
                j += 1
            assert j < len(body), f"get! block without its end: {ln.strip()}"
            for k in range(i, j + 1):
                mask[k] = True
            i = j + 1
            continue
        if re.search(r"#\s*>>>\s*plan time:", ln):
            j = i + 1
            while j < len(body) and not re.search(r"#\s*<<<\s*plan time", body[j]):
                j += 1
            assert j < len(body), f"plan-time region without its closing marker: {ln.strip()}"
            for k in range(i, j + 1):
                mask[k] = True
            i = j + 1
            continue
        i += 1
    return mask


def audit(text):
    """(violations, annotations, functions).  violations: [(function, line number, message)]; annotations: {(function name,
    first line): [(line number, text)]} of per-call PCIe annotations."""
    funcs = parse_functions(text)
    violations, annotations = [], {}
    for name, first, last, body in funcs:
        key = (name, first)
        annotations[key] = []
        if name in PLAN_TIME_FUNCTIONS:
            continue
        mask = plan_time_mask(body)
        one_liner = not body[0].startswith("function")
        for off, ln in enumerate(body):
            if (off == 0 and not one_liner) or mask[off]:
                continue
            code = _code(ln)
            a = ANNOT.search(ln)
            hits = []
            if PULL.search(code):
                hits.append("device -> host copy")
            if PUSH.search(code):
                hits.append("host -> device copy")
            if PARENT_STAGING.search(code):
                hits.append("parent function that stages through the host")
            if FORBIDDEN.search(code):
                violations.append((name, first + off, "HPCMatrix_local / HPCVector_local in a per-call path: the parent's "
                                                      "constructor copies a device array to the host and back (src/dense.jl:153)"))
                continue
            if hits and not a:
                violations.append((name, first + off, f"{' + '.join(hits)} in a per-call path without a `# PCIe:` annotation: {ln.strip()[:110]}"))
            if a and not hits:
                violations.append((name, first + off, f"`# PCIe:` annotation on a line that moves nothing: {ln.strip()[:110]}"))
            if a and hits and not a.group(1).strip().lower().startswith("none"):
                annotations[key].append((first + off, a.group(1).strip()))
    return violations, annotations, funcs


def reachable_annotations(funcs, annotations):
    """per (name, first line): the annotations reachable through calls to the file's own functions (by name, all methods of a
    helper name taken together); plan-time lines and plan-time functions do not propagate"""
    by_name = {}
    for name, first, last, body in funcs:
        if "." not in name:                      # file-private helpers only: `sum(` on a host list is not this file's Base.sum method
            by_name.setdefault(name, []).append((name, first, body))
    calls = {}
    for name, first, last, body in funcs:
        mask = [False] * len(body) if name in PLAN_TIME_FUNCTIONS else plan_time_mask(body)
        called = set()
        if name not in PLAN_TIME_FUNCTIONS:
            for off, ln in enumerate(body):
                if (off == 0 and body[0].startswith("function")) or mask[off]:
                    continue
                for m in re.finditer(r"(?<![\w.:])([A-Za-z_]\w*!?)\(", _code(ln)):
                    if m.group(1) in by_name and m.group(1) != name:
                        called.add(m.group(1))
        calls[(name, first)] = called
    out = {}
    for key in calls:
        seen, stack, acc = set(), list(calls[key]), list(annotations.get(key, []))
        while stack:
            h = stack.pop()
            if h in seen:
                continue
            seen.add(h)
            for hname, hfirst, _ in by_name[h]:
                acc += annotations.get((hname, hfirst), [])
                stack += list(calls[(hname, hfirst)])
        out[key] = sorted(set(acc))
    return out


def is_operator(name):
    return "." in name or name in ("rocm_cg_iterations",)


def _text():
    return open(EXT).read()


def test_parser_sees_the_methods_of_the_extension():
    funcs = parse_functions(_text())
    names = [f[0] for f in funcs]
    for want in ("Base.:*", "LinearAlgebra.mul!", "LinearAlgebra.dot", "LinearAlgebra.norm", "HPCLinearAlgebra.execute_plan!",
                 "HPCLinearAlgebra.VectorPlan", "_spmm_colmajor", "_spmm_split!", "_device_plan", "_spmm_halo", "_matrix_values!",
                 "Base.:+", "Base.:-", "Base.:/", "_host_scalar", "_spmm_result"):
        assert want in names, f"{want} not found by the parser"
    assert names.count("Base.:*") >= 7           # A*x (f64, f32), A*M (f64, f32), A*B sparse, dense A*x, a*v, v*a
    assert len(funcs) >= 60
    # every plan-time exemption names a function that exists
    missing = [n for n in PLAN_TIME_FUNCTIONS if n not in names]
    assert not missing, f"PLAN_TIME_FUNCTIONS names functions the file does not define: {missing}"


def test_no_per_call_host_staging():
    violations, _, _ = audit(_text())
    assert not violations, "\n".join(f"{n} (line {ln}): {msg}" for n, ln, msg in violations)


def test_spmm_returns_through_the_inner_constructor():
    """The three A * B::HPCMatrix methods and A * x build their results with the struct's inner constructor (HPCMatrix{T,B}(...),
    HPCVector{T,B}(...)), as src/sparse.jl:2122-2127 does, never through HPCMatrix_local / HPCVector_local."""
    text = _text()
    assert not re.search(r"HPCLinearAlgebra\.HPCMatrix_local\(", "\n".join(_code(l) for l in text.split("\n")))
    assert text.count("return _spmm_result(A, C, k)") == 3
    m = re.search(r"_spmm_result\(A::HPCSparseMatrix\{T,Ti,B\}, C::ROCMatrix\{T\}, k::Int\) where \{T,Ti,B\} =\s*\n\s*HPCMatrix\{T,B\}\(nothing, "
                  r"copy\(A\.row_partition\), HPCLinearAlgebra\.uniform_partition\(k, comm_size\(A\.backend\.comm\)\), C, A\.backend\)", text)
    assert m, "_spmm_result must build HPCMatrix{T,B}(nothing, copy(A.row_partition), uniform_partition(k, nranks), C, A.backend)"


def _integration_rows():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 6. Bytes that cross PCIe per call"):]
    nxt = re.search(r"\n## \d", sec[5:])
    sec = sec[:nxt.start() + 5] if nxt else sec
    rows = {}
    for ln in sec.split("\n"):
        m = re.match(r"\|\s*`([^`]+)`\s*\|([^|]*)\|([^|]*)\|", ln)
        if m:
            assert m.group(1) not in rows, f"duplicate row {m.group(1)}"
            rows[m.group(1)] = (m.group(2).strip(), m.group(3).strip())
    return rows


def test_integration_md_pcie_table_matches_the_annotations():
    violations, annotations, funcs = audit(_text())
    reach = reachable_annotations(funcs, annotations)
    rows = _integration_rows()
    ops = {}
    for n, first, _, body in funcs:
        if is_operator(n) and n not in PLAN_TIME_FUNCTIONS:
            mk = method_key(body)
            assert mk not in ops, f"two methods share the table key {mk}"
            ops[mk] = (n, first)
    problems = []
    for mk, key in sorted(ops.items(), key=lambda kv: kv[1][1]):
        if mk not in rows:
            problems.append(f"`{mk}` (extension line {key[1]}) has no row in INTEGRATION.md section 6")
            continue
        says_none = rows[mk][0].lower().startswith("none")
        if says_none and reach[key]:
            problems.append(f"`{mk}`: the table says none, the method reaches {reach[key]}")
        if not says_none and not reach[key]:
            problems.append(f"`{mk}`: the table says '{rows[mk][0]}', no annotated transfer is reachable")
    for mk in rows:
        if mk not in ops:
            problems.append(f"INTEGRATION.md section 6 row `{mk}` matches no per-call operator of the file")
    assert not problems, "\n".join(problems)


MUTATIONS = [
    # (description, old, new, substring expected in some violation message)
    ("round 5's defect: A * B returns through the parent's HPCMatrix_local",
     "    return _spmm_result(A, C, k)\nend\n\n# Row-major B rows",
     "    return HPCLinearAlgebra.HPCMatrix_local(C, A.backend)\nend\n\n# Row-major B rows", "HPCMatrix_local"),
    ("a product buffer pulled to the host inside the launch helper",
     "    _spmm_order!(rp0, 1)                               # the gather kernel's launches",
     "    Ch = Array(C)\n    _spmm_order!(rp0, 1)                               # the gather kernel's launches", "device -> host"),
    ("sparse A * B back on the parent's host-staged value exchange",
     "    _matrix_values!(gval, plan, Bm)", "    HPCLinearAlgebra.execute_plan!(plan, Bm, gval)", "parent function"),
    ("a block list built on the host and uploaded per product",
     "        _spmm_split!(C, A, d.rowptr0, d.colval_split, d.n_own, Brow, kp, C_NULL, k, nothing, true)",
     "        _spmm_split!(C, A, d.rowptr0, d.colval_split, d.n_own, Brow, kp, C_NULL, k, ROCVector(Int32.(0:9)), true)", "host -> device"),
    ("the scalar read-back loses its annotation",
     "# PCIe: 8 B", "# 8 B", "without a `# PCIe:` annotation"),
]


def test_mutations_are_caught():
    text = _text()
    assert not audit(text)[0]
    for what, old, new, expect in MUTATIONS:
        assert text.count(old) >= 1, f"mutation anchor not found ({what}): {old[:60]!r}"
        mutated = text.replace(old, new, 1)
        violations, _, _ = audit(mutated)
        assert any(expect in msg for _, _, msg in violations), f"mutation not caught: {what}\n{violations}"
