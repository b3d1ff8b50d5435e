"""Mechanical lint of the Julia shim in INTEGRATION.md against include/lsqhip.h (the part of "the shim would
load and bind correctly" that needs no Julia runtime):

* every `ccall((:name, lib), Ret, (argtypes...), args...)` is parsed; `name` must be declared in the header, the return type
  and every argument type must be the Julia spelling of the C type at that position, and the number of values passed must
  equal the number of types;
* the `LsqOptions` / `LsqResult` mirror structs must match `lsq_options` / `lsq_result` field by field (name, type, order);
* structural checks of the dispatch the reference relies on (`required_methods`, `allocated_solver_methods`).
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lsqhip.h")
RCCL_HEADER = os.path.join(ROOT, "include", "lsqrccl.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")


# ------------------------------------------------------------------------------------------------ C side
def _strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"^\s*#.*$", " ", txt, flags=re.M)                  # preprocessor lines
    return txt.replace('extern "C" {', " ")


def c_class(ctype):
    """Normalise a C parameter/return type to the class a Julia ccall type must belong to."""
    t = re.sub(r"\bconst\b", " ", ctype)
    t = re.sub(r"\s+", " ", t).strip()
    t = t.replace(" *", "*").replace("* ", "*")
    stars = t.count("*")
    base = t.replace("*", "").strip()
    callbacks = {"lsq_f_callback", "lsq_g_callback", "lsq_allreduce_callback", "lsq_precond_callback", "lsq_device_allreduce_callback",
                 "lsq_precond_update_callback", "lsq_precond_ldiv_callback",
                 "lsq_op_mul_callback", "lsq_op_colsum_callback", "lsq_xchg_issue_fn", "lsq_xchg_finish_fn"}
    handles = {"lsq_ctx", "lsq_mat", "lsq_solver", "lsq_model", "void"}
    if base in callbacks and stars == 0:
        return "ptr:void"
    if stars == 0:
        return {"int": "int", "double": "double", "size_t": "size_t", "float": "float", "long long": "longlong",
                "unsigned long long": "ulonglong", "void": "void"}[base]
    if stars == 1:
        if base in handles:
            return "ptr:void"
        if base in ("lsq_options", "lsq_result"):
            return "ptr:" + base
        return "ptr:" + {"int": "int", "double": "double", "float": "float", "char": "char", "long long": "longlong",
                         "unsigned char": "uchar"}[base]
    if stars == 2 and base in handles:
        return "ptrptr"
    raise ValueError("unclassified C type %r" % ctype)


def header_prototypes(path=None):
    txt = _strip_comments(open(path or HEADER).read())
    txt = re.sub(r"typedef\s+(struct|enum)\s*\{.*?\}\s*\w+\s*;", " ", txt, flags=re.S)
    txt = re.sub(r"typedef[^;]*;", " ", txt)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(lsq_\w+)\s*\(([^;{}]*?)\)\s*;", txt):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):            # every parameter of the header is named: `type [*]name` or `type name[k]`
                a = a.strip()
                arr = re.search(r"\[\d*\]$", a)
                if arr:                                                # `double h_avg_ms[2]` is a pointer parameter
                    a = a[:arr.start()]
                a = re.sub(r"\w+$", "", a.strip())                     # drop the parameter name
                params.append(c_class(a + ("*" if arr else "")))
        protos[name] = (c_class(ret), params)
    return protos


def header_struct(name):
    txt = _strip_comments(open(HEADER).read())
    m = re.search(r"typedef\s+struct\s*\{([^{}]*)\}\s*%s\s*;" % name, txt, flags=re.S)
    fields = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = [d.strip() for d in decl.split(",")]
        mm = re.match(r"(.*?)(\**)\s*(\w+)$", first)
        base, stars, fname = mm.group(1).strip(), mm.group(2), mm.group(3)
        fields.append((fname, c_class(base + stars)))
        for r in rest:
            mm = re.match(r"(\**)\s*(\w+)$", r)
            fields.append((mm.group(2), c_class(base + mm.group(1))))
    return fields


# ------------------------------------------------------------------------------------------------ Julia side
JULIA_CLASS = {"Cint": {"int"}, "Cdouble": {"double"}, "Csize_t": {"size_t"}, "Cfloat": {"float"},
               "Clonglong": {"longlong"}, "Culonglong": {"ulonglong"}, "Cvoid": {"void"}, "Cstring": {"ptr:char"},
               "Ptr{Cvoid}": {"ptr:void", "ptr:lsq_options", "ptr:lsq_result", "ptrptr"},
               "Ptr{Cdouble}": {"ptr:double"}, "Ref{Cdouble}": {"ptr:double"},
               "Ptr{Cint}": {"ptr:int"}, "Ref{Cint}": {"ptr:int"}, "Ptr{Cfloat}": {"ptr:float"}, "Ref{Cfloat}": {"ptr:float"},
               "Ref{Ptr{Cvoid}}": {"ptrptr"}, "Ref{LsqOptions}": {"ptr:lsq_options"}, "Ref{LsqResult}": {"ptr:lsq_result"},
               "Ptr{UInt8}": {"ptr:uchar", "ptr:char"}, "Ptr{Clonglong}": {"ptr:longlong"}, "Ref{Clonglong}": {"ptr:longlong"}}


def julia_blocks():
    return "\n".join(re.findall(r"```julia\n(.*?)```", open(DOC).read(), flags=re.S))


def _split_top(s):
    """Split on top-level commas (respecting (), {}, [] and strings / #= =# comments)."""
    out, depth, cur, i = [], 0, "", 0
    while i < len(s):
        c = s[i]
        if s.startswith("#=", i):
            j = s.index("=#", i) + 2
            i = j
            continue
        if c in "({[":
            depth += 1
        elif c in ")}]":
            depth -= 1
        if c == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += c
        i += 1
    if cur.strip():
        out.append(cur.strip())
    return out


def _balanced(s, start):
    """s[start] == '(' -> index just past the matching ')'."""
    depth = 0
    for i in range(start, len(s)):
        if s[i] in "([{":
            depth += 1
        elif s[i] in ")]}":
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced")


def ccalls(libname="lib"):
    """[(name, ret, [argtypes], nvalues, line)] for every ccall((:name, <libname>), ...) in the julia blocks of INTEGRATION.md
    (lib = liblsqhip.so / include/lsqhip.h, rlib = liblsqrccl.so / include/lsqrccl.h)."""
    src = "\n".join(line.split(" # ")[0] if not line.lstrip().startswith("#") else "" for line in julia_blocks().split("\n"))
    found = []
    for m in re.finditer(r"ccall\(", src):
        end = _balanced(src, m.end() - 1)
        parts = _split_top(src[m.end():end - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*(r?lib)\s*\)", parts[0])
        assert sym, "ccall without (:name, lib) / (:name, rlib): %s" % parts[0]
        if sym.group(2) != libname:
            continue
        ret = parts[1]
        assert parts[2].startswith("(") and parts[2].endswith(")"), parts[2]
        types = _split_top(parts[2][1:-1])
        found.append((sym.group(1), ret, types, len(parts) - 3, src[:m.start()].count("\n") + 1))
    return found


def julia_struct(name):
    m = re.search(r"struct %s\b[^\n]*\n(.*?)\nend" % name, julia_blocks(), flags=re.S)
    fields = []
    for decl in re.split(r"[;\n]", m.group(1)):
        decl = decl.split("#")[0].strip()
        if decl:
            fname, ftype = decl.split("::")
            fields.append((fname.strip(), ftype.strip()))
    return fields


# what the reference's own code calls on nls.x / nls.y / nls.J (and therefore the shim must define), with the line that
# needs it; (regex over the shim's julia text)
REQUIRED_METHODS = [
    (r"Base\.adjoint\(J::HipCSC\)", "J' at levenberg_marquardt.jl:102, dogleg.jl:99"),
    (r"Base\.adjoint\(J::HipDense\)", "J' at levenberg_marquardt.jl:102, dogleg.jl:99"),
    (r"LinearAlgebra\.mul!\(y::HipVector, J::HipCSC, x::HipVector", "mul!(fpredict, J, δx, 1, 0) levenberg_marquardt.jl:114"),
    (r"LinearAlgebra\.mul!\(y::HipVector, J::HipDense, x::HipVector", "same, dense J"),
    (r"LinearAlgebra\.mul!\(x::HipVector, Jt::HipAdjoint, y::HipVector", "mul!(dtd, J', fcur, 1, 0) levenberg_marquardt.jl:102"),
    (r"LeastSquaresOptim\.colsumabs2!\(v::HipVector, J::HipCSC\)", "colsumabs2!(dtd, J) levenberg_marquardt.jl:82"),
    (r"LeastSquaresOptim\.colsumabs2!\(v::HipVector, J::HipDense\)", "colsumabs2!(dtd, J) dogleg.jl:85"),
    (r"Base\.size\(J::HipCSC, d\)", "size(J, 2) types.jl:14-15"),
    (r"Base\.size\(J::HipDense, d\)", "size(J, 2) types.jl:14-15"),
    (r"Base\.similar\(v::HipVector\)", "_zeros(x) = fill!(similar(x), 0) utils.jl:163"),
    (r"Base\.fill!\(v::HipVector", "fill!(δgn, 0) dogleg.jl:117"),
    (r"Base\.copyto!\(d::HipVector, s::HipVector\)", "copyto!(fcur, ftrial) levenberg_marquardt.jl:127"),
    (r"LinearAlgebra\.rmul!\(v::HipVector", "rmul!(dtd, 1/Δ) levenberg_marquardt.jl:86"),
    (r"LinearAlgebra\.axpy!\(a::Number, x::HipVector, y::HipVector\)", "axpy!(-1, δx, x) levenberg_marquardt.jl:106"),
    (r"Base\.sum\(::typeof\(abs2\), v::HipVector\)", "sum(abs2, fcur) levenberg_marquardt.jl:60"),
    (r"Base\.sum\(v::HipVector\)", "sum(dtd) levenberg_marquardt.jl:84"),
    (r"Base\.clamp!\(v::HipVector", "clamp!(dtd, ...) levenberg_marquardt.jl:85"),
    (r"Base\.maximum\(::typeof\(abs\), v::HipVector\)", "maximum(abs, δx) utils.jl:24"),
    (r"Base\.map!\(::typeof\(/\), o::HipVector", "map!(/, δgr, δgr, dtd) dogleg.jl:105"),
    (r"LeastSquaresOptim\.wdot\(x::HipVector, y::HipVector, w::HipVector\)", "wnorm / wdot dogleg.jl:93,106,134"),
    (r"LeastSquaresOptim\.maxabs_projected_gradient\(g::HipVector", "levenberg_marquardt.jl:104"),
    (r"LinearAlgebra\.norm\(v::HipVector\)", "lsmr.jl:74 (only if the reference's own LSMR runs on device vectors)"),
    (r"LinearAlgebra\.ldiv!\(x::HipVector, J::HipJacobian, y::HipVector, A::HipAllocatedSolver\)", "dogleg.jl:115"),
    (r"LinearAlgebra\.ldiv!\(x::HipVector, J::HipJacobian, y::HipVector, damp::HipVector, A::HipAllocatedSolver\)",
     "levenberg_marquardt.jl:87"),
]

# the reference's AbstractAllocatedSolver methods (argument 2 as written there)
REFERENCE_SOLVER_METHODS = {
    "Dogleg{LSMR{T1,T2}}": "iterative_lsmr.jl:173", "LevenbergMarquardt{LSMR{T1,T2}}": "iterative_lsmr.jl:233",
    "Dogleg{QR}": "dense_qr.jl:25", "LevenbergMarquardt{QR}": "dense_qr.jl:50",
    "Dogleg{Cholesky}": "dense_cholesky.jl:19 (AbstractOptimizer{Cholesky})",
    "LevenbergMarquardt{Cholesky}": "dense_cholesky.jl:19 (AbstractOptimizer{Cholesky})",
}


def allocated_solver_methods():
    """[(arg1 type, arg2 type)] of every AbstractAllocatedSolver method the shim defines."""
    out = []
    for m in re.finditer(r"AbstractAllocatedSolver\(nls::([^,]+),\s*o::([^)]+)\)", julia_blocks()):
        out.append((m.group(1).strip(), re.sub(r"\s+", "", m.group(2))))
    return out
