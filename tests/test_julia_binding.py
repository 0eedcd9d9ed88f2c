"""CPU: the Julia binding (julia/hip_ext/*.jl) cannot be executed here (no Julia toolchain), so it is checked as text:

  1. every `ccall((:name, libcnf), Cint, (types...), ...)` names an entry point of include/cnf.h with the same number of
     parameters and compatible scalar kinds;
  2. every method the binding adds to a reference generic function (`augmented_f`, `base_sol`, `inference_sol`,
     `make_ode_func`, the `rrule`s for `loss`) is, argument by argument, either at least as specific as or provably disjoint
     from EVERY reference method of the same name and arity — i.e. it can never be ambiguous with one (the
     markdown sketch's `mode::Mode` / untyped `prob` methods were);
  3. the helpers the methods call are defined in the binding, and core.jl includes every file.

The reference's method table is the committed fixture tests/golden/reference_signatures.json (argument types + file:line);
when the reference tree is present it is re-derived and must agree."""
import glob
import json
import os
import re

import pytest

from conftest import ROOT
from jl_signatures import _balanced, methods, split_top, type_params

JL_DIR = os.path.join(ROOT, "julia", "hip_ext")
JL_FILES = sorted(glob.glob(os.path.join(JL_DIR, "*.jl")))


def header_protos():
    hdr = open(os.path.join(ROOT, "include", "cnf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {m.group(1): [a.strip() for a in m.group(2).split(",")]
              for m in re.finditer(r"\bint\s+(cnf_\w+)\s*\(([^;{]*?)\)\s*;", hdr)}
    for m in re.finditer(r"\bconst\s+char\s*\*\s*(cnf_\w+)\s*\(([^;{]*?)\)\s*;", hdr):   # the entry points that return a string
        protos[m.group(1)] = [a.strip() for a in m.group(2).split(",")]
    assert protos["cnf_last_error"] == ["void"]
    return protos


def ccalls(code):
    for m in re.finditer(r"ccall\(\s*\(:(\w+),\s*(\w+)\),\s*(\w+),\s*\(", code):
        name, lib, start = m.group(1), m.group(2), m.end()
        end = _balanced(code, start - 1)
        types = [re.sub(r"\s+", "", t) for t in split_top(code[start:end - 1])]
        yield name, lib, m.group(3), types


def test_binding_files_exist_and_core_includes_them():
    names = {os.path.basename(f) for f in JL_FILES}
    assert {"core.jl", "types.jl", "libcnf.jl", "handle.jl", "hot_path.jl", "rrule.jl", "comm.jl"} <= names
    core = open(os.path.join(JL_DIR, "core.jl")).read()
    included = set(re.findall(r'^include\("(\w+\.jl)"\)', core, flags=re.M))
    assert included == names - {"core.jl", "amdgpu.jl"}, included      # amdgpu.jl is the optional weak-dependency file
    order = re.findall(r'^include\("(\w+\.jl)"\)', core, flags=re.M)
    assert order.index("types.jl") < order.index("libcnf.jl") < order.index("handle.jl") < order.index("hot_path.jl") < order.index("rrule.jl")


def test_every_ccall_matches_the_header():
    protos = header_protos()
    seen = set()
    for f in JL_FILES:
        code = open(f).read()
        for name, lib, ret, types in ccalls(code):
            if lib != "libcnf":
                assert lib == "libhip" and name in ("hipMalloc", "hipFree", "hipMemcpy"), (f, name, lib)
                continue
            assert name in protos, f"{f}: ccall of an unknown entry point {name}"
            params = [] if protos[name] == ["void"] else protos[name]
            assert len(types) == len(params), (f, name, types, params)
            assert ret == ("Cstring" if name in ("cnf_last_error", "cnf_kernel_name", "cnf_build_info") else "Cint"), (name, ret)
            for jt, cp in zip(types, params):
                if jt in ("Cint", "Int32"):
                    assert re.match(r"(int|int32_t)\s+\w+$", cp), (name, jt, cp)
                elif jt == "Int64":
                    assert cp.startswith("int64_t "), (name, jt, cp)
                elif jt == "Cfloat":
                    assert cp.startswith("float ") and "*" not in cp, (name, jt, cp)
                elif jt == "Csize_t":
                    assert cp.startswith("size_t "), (name, jt, cp)
                else:
                    assert jt.startswith(("Ptr{", "Ref{")) and "*" in cp, (name, jt, cp)
                    if "Float64" in jt:
                        assert "double" in cp, (name, jt, cp)
                    if "Float32" in jt:
                        assert "float" in cp, (name, jt, cp)
            seen.add(name)
    need = {"cnf_create", "cnf_destroy", "cnf_set_params", "cnf_aug_f", "cnf_integrate_fixed_dt", "cnf_inference_fixed_dt",
            "cnf_solve_vcabm", "cnf_solve_tsit5", "cnf_loss_adaptive", "cnf_loss_grad_grid", "cnf_loss_grad_adaptive", "cnf_last_error",
            "cnf_comm_init", "cnf_comm_unique_id", "cnf_comm_destroy", "cnf_allreduce_loss",
            "cnf_kernel_family", "cnf_kernel_family_for", "cnf_kernel_name", "cnf_grad_path_for", "cnf_grad_form_for", "cnf_build_info"}
    assert need <= seen, need - seen


def test_config_struct_mirrors_the_header(pkg):
    code = open(os.path.join(JL_DIR, "libcnf.jl")).read()
    body = re.search(r"struct CnfConfig\n(.*?)\nend", code, flags=re.S).group(1)
    fields = [l.strip() for l in body.splitlines() if l.strip()]
    names = [f.split("::")[0] for f in fields]
    assert names == [n for n, _ in pkg._lib.CnfConfig._fields_]
    assert "widths::NTuple{9, Int32}" in fields and "acts::NTuple{8, Int32}" in fields
    assert all(f.endswith("::Int32") for f in fields if not f.startswith(("widths", "acts")))


# ---- dispatch: never ambiguous with a reference method ---------------------------------------------------------
MODES_BELOW = {"ComputeMode": None, "MatrixMode": "ComputeMode", "VectorMode": "ComputeMode", "DIVectorMode": "VectorMode",
               "DIVecJacVectorMode": "DIVectorMode", "DIJacVecVectorMode": "DIVectorMode", "DIMatrixMode": "MatrixMode",
               "DIVecJacMatrixMode": "DIMatrixMode", "DIJacVecMatrixMode": "DIMatrixMode", "LuxMatrixMode": "MatrixMode",
               "LuxVecJacMatrixMode": "LuxMatrixMode", "LuxJacVecMatrixMode": "LuxMatrixMode",
               "HIPMatrixMode": "MatrixMode", "HIPVecJacMatrixMode": "HIPMatrixMode", "HIPJacVecMatrixMode": "HIPMatrixMode"}
SUPER = {"TestMode": "Mode", "TrainMode": "Mode", "AbstractMatrix": "AbstractVecOrMat", "AbstractVector": "AbstractVecOrMat",
         "ICNF": "AbstractICNF", **{k: v for k, v in MODES_BELOW.items() if v}}


def ancestors(name):
    out = [name]
    while out[-1] in SUPER:
        out.append(SUPER[out[-1]])
    return out


def norm(t):
    t = re.sub(r"\s+", "", t)
    t = re.sub(r"\b(SciMLBase|LuxCore|ChainRulesCore|Distributions|Lux)\.", "", t)
    return t


def relation(mine, theirs):
    """'le' (mine <: theirs), 'disjoint', or 'unknown' for two Julia type expressions, by a conservative textual reasoning
    over the type names this package uses.  Type variables (T, INPLACE, ...) are treated as equal when spelled equally."""
    a, b = norm(mine), norm(theirs)
    # every method in play declares `T <: AbstractFloat`: the bare variable and the bound are the same constraint
    a = "<:AbstractFloat" if a == "T" else a
    b = "<:AbstractFloat" if b == "T" else b
    if b == "Any" or a == b:
        return "le"
    if a == "Any":
        return "unknown"
    for pre in ("<:",):
        if a.startswith(pre) and b.startswith(pre):
            return relation(a[2:], b[2:])
    ha, pa = type_params(a)
    hb, pb = type_params(b)
    ha_s, hb_s = ha.replace("typeof(", "typeof("), hb
    if ha_s.startswith("typeof(") or hb_s.startswith("typeof("):
        return "le" if a == b else "disjoint"
    if hb in ancestors(ha):
        # same family: parameters must be pairwise le (missing trailing parameters are free)
        if (ha, hb) == ("ICNF", "AbstractICNF"):
            # ICNF{T, CM, INPLACE, CONDITIONED, AUTONOMOUS, AUGMENTED, STEER, NORM_Z, NORM_J, NORM_Z_AUG} <: AbstractICNF{T, CM, INPLACE, CONDITIONED, AUGMENTED, STEER, NORM_Z_AUG}
            idx = [0, 1, 2, 3, 5, 6, 9]
            pa = [pa[i] if i < len(pa) else None for i in idx]
        rels = []
        for x, y in zip(pa + [None] * len(pb), pb):
            if x is None:                      # mine leaves the parameter free: fine only if theirs is a free type variable too
                rels.append("le" if re.fullmatch(r"[A-Z_]+\d*|T", norm(y)) else "unknown")
                continue
            rels.append(relation(x, y))
        if "disjoint" in rels:
            return "disjoint"
        return "le" if all(r == "le" for r in rels) else "unknown"
    if ha in ancestors(hb):
        # theirs is below mine in the hierarchy
        return "unknown"
    known = set(SUPER) | set(SUPER.values())
    if ha in known and hb in known:
        return "disjoint"                      # unrelated branches of a tree of abstract / concrete types
    if {a, b} <= {"true", "false"}:
        return "le" if a == b else "disjoint"
    if re.fullmatch(r"[A-Z_]+\d*", b) and a in ("true", "false"):
        return "le"                            # a type variable of the reference method
    if re.fullmatch(r"[A-Z_]+\d*", a) and re.fullmatch(r"[A-Z_]+\d*", b):
        return "le"
    return "unknown"


def binding_methods():
    out = []
    for f in JL_FILES:
        text = open(f).read()
        for name, types, line in methods(text, ("augmented_f", "base_sol", "inference_sol", "make_ode_func", "rrule", "loss")):
            out.append((name, types, f"{os.path.relpath(f, ROOT)}:{line}"))
    return out


def reference_table():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))


def test_relation_reasoner_on_known_cases():
    assert relation("ICNF{T, <:HIPMatrixMode, false}", "ICNF{T, <:MatrixMode, false}") == "le"
    assert relation("ICNF{T, <:HIPMatrixMode, false}", "ICNF{T, <:MatrixMode, true}") == "disjoint"
    assert relation("ICNF{T, <:HIPMatrixMode, false}", "ICNF{T, <:LuxVecJacMatrixMode, false}") == "disjoint"
    assert relation("ICNF{T, <:HIPMatrixMode, INPLACE}", "AbstractICNF{T, <:ComputeMode, INPLACE}") == "le"
    assert relation("ICNF{T, <:HIPMatrixMode, INPLACE}", "AbstractICNF{T, <:VectorMode, INPLACE}") == "disjoint"
    assert relation("Mode", "TestMode") == "unknown" and relation("TestMode", "Mode") == "le" and relation("TrainMode", "TestMode") == "disjoint"
    assert relation("AbstractMatrix{T}", "AbstractVecOrMat{T}") == "le" and relation("AbstractMatrix{T}", "AbstractVector{T}") == "disjoint"
    assert relation("SciMLBase.AbstractODEProblem{<:AbstractMatrix{<:Real}, NTuple{2, T}, INPLACE}",
                    "SciMLBase.AbstractODEProblem{<:AbstractVecOrMat{<:Real}, NTuple{2, T}, INPLACE}") == "le"
    assert relation("Any", "SciMLBase.AbstractODEProblem{<:AbstractMatrix{<:Real}, NTuple{2, T}, INPLACE}") == "unknown"   # the r1 sketch's untyped prob
    # the round-1 sketch: augmented_f(..., icnf::ICNF{T,<:HIPMatrixMode,true}, mode::Mode, ...) against the reference's TestMode method
    mine = ["Any", "Any", "Any", "Any", "ICNF{T, <:HIPMatrixMode, true}", "Mode", "Any", "Any", "AbstractMatrix{T}"]
    ref = ["Any", "Any", "Any", "Any", "ICNF{T, <:MatrixMode, true}", "TestMode", "LuxCore.AbstractLuxLayer", "NamedTuple", "AbstractMatrix{T}"]
    rels = [relation(a, b) for a, b in zip(mine, ref)]
    assert "disjoint" not in rels and not all(r == "le" for r in rels)        # = ambiguous: the checker catches the old sketch


def test_no_binding_method_can_be_ambiguous_with_a_reference_method():
    ref = reference_table()
    mine = binding_methods()
    extended = {m[0] for m in mine}
    assert {"augmented_f", "base_sol", "inference_sol", "make_ode_func", "rrule"} <= extended
    assert sum(1 for m in mine if m[0] == "augmented_f") == 4          # TestMode / TrainMode x out-of-place / in-place
    checked = 0
    for name, types, where in mine:
        fname = "loss" if name == "rrule" else name
        mtypes = types[1:] if name == "rrule" else types                # rrule(::typeof(loss), args...)
        if name == "rrule":
            assert norm(types[0]) == "typeof(loss)", where
        for r in ref:
            if r["function"] != fname or len(r["args"]) != len(mtypes):
                continue
            rels = [relation(a, b) for a, b in zip(mtypes, r["args"])]
            ok = "disjoint" in rels or all(x == "le" for x in rels)
            assert ok, f"{where} may be ambiguous with {r['where']}: {list(zip(mtypes, r['args'], rels))}"
            if name != "rrule" and all(x == "le" for x in rels):
                assert [norm(a) for a in mtypes] != [norm(b) for b in r["args"]], f"{where} would overwrite {r['where']}"
            checked += 1
    assert checked >= 30


def test_conditioned_loss_has_an_rrule_and_helpers_are_defined():
    mine = binding_methods()
    rr = [m for m in mine if m[0] == "rrule"]
    assert sorted(len(m[1]) for m in rr) == [6, 7]                       # typeof(loss) + 5 (unconditioned) / + 6 (conditioned: ys)
    cond = [m for m in rr if len(m[1]) == 7][0]
    assert norm(cond[1][4]) == "AbstractMatrix{<:Real}"                  # ys, as src/core/icnf.jl:639-649
    text = "\n".join(open(f).read() for f in JL_FILES)
    for helper in ("cached_handle", "fixed_step_args", "bind_params!", "param_offsets", "cnf_config", "conditions_of",
                   "hip_aug_f!", "hip_loss_and_gradient", "fixed_dt_grid", "is_std_normal", "DeviceArg", "finish!", "cnf_check"):
        assert re.search(rf"^(function\s+)?{re.escape(helper)}\(", text, flags=re.M) or re.search(rf"^(mutable\s+)?struct {helper}\b", text, flags=re.M), helper
    assert "icnf_inplace_view" not in text                                # the r1 sketch's undefined helper is gone, not renamed


def test_gradient_rule_refuses_a_base_distribution_the_kernels_do_not_fuse():
    """the reverse-sweep kernels hard-code the terminal costate of MvNormal(0, I); `inference_sol` falls back for
    another `basedist`, so the rrule's body must not return a silently wrong gradient - it errors before any ccall."""
    text = open(os.path.join(ROOT, "julia", "hip_ext", "rrule.jl")).read()
    body = text[text.index("function hip_loss_and_gradient"):]
    body = body[:body.index("\nend\n")]
    guard, first_ccall = body.find("is_std_normal(icnf.basedist) || error("), body.find("ccall(")
    assert 0 <= guard < first_ccall and guard < body.find("cached_handle(")
    # and parameters are bound on every call (no content signature that an in-place update can defeat)
    h = open(os.path.join(ROOT, "julia", "hip_ext", "handle.jl")).read()
    bind = h[h.index("function bind_params!"):]
    bind = bind[:bind.index("\nend\n")]
    assert "return nothing" not in bind[:bind.index("ccall(")] and "sum(p)" not in bind


def test_reference_golden_recipe_calls_the_reference_as_it_is_declared():
    """julia/make_reference_golden.jl cannot be executed here (no Julia): as text, every `CNF.<function>(...)` call in it has the
    arity of a method the reference declares for that function (tests/golden/reference_signatures.json), the `ICNF(; ...)` call
    uses only keywords of the reference's constructor (src/core/icnf.jl:53-103), the compute modes / modes it names exist
    (src/core/types.jl), and the header states the NPZ.jl return types the script relies on."""
    from jl_signatures import _balanced, split_top, strip_comments
    text = strip_comments(open(os.path.join(ROOT, "julia", "make_reference_golden.jl")).read())
    table = reference_table()
    arities = {}
    for m in table:
        arities.setdefault(m["function"], set()).add(len(m["args"]))
    seen = {}
    for m in re.finditer(r"CNF\.(\w+)\(", text):
        name = m.group(1)
        end = _balanced(text, m.end() - 1)
        args = split_top(text[m.end():end - 1].replace("\n", " "))
        kw = [a for a in args if a.startswith(";") or re.match(r"^\w+\s*=[^=]", a)]
        pos = [a for a in args if a not in kw]
        seen.setdefault(name, []).append((len(pos), args))
    for fn in ("augmented_f", "inference", "inference_prob", "base_sol", "add_conditions_nn"):
        assert fn in seen, fn
        for n, args in seen[fn]:
            assert n in arities[fn], (fn, n, sorted(arities[fn]), args)
    # conditioned and unconditioned forms are both exercised
    assert {n for n, _ in seen["inference"]} == {5, 6} and {n for n, _ in seen["add_conditions_nn"]} == {1, 2}
    # the out-of-place dynamics call: (u, p, t, icnf, mode, nn, st, eps) - 8 positional arguments (src/core/icnf.jl:517-536)
    assert {n for n, _ in seen["augmented_f"]} == {8}
    # constructor keywords
    kws = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constructor_keywords.json")))
    m = re.search(r"CNF\.ICNF\(;", text)
    end = _balanced(text, m.end() - 2)
    used = []
    for a in split_top(text[m.end():end - 1].replace("\n", " ")):
        k = re.match(r"\s*([^\s=,]+)", a).group(1)
        used.append(k)
    assert used and all(k in kws for k in used), [k for k in used if k not in kws]
    for must in ("nvariables", "naugments", "nconditions", "autonomous", "nn", "compute_mode", "steer_rate", "epsdist", "sol_kwargs"):
        assert must in used, must
    # types named by the recipe exist in the reference (names only)
    for tname in ("LuxVecJacMatrixMode", "LuxJacVecMatrixMode", "TestMode", "TrainMode"):
        assert f"CNF.{tname}" in text
    ref_root = os.environ.get("CNF_REFERENCE", "/root/reference")
    if os.path.isdir(os.path.join(ref_root, "src", "core")):
        types_jl = open(os.path.join(ref_root, "src", "core", "types.jl")).read()
        for tname in ("LuxVecJacMatrixMode", "LuxJacVecMatrixMode", "TestMode", "TrainMode"):
            assert re.search(rf"struct {tname}\b", types_jl), tname
    # the header says what it assumes of NPZ.jl
    raw = open(os.path.join(ROOT, "julia", "make_reference_golden.jl")).read()
    assert "NPZ.jl return types assumed" in raw


def test_reference_signature_fixture_is_current():
    ref_root = os.environ.get("CNF_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "src", "core")):
        pytest.skip("reference tree not present (GPU box): the committed fixture is used as is")
    import importlib.util
    spec = importlib.util.spec_from_file_location("mrs", os.path.join(ROOT, "tests", "golden", "make_reference_signatures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.derive() == reference_table()
    assert mod.constructor_keywords() == json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constructor_keywords.json")))


def test_julia_environment_carries_the_references_bounds():
    """julia/Project.toml: the compat bounds INTEGRATION.md states in prose, as a file Pkg can resolve - every
    entry is the reference's own UUID and bound (checked against /root/reference/Project.toml where that tree exists), and every
    package the recipe and the binding import is either listed, a standard library, the reference itself, or one of the four that
    julia/setup_env.jl adds by name."""
    import tomli
    root = os.path.dirname(JL_DIR)
    proj = tomli.load(open(os.path.join(root, "Project.toml"), "rb"))
    deps, compat = proj["deps"], proj["compat"]
    assert set(deps) | {"julia"} == set(compat) and compat["julia"] == "1.10"
    ref = "/root/reference/Project.toml"
    if os.path.exists(ref):
        r = tomli.load(open(ref, "rb"))
        for name, uuid in deps.items():
            assert r["deps"][name] == uuid and r["compat"][name] == compat[name], name
        assert r["compat"]["julia"] == compat["julia"]
    imported = set()
    for f in [os.path.join(root, "make_reference_golden.jl")] + [os.path.join(JL_DIR, n) for n in os.listdir(JL_DIR) if n.endswith(".jl")]:
        for line in open(f):
            m = re.match(r"\s*(?:import|using)\s+([A-Za-z0-9_., :]+)", line)
            if m:
                for part in m.group(1).split(","):
                    name = part.strip().split(":")[0].split(".")[0].strip()
                    if name:
                        imported.add(name)
    by_name = {"OrdinaryDiffEqTsit5", "OrdinaryDiffEqLowOrderRK", "NPZ", "JSON"}
    setup = open(os.path.join(root, "setup_env.jl")).read()
    assert all(n in setup for n in by_name)
    known = set(deps) | by_name | {"ContinuousNormalizingFlows", "Libdl", "LinearAlgebra", "Random", "Pkg", "CNF"}
    optional = {"AMDGPU", "MPI"}            # weak dependencies of the binding's device / multi-process glue, loaded by the user
    assert imported - known - optional == set(), imported - known - optional
