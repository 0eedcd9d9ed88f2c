"""Derive the argument-type table of the reference methods the HIP binding specialises (names and `file:line` only — metadata,
not source) from /root/reference, into tests/golden/reference_signatures.json.  tests/test_julia_binding.py checks the binding's
methods against this table for dispatch ambiguity, and re-derives it when the reference tree is present.
Run from the repo root:  python tests/golden/make_reference_signatures.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from jl_signatures import methods  # noqa: E402

REF = os.environ.get("CNF_REFERENCE", "/root/reference")
NAMES = ("augmented_f", "base_sol", "inference_sol", "generate_sol", "make_ode_func", "loss", "inference_prob", "rrule",
         # called by julia/make_reference_golden.jl (arity / argument-kind check in tests/test_julia_binding.py)
         "inference", "add_conditions_nn")
FILES = ("src/core/icnf.jl", "src/core/base_icnf.jl", "src/core/utils.jl")


def derive():
    table = []
    for f in FILES:
        text = open(os.path.join(REF, f)).read()
        for name, types, line in methods(text, NAMES):
            table.append({"function": name, "args": types, "where": f"{f}:{line}"})
    return table


def constructor_keywords():
    """Keyword names of the reference's `ICNF(; ...)` constructor (src/core/icnf.jl:53-103) - names only."""
    import re
    text = open(os.path.join(REF, "src/core/icnf.jl")).read()
    start = text.index("function ICNF(;")
    depth, i = 0, text.index("(", start)
    j = i
    while True:
        c = text[j]
        depth += c == "("
        depth -= c == ")"
        if depth == 0:
            break
        j += 1
    from jl_signatures import split_top
    kws = []
    for a in split_top(text[i + 2:j].replace("\n", " ")):
        m = re.match(r"\s*([^\s:=]+)\s*(::|=)", a)
        if m:
            kws.append(m.group(1))
    return kws


if __name__ == "__main__":
    t = derive()
    json.dump(t, open(os.path.join(HERE, "reference_signatures.json"), "w"), indent=0)
    json.dump(constructor_keywords(), open(os.path.join(HERE, "reference_constructor_keywords.json"), "w"), indent=0)
    print(len(t), "methods")
