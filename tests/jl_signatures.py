"""A small reader of Julia method signatures (text only — there is no Julia toolchain here): used to derive the reference's
method table for the functions the HIP binding extends (tests/golden/make_reference_signatures.py) and to check the binding's
own methods against it (tests/test_julia_binding.py)."""
import re


def _balanced(text, start, open_ch="(", close_ch=")"):
    """index just past the parenthesis that closes the one at text[start]"""
    depth, i = 0, start
    while True:
        c = text[i]
        if c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for c in s:
        if c in "({[":
            depth += 1
        elif c in ")}]":
            depth -= 1
        if c == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += c
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def strip_comments(text):
    text = re.sub(r"#=.*?=#", "", text, flags=re.S)
    return "\n".join(l.split("#")[0] if '"' not in l else l for l in text.splitlines())


def methods(text, names):
    """[(function name, [argument type strings], line number)] of every `function name(...)` / `name(...) = ...` definition."""
    text_nc = strip_comments(text)
    out = []
    for m in re.finditer(r"^(?:function\s+)?((?:\w+\.)?(\w+))\(", text_nc, flags=re.M):
        name = m.group(2)
        if name not in names:
            continue
        is_fn = m.group(0).startswith("function")
        start = m.end() - 1
        end = _balanced(text_nc, start)
        if not is_fn and not re.match(r"\s*(where\s*\{[^}]*\}\s*)?=", text_nc[end:end + 200]):
            continue                                             # a call, not a short-form definition
        args = split_top(text_nc[start + 1:end - 1].replace("\n", " "))
        args = [a for a in args if not a.startswith(";")]
        types = []
        for a in args:
            a = re.sub(r"\s+", " ", a).strip().rstrip(",")
            if a.startswith("(") and ")::" in a:                 # destructured tuple argument
                types.append(a.split(")::", 1)[1].strip())
            elif "::" in a:
                types.append(a.split("::", 1)[1].strip())
            else:
                types.append("Any")
        out.append((name, types, text_nc[:m.start()].count("\n") + 1))
    return out


def type_params(t):
    """'ICNF{T, <:HIPMatrixMode, false}' -> ('ICNF', ['T', '<:HIPMatrixMode', 'false'])"""
    t = t.strip()
    if "{" not in t:
        return t, []
    head = t[:t.index("{")]
    inner = t[t.index("{") + 1:_balanced(t, t.index("{"), "{", "}") - 1]
    return head, split_top(inner)
