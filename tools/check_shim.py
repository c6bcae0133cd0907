#!/usr/bin/env python3
"""Static checks of the Julia shim (dril.jl_amd/julia/DRiLHIP.jl) — what stands in for running it in an image without Julia.

  1. DISPATCH.  Every method the shim adds to one of DRiL's generic functions is compared, argument by argument, with every reference method of
     the same name and positional arity (parsed from /root/reference/src).  Julia picks a method only if it is at least as specific in EVERY
     positional argument and strictly more specific in at least one; "narrower in one argument, wider in another" is `MethodError: ambiguous`
     (round 1's bug: `agent::Agent` against the reference's `Agent{<:AbstractActorCriticLayer,<:PPO,...}`).  Keyword arguments take no part.
  2. CALLBACK LOCALS.  The keys the shim puts into the `locals` Dict == the Python mirror's (dril.jl_amd/host.py) and ⊇ the keys the reference's
     test reads (test/test_callbacks.jl:25-27,36-39).
  3. C ABI.  Every `ccall((:symbol, LIB[]), ...)` names a function declared in include/*.h; the Julia mirror structs list the header structs'
     fields in the same order.
  4. ccall PROTOTYPES.  Arity, pointer-ness and scalar width / kind of every argument type and of the result type of every ccall against the C prototype.
  5. BLOCK BALANCE.  Per .jl file: block openers == `end`s.

Exit status 0 = all checks pass.  `--markdown` prints the dispatch table of INTEGRATION.md §2.  Without /root/reference (the GPU box) check 1 and the
test-file half of check 2 are skipped and said so.
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
SHIM = ROOT / "dril.jl_amd" / "julia" / "DRiLHIP.jl"


def read_shim(path: Path) -> str:
    """the shim with its `include("file.jl")` lines replaced by the files' text (DRiLHIP.jl includes its host-env, extras and SAC parts)"""
    text = path.read_text()
    def sub(m):
        f = path.parent / m.group(1)
        return f.read_text() if f.exists() else m.group(0)
    return re.sub(r'^include\("([^"]+)"\).*$', sub, text, flags=re.M)
REF = Path("/root/reference")

BUILTIN_PARENTS = {"Int": "Signed", "Int64": "Signed", "Int32": "Signed", "Signed": "Integer", "Integer": "Real", "Real": "Number", "Number": "Any",
                   "Float32": "AbstractFloat", "Float64": "AbstractFloat", "AbstractFloat": "Real", "String": "AbstractString", "AbstractString": "Any",
                   "Vector": "AbstractVector", "AbstractVector": "AbstractArray", "AbstractArray": "Any", "Symbol": "Any", "Bool": "Integer"}


# ---------------------------------------------------------------------------------------------------------------------------------
# a very small Julia type-expression parser: Name | Name{p, ...} | <:T | T where T is again a type expression
# ---------------------------------------------------------------------------------------------------------------------------------
def split_top(s: str, sep: str = ",") -> list[str]:
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{([":
            depth += 1
        elif ch in "})]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def strip_mod(name: str) -> str:
    return name.split(".")[-1]


def parse_type(s: str, aliases: dict, typevars: dict):
    """-> ("bound", T) for `<:T`, else ("type", name, params)"""
    s = s.strip()
    if s.startswith("<:"):
        return ("bound", parse_type(s[2:], aliases, typevars))
    m = re.match(r"^([\w.]+)\s*(\{(.*)\})?$", s, re.S)
    if not m:
        raise ValueError(f"cannot parse type {s!r}")
    name = strip_mod(m.group(1))
    if m.group(2) is None:
        if name in typevars:                                   # a `where` type variable: T or T <: Bound
            return ("bound", parse_type(typevars[name], aliases, {}))
        if name in aliases:
            return parse_type(aliases[name], aliases, typevars)
        return ("type", name, None)
    params = [parse_type(p, aliases, typevars) for p in split_top(m.group(3))]
    return ("type", name, params)


def name_sub(a: str, b: str, parents: dict) -> bool:
    seen = set()
    while True:
        if a == b or b == "Any":
            return True
        if a in seen or a not in parents:
            return False
        seen.add(a); a = parents[a]


def issub(a, b, parents) -> bool:
    """a ⊆ b for the shapes this tool meets (invariant parameters, `<:` bounds)"""
    if b[0] == "bound":
        inner = a[1] if a[0] == "bound" else a
        return issub(inner, b[1], parents)
    if a[0] == "bound":
        return b[1] == "Any"                                    # a set of types is inside one concrete type only if that type is Any
    if not name_sub(a[1], b[1], parents):
        return False
    if b[2] is None:
        return True                                             # b = the whole family Name{...}
    if a[2] is None or a[1] != b[1] or len(a[2]) != len(b[2]):
        return False
    for pa, pb in zip(a[2], b[2]):
        if pb[0] == "bound":
            if not issub(pa, pb, parents):
                return False
        else:                                                   # invariant parameter: must be the same type
            if pa[0] == "bound" or not (issub(pa, pb, parents) and issub(pb, pa, parents)):
                return False
    return True


# ---------------------------------------------------------------------------------------------------------------------------------
# source scanning
# ---------------------------------------------------------------------------------------------------------------------------------
def julia_sources(root: Path) -> str:
    return "\n".join(p.read_text() for p in sorted(root.rglob("*.jl")))


def type_parents(src: str) -> dict:
    parents = dict(BUILTIN_PARENTS)
    for m in re.finditer(r"(?:abstract type|(?:mutable\s+)?struct)\s+(\w+)\s*(\{)?", src):
        name, i = m.group(1), m.end()
        if m.group(2):                                          # skip the {...} parameter list (may span lines)
            depth = 1
            while depth and i < len(src):
                depth += src[i] == "{"; depth -= src[i] == "}"; i += 1
        rest = src[i:i + 200]
        pm = re.match(r"\s*<:\s*([\w.]+)", rest)
        parents.setdefault(name, strip_mod(pm.group(1)) if pm else "Any")
    return parents


def methods(src: str, names: set[str]) -> list[dict]:
    """function NAME(positional...; kwargs...) [where {...}] — multi-line argument lists included"""
    out = []
    for m in re.finditer(r"^\s*function\s+((?:\w+\.)?(\w+!?))\s*\(", src, re.M):
        if m.group(2) not in names:
            continue
        i, depth = m.end(), 1
        while depth and i < len(src):
            depth += src[i] in "([{"; depth -= src[i] in ")]}"; i += 1
        arglist = src[m.end():i - 1]
        wm = re.match(r"\s*where\s*(\{[^}]*\}|\w+)", src[i:i + 120])
        typevars = {}
        if wm:
            for tv in split_top(wm.group(1).strip("{}")):
                tm = re.match(r"(\w+)\s*(?:<:\s*(.+))?$", tv)
                typevars[tm.group(1)] = tm.group(2) or "Any"
        positional = split_top(split_top(re.sub(r"#[^\n]*", "", arglist), ";")[0])
        args = []
        for a in positional:
            a = a.split("=")[0].strip()
            am = re.match(r"^(\w*)\s*(?:::\s*(.+))?$", a, re.S)
            args.append((am.group(1), (am.group(2) or "Any").strip()))
        out.append({"name": m.group(2), "args": args, "typevars": typevars, "line": src.count("\n", 0, m.start()) + 1})
    return out


def aliases_of(src: str) -> dict:
    return {m.group(1): m.group(2).strip() for m in re.finditer(r"^const\s+(\w+)\s*=\s*(\w[\w.]*\{.*\})\s*$", src, re.M)}


def check_dispatch(markdown: bool) -> list[str]:
    errors = []
    shim = read_shim(SHIM)
    if not REF.exists():
        print("dispatch: /root/reference is absent here — skipped"); return errors
    ref = julia_sources(REF / "src")
    parents = type_parents(ref); parents.update({k: v for k, v in type_parents(shim).items() if k not in parents})
    al = aliases_of(shim)
    generic = set(re.findall(r"import DRiL:\s*([^\n]+(?:\n\s+[^\n]+)*)", shim)[0].replace("\n", " ").replace(" ", "").split(","))
    generic |= {"evaluate_agent", "log_stats", "save_normalization_stats", "load_normalization_stats!"}
    rows = []
    for sm in methods(shim, generic):
        s_types = [parse_type(t, al, sm["typevars"]) for _, t in sm["args"]]
        for rm in methods(ref, {sm["name"]}):
            if len(rm["args"]) != len(sm["args"]):
                continue
            try:
                r_types = [parse_type(t, {}, rm["typevars"]) for _, t in rm["args"]]
            except ValueError:
                continue                                        # Union-typed positional etc.: not a method the shim competes with
            rel = []
            for st, rt in zip(s_types, r_types):
                le, ge = issub(st, rt, parents), issub(rt, st, parents)
                rel.append("=" if le and ge else "<" if le else ">" if ge else "x")
            if "x" in rel:
                verdict = "disjoint (never both applicable)"
            elif "<" in rel and ">" in rel:
                verdict = "AMBIGUOUS"; errors.append(f"DRiLHIP.jl:{sm['line']} {sm['name']} vs reference: {rel}")
            elif "<" in rel:
                verdict = "shim wins (argument " + ", ".join(str(i + 1) for i, r in enumerate(rel) if r == "<") + ")"
            elif ">" in rel:
                verdict = "reference more specific"; errors.append(f"DRiLHIP.jl:{sm['line']} {sm['name']}: the reference method is MORE specific {rel}")
            else:
                verdict = "IDENTICAL SIGNATURE (overwrites the reference method)"; errors.append(f"DRiLHIP.jl:{sm['line']} {sm['name']} redefines a reference method")
            sig = lambda mm: ", ".join(t for _, t in mm["args"])
            if "x" not in rel:                                  # disjoint pairs are noise in the table
                rows.append((f"`{sm['name']}({sig(sm)})` :{sm['line']}", f"`{sm['name']}({sig(rm)})`", " ".join(rel), verdict))
    if markdown:
        print("| shim method (DRiLHIP.jl:line) | reference method | per-argument (shim vs reference) | dispatch |\n|---|---|---|---|")
        for r in rows:
            print("| " + " | ".join(r) + " |")
    else:
        for r in rows:
            print(f"dispatch: {r[0]:<110} {r[2]:<10} {r[3]}")
    if not any(r[0].startswith("`train!") for r in rows):
        errors.append("no train! method pair found — parser out of date?")
    return errors


def check_locals() -> list[str]:
    errors = []
    shim = read_shim(SHIM)
    tup = lambda name, text: tuple(re.findall(r":?\"?(\w+)\"?", re.search(name + r"\s*=\s*\(([^)]*)\)", text).group(1)))
    host = (ROOT / "dril.jl_amd" / "host.py").read_text()
    for name in ("TRAINING_START_LOCALS", "ROLLOUT_START_LOCALS"):
        j, p = tup(r"const " + name, shim), tup(name, host)
        if j != p:
            errors.append(f"{name}: shim {j} != python mirror {p}")
    start, roll = set(tup(r"const TRAINING_START_LOCALS", shim)), set(tup(r"const ROLLOUT_START_LOCALS", shim))
    dict_keys = [set(re.findall(r":(\w+)\s*=>", m)) for m in re.findall(r"locals\(\) = Dict\{Symbol, Any\}\(([^\n]+\n[^\n]+\n[^\n]+)", shim)]
    if len(dict_keys) < 2:
        errors.append("expected the locals() Dict of both PPO train! methods")
    for ks in dict_keys:
        if not (start | roll) <= ks:
            errors.append(f"a locals() Dict lacks {sorted((start | roll) - ks)}")
    test = REF / "test" / "test_callbacks.jl"
    if test.exists():
        t = test.read_text()
        need_start = set(re.findall(r":(\w+)", re.search(r"OnTrainingStartCheckLocalsCallback.*?keys::Vector\{Symbol\} = \[(.*?)\]", t, re.S).group(1)))
        need_roll = set(re.findall(r":(\w+)", re.search(r"first_keys::Vector\{Symbol\} = \[(.*?)\]", t, re.S).group(1)))
        if not need_start <= start:
            errors.append(f"training-start locals lack {sorted(need_start - start)} (test/test_callbacks.jl:25-27)")
        if not need_roll <= start | roll:
            errors.append(f"rollout-start locals lack {sorted(need_roll - (start | roll))} (test/test_callbacks.jl:36-39)")
        print(f"locals: test_callbacks.jl needs {sorted(need_start | need_roll)}; shim provides {sorted(start | roll)}")
    else:
        print("locals: /root/reference/test/test_callbacks.jl is absent here — only shim == python mirror was checked")
    return errors


def check_abi() -> list[str]:
    errors = []
    shim = read_shim(SHIM)
    headers = "\n".join(p.read_text() for p in sorted((ROOT / "include").glob("*.h")))
    declared = set(re.findall(r"\b(dril_\w+)\s*\(", headers))
    used = set(re.findall(r"ccall\(\(:(\w+),\s*LIB\[\]\)", shim))
    for sym in sorted(used - declared):
        errors.append(f"ccall of {sym}: not declared in include/*.h")
    print(f"abi: {len(used)} distinct ccall symbols, all declared" if not (used - declared) else f"abi: undeclared {sorted(used - declared)}")
    for cname, jname in (("dril_config", "DrilConfig"), ("dril_ppo_stats", "DrilPPOStats"), ("dril_sac_config", "DrilSacConfig"), ("dril_sac_stats", "DrilSacStats"), ("dril_eval_stats", "DrilEvalStats")):
        cm = re.search(r"typedef struct " + cname + r"\s*\{(.*?)\}\s*" + cname + ";", headers, re.S)
        jm = re.search(r"struct " + jname + r"\n(.*?)\nend", shim, re.S)
        if not cm or not jm:
            errors.append(f"struct {cname} / {jname} not found"); continue
        body = re.sub(r"/\*.*?\*/", "", cm.group(1), flags=re.S)
        cf = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if " " in decl else decl
            cf += [re.sub(r"\[.*\]", "", n).strip().lstrip("*") for n in names.split(",")]
        jf = re.findall(r"(\w+)::", jm.group(1))
        if cf != jf:
            errors.append(f"{jname} fields {jf} != {cname} fields {cf}")
    return errors


# ---------------------------------------------------------------------------------------------------------------------------------
# 4. ccall PROTOTYPES: for every `ccall((:sym, LIB[]), Ret, (T...), args...)` arity, pointer-ness and scalar width / kind of every argument and of the result
#    against the C prototype of include/*.h (done by hand once — a ccall with a wrong width does not fail, it corrupts).
# 5. BLOCK BALANCE per file: openers (function / struct / if / for / while / let / begin / do / try / module / macro / quote) == `end`s, outside strings, comments,
#    brackets (a[end], generators) and symbols — what a parser would refuse first.
# ---------------------------------------------------------------------------------------------------------------------------------
C_SCALARS = {"int32_t": ("i", 4), "uint32_t": ("i", 4), "int": ("i", 4), "unsigned": ("i", 4), "int64_t": ("i", 8), "uint64_t": ("i", 8), "size_t": ("i", 8),
             "long long": ("i", 8), "unsigned long long": ("i", 8), "float": ("f", 4), "double": ("f", 8), "uint8_t": ("i", 1), "int8_t": ("i", 1), "char": ("i", 1), "void": ("v", 0)}
JL_SCALARS = {"Int32": ("i", 4), "UInt32": ("i", 4), "Cint": ("i", 4), "Cuint": ("i", 4), "Int64": ("i", 8), "UInt64": ("i", 8), "Csize_t": ("i", 8), "Clonglong": ("i", 8),
              "Float32": ("f", 4), "Cfloat": ("f", 4), "Float64": ("f", 8), "Cdouble": ("f", 8), "UInt8": ("i", 1), "Int8": ("i", 1), "Cvoid": ("v", 0), "Nothing": ("v", 0)}


def c_class(t: str):
    """C parameter / return type -> ("ptr", element class or None) | (kind, bytes)"""
    t = re.sub(r"\b(const|struct|restrict|volatile)\b", " ", t)
    ptr = "*" in t or "[" in t
    base = re.sub(r"\[.*?\]", "", t).replace("*", " ")
    words = base.split()
    # drop the parameter name (last word) when the remaining words still form a type
    for cut in (len(words), len(words) - 1):
        name = " ".join(words[:cut])
        if name in C_SCALARS:
            return ("ptr", C_SCALARS[name]) if ptr else C_SCALARS[name]
    return ("ptr", None) if ptr else ("?", " ".join(words))     # a struct / opaque handle pointer, or something this table does not know


def jl_class(t: str):
    t = t.strip()
    if t == "Cstring":
        return ("ptr", ("i", 1))
    m = re.match(r"^(Ptr|Ref)\{(.*)\}$", t)
    if m:
        return ("ptr", JL_SCALARS.get(m.group(2).strip()))      # Ptr{Cvoid} / Ptr{SomeStruct} -> element None or ("v", 0): compatible with any pointer
    return JL_SCALARS.get(t, ("?", t))


def c_prototypes() -> dict:
    text = "\n".join(p.read_text() for p in sorted((ROOT / "include").glob("*.h")))
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S); text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"([\w \*]+?)\b(dril_\w+)\s*\(([^;{}]*?)\)\s*;", text, re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "(" in ret:
            continue
        plist = [] if params in ("", "void") else [p.strip() for p in split_top(params)]
        protos[name] = (ret, plist)
    return protos


def compatible(j, c) -> bool:
    if j[0] == "?" or c[0] == "?":
        return False
    if (j[0] == "ptr") != (c[0] == "ptr"):
        return False
    if j[0] == "ptr":
        je, ce = j[1], c[1]
        return je is None or ce is None or je[0] == "v" or ce[0] == "v" or je == ce      # element types compared only when both sides name a scalar
    return j == c


def check_ccalls() -> list[str]:
    errors, n = [], 0
    protos = c_prototypes()
    shim = read_shim(SHIM)
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\[\]\),\s*([\w{}]+),\s*\(", shim):
        sym, ret = m.group(1), m.group(2)
        # the argument-type tuple: balanced parentheses from the '(' that ends the match
        i, depth = m.end(), 1
        while depth and i < len(shim):
            depth += shim[i] in "({["; depth -= shim[i] in ")}]"; i += 1
        types = [t for t in split_top(shim[m.end():i - 1]) if t]
        line = shim.count("\n", 0, m.start()) + 1
        if sym not in protos:
            errors.append(f"ccall {sym} (line {line} of the joined shim): no prototype parsed from include/*.h"); continue
        cret, cparams = protos[sym]
        n += 1
        if len(types) != len(cparams):
            errors.append(f"ccall {sym} (line {line}): {len(types)} argument types {types}, the header declares {len(cparams)}: {cparams}"); continue
        if not compatible(jl_class(ret), c_class(cret + " _")):
            errors.append(f"ccall {sym} (line {line}): result {ret} vs C `{cret}`")
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            if not compatible(jl_class(jt), c_class(ct)):
                errors.append(f"ccall {sym} (line {line}): argument {k + 1} {jt} vs C `{ct}`")
    print(f"ccall prototypes: {n} call sites checked against {len(protos)} prototypes of include/*.h")
    if n < 40:
        errors.append(f"only {n} ccall sites parsed — parser out of date?")
    return errors


def block_balance(text: str) -> tuple[int, int]:
    """(openers, ends) at bracket depth 0 outside strings / comments / symbols"""
    text = re.sub(r'"""(?:.|\n)*?"""', '""', text)
    text = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', text)
    text = re.sub(r"#=(?:.|\n)*?=#", " ", text); text = re.sub(r"#[^\n]*", " ", text)
    text = re.sub(r"'(?:\\.|[^'\\])'", "' '", text)
    opens = ends = depth = 0
    for m in re.finditer(r"[\[\]()]|:?\b[A-Za-z_]\w*\b", text):
        tok = m.group(0)
        if tok in "[(":
            depth += 1
        elif tok in "])":
            depth -= 1
        elif depth == 0 and not tok.startswith(":"):
            prev = text[max(0, m.start() - 1):m.start()]
            if prev == ".":                                     # a field named like a keyword (x.end)
                continue
            if tok in ("function", "struct", "if", "for", "while", "let", "begin", "do", "try", "module", "macro", "quote"):
                opens += 1
            elif tok == "end":
                ends += 1
    return opens, ends


def bracket_balance(text: str) -> list[str]:
    """unbalanced (), [], {} outside strings / comments / character literals, with the line of the first offence"""
    text = re.sub(r'"""(?:.|\n)*?"""', lambda m: '""' + "\n" * m.group(0).count("\n"), text)
    text = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', text)
    text = re.sub(r"#=(?:.|\n)*?=#", lambda m: "\n" * m.group(0).count("\n"), text); text = re.sub(r"#[^\n]*", " ", text)
    text = re.sub(r"'(?:\\.|[^'\\\n])'", "' '", text)
    pairs, stack, out = {")": "(", "]": "[", "}": "{"}, [], []
    for i, ch in enumerate(text):
        if ch in "([{":
            stack.append((ch, text.count("\n", 0, i) + 1))
        elif ch in ")]}":
            if not stack or stack[-1][0] != pairs[ch]:
                out.append(f"line {text.count(chr(10), 0, i) + 1}: unexpected `{ch}`"); break
            stack.pop()
    if not out and stack:
        out.append(f"line {stack[-1][1]}: `{stack[-1][0]}` never closed")
    if text.count('"') % 2:
        out.append("an unterminated string literal")
    return out


def check_blocks() -> list[str]:
    errors = []
    for p in sorted(SHIM.parent.glob("*.jl")) + [ROOT / "tests" / "golden" / "gen_reference_golden.jl"]:
        if not p.exists():
            continue
        for msg in bracket_balance(p.read_text()):
            errors.append(f"{p.name}: {msg}")
    for p in sorted(SHIM.parent.glob("*.jl")):
        o, e = block_balance(p.read_text())
        if o != e:
            errors.append(f"{p.name}: {o} block openers but {e} `end`s")
    print("block balance: " + ", ".join(f"{p.name} {block_balance(p.read_text())[0]}" for p in sorted(SHIM.parent.glob('*.jl'))))
    return errors


def main() -> int:
    global SHIM
    if "--shim" in sys.argv:                                    # check another copy (tests mutate one to prove that the round-1 bug would be caught)
        SHIM = Path(sys.argv[sys.argv.index("--shim") + 1])
    errors = check_dispatch("--markdown" in sys.argv) + check_locals() + check_abi() + check_ccalls() + check_blocks()
    for e in errors:
        print("ERROR:", e)
    print("check_shim:", "FAILED" if errors else "ok")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
