#!/usr/bin/env python3
"""Records, from the reference's own circuit sources, WHICH names each template declares: its signals (with their kind) and its
sub-components (with the template each one instantiates).  Output: tests/golden/circom_names.json -- data, no source text.
tests/test_layout_names.py walks every stored signal of this repository's witness layout (include/hz_layout.h, the file the
product and the oracle share) down these declarations, so a mis-named signal or component is caught by something neither of
them wrote.

    python tests/golden/extract_circom_names.py [/root/reference]        (run in the build container; the reference does not travel)
"""
import json
import os
import re
import sys


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def templates(src):
    """yield (name, body) for every `template Name(args) { ... }`"""
    for m in re.finditer(r"\btemplate\s+(\w+)\s*\(([^)]*)\)\s*\{", src):
        depth, i = 1, m.end()
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        yield m.group(1), src[m.end():i - 1]


def declared(body):
    """signals {name: kind}, components {name: template}, and how many array dimensions each name was declared with"""
    signals, components, sdims, cdims = {}, {}, {}, {}
    for m in re.finditer(r"\bsignal\s+(private\s+input|input|output)?\s*([^;]+);", body):
        kind = (m.group(1) or "intermediate").replace("private input", "input").strip()
        for part in m.group(2).split(","):
            nm = re.match(r"\s*(\w+)\s*((?:\[[^\]]*\]\s*)*)", part)
            if nm:
                signals[nm.group(1)] = kind
                sdims[nm.group(1)] = nm.group(2).count("[")
    for m in re.finditer(r"\bcomponent\s+(\w+)\s*((?:\[[^\]]*\]\s*)*)(?:=\s*(\w+)\s*\()?", body):
        name, tmpl = m.group(1), m.group(3)
        if tmpl is None:   # `component x[n];` assigned later: x[i] = Template(...)
            a = re.search(r"\b%s\s*(?:\[[^\]]*\]\s*)+=\s*(\w+)\s*\(" % re.escape(name), body)
            tmpl = a.group(1) if a else None
        components[name] = tmpl
        cdims[name] = m.group(2).count("[")
    return signals, components, sdims, cdims


def classify(body, name):
    """'linear' when every definition of the intermediate signal `name` is a wire-through or a linear combination with constant
    coefficients (this repository's layout does not store those), 'product' when one multiplies two signals, 'hint' when it is
    assigned with `<--`, 'unknown' otherwise."""
    if re.search(r"\b%s\s*(?:\[[^\]]*\]\s*)*<--" % re.escape(name), body) or re.search(r"-->\s*%s\b" % re.escape(name), body):
        return "hint"   # assigned with <-- : a value the witness generator computes, constrained separately
    rhs = [m.group(1) for m in re.finditer(r"\b%s\s*(?:\[[^\]]*\]\s*)*<==\s*([^;]+);" % re.escape(name), body)]
    rhs += [m.group(1) for m in re.finditer(r"[;{}]\s*([^;{}]+?)\s*==>\s*%s\s*(?:\[[^\]]*\]\s*)*;" % re.escape(name), body)]
    if not rhs:
        return "unknown"
    for e in rhs:
        e = re.sub(r"\(\s*1\s*<<\s*\w+\s*\)", "C", e)          # (1 << 32)
        e = re.sub(r"\b\d+\b", "C", e)                              # literals
        e = re.sub(r"\b[A-Z][A-Z0-9_]*\b", "C", e)                   # CONST_SIG style constants
        for f in re.finditer(r"([\w\]\)\.]+)\s*\*\s*([\w\(\.]+)", e):
            if f.group(1) != "C" and f.group(2) != "C":
                return "product"
    return "linear"


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {}
    for root, _, files in os.walk(os.path.join(ref, "src")):
        for f in sorted(files):
            if not f.endswith(".circom"):
                continue
            rel = os.path.relpath(os.path.join(root, f), ref)
            src = strip_comments(open(os.path.join(root, f)).read())
            for name, body in templates(src):
                sig, comp, sdims, cdims = declared(body)
                out[name] = {"file": rel, "signals": sig, "components": comp, "signal_dims": sdims, "component_dims": cdims,
                             "intermediates": {n: classify(body, n) for n, k in sig.items() if k == "intermediate"}}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "circom_names.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("%d templates -> %s" % (len(out), dst))


if __name__ == "__main__":
    main()
