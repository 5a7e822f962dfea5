#!/usr/bin/env python3
"""Layout probe for the drop-in boundary.

Parses the public headers under include/libplacebo for every struct / union
definition and its direct members, emits one C program that prints
`sizeof` of each aggregate and `offsetof` + size of each member, and compiles +
runs it against a chosen include root. Running it once against include/ and
once against the reference's src/include (+ the generated config.h under
oracle/_ref/gen) gives two tables that must be equal for the structs that
cross the C ABI to be layout-compatible with libplacebo's.

    tools/abi_probe.py ours            # table for include/
    tools/abi_probe.py ref             # table for /root/reference (needs oracle/_ref/gen)
    tools/abi_probe.py golden          # rewrite tests/golden/abi_layout.json from the reference

Test infrastructure; the product does not use it.
"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "include")
REF = os.environ.get("PL_REFERENCE", "/root/reference")
GOLDEN = os.path.join(ROOT, "tests", "golden", "abi_layout.json")

# headers that only exist on this backend (no reference counterpart)
OWN_HEADERS = {"libplacebo/hip.h"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def split_members(body):
    """Top-level ';'-separated declarations of a struct body (nested braces kept intact)."""
    out, depth, cur = [], 0, ""
    for ch in body:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        if ch == ";" and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    return [m for m in out if m]


def member_names(decl):
    """Names declared by one member declaration (`int a, b[4];`, `void (*cb)(void *);`, ...)."""
    decl = re.sub(r"\{.*\}", "", decl, flags=re.S)  # anonymous/nested aggregate body
    m = re.search(r"\(\s*\*\s*(\w+)\s*\)\s*\(", decl)
    if m:
        return [m.group(1)]
    names = []
    parts = decl.split(",")
    for i, part in enumerate(parts):
        part = re.sub(r"\[[^\]]*\]", "", part).strip()
        part = re.sub(r":\s*\d+$", "", part)
        ids = re.findall(r"[A-Za-z_]\w*", part)
        if not ids:
            continue
        if i == 0 and len(ids) < 2:
            continue  # anonymous member (nested union without a name)
        names.append(ids[-1])
    return names


def parse_header(path):
    text = strip_comments(open(path).read())
    # drop preprocessor lines (macros may contain braces)
    text = re.sub(r"^[ \t]*#.*?(?<!\\)$", "", text, flags=re.M | re.S)
    aggs = []
    for m in re.finditer(r"\b(struct|union)\s+(\w+)\s*\{", text):
        # only top-level definitions
        start = m.end()
        depth, i = 1, start
        while depth and i < len(text):
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        before = text[:m.start()]
        if before.count("{") != before.count("}"):
            continue  # nested definition
        body = text[start:i - 1]
        fields = []
        for decl in split_members(body):
            fields += member_names(decl)
        aggs.append((m.group(1), m.group(2), fields))
    return aggs


def our_headers():
    hs = sorted(glob.glob(os.path.join(OURS, "libplacebo", "**", "*.h"), recursive=True))
    return [os.path.relpath(h, OURS) for h in hs]


def gen_program(headers, aggs):
    lines = ["#include <stdio.h>", "#include <stddef.h>"]
    lines += [f"#include <{h}>" for h in headers]
    lines.append("#define FSZ(t, f) sizeof(((t *) 0)->f)")
    lines.append("int main(void) {")
    for kind, name, fields in aggs:
        t = f"{kind} {name}"
        lines.append(f'    printf("S %s %zu\\n", "{t}", sizeof({t}));')
        for f in fields:
            lines.append(f'    printf("F %s.%s %zu %zu\\n", "{t}", "{f}", '
                         f'offsetof({t}, {f}), FSZ({t}, {f}));')
    lines += ["    return 0;", "}"]
    return "\n".join(lines) + "\n"


def aggregates(which):
    """{(kind, name): [fields]} of every aggregate the chosen header set defines, restricted
    to the headers that exist on this backend."""
    root = OURS if which == "ours" else os.path.join(REF, "src", "include")
    out = {}
    for h in our_headers():
        if h in OWN_HEADERS and which != "ours":
            continue
        path = os.path.join(root, h)
        if os.path.exists(path):
            for kind, name, fields in parse_header(path):
                out[(kind, name)] = fields
    return out


def member_diff():
    """Members the reference declares that include/ lacks, and the other way round, for
    every aggregate both define."""
    a, b = aggregates("ours"), aggregates("ref")
    missing = {f"{k[0]} {k[1]}": [f for f in b[k] if f not in a[k]] for k in a if k in b}
    extra = {f"{k[0]} {k[1]}": [f for f in a[k] if f not in b[k]] for k in a if k in b}
    only_ours = sorted(f"{k[0]} {k[1]}" for k in a if k not in b)
    return ({k: v for k, v in missing.items() if v}, {k: v for k, v in extra.items() if v},
            only_ours)


def run_probe(which, aggs=None):
    """Table for `which` in ("ours", "ref"). The member list comes from `aggs`
    ({(kind, name): fields}); default: the reference's members of the aggregates both
    header sets define (so a member include/ lacks is a compile error = a finding)."""
    headers = [h for h in our_headers() if which == "ours" or h not in OWN_HEADERS]
    if aggs is None:
        ours = aggregates("ours")
        aggs = {k: v for k, v in aggregates("ref").items() if k in ours}
    aggs = [(k[0], k[1], v) for k, v in aggs.items()]
    prog = gen_program(headers, aggs)
    if which == "ours":
        inc = ["-I", OURS]
    else:
        inc = ["-I", os.path.join(ROOT, "oracle", "_ref", "gen"), "-I",
               os.path.join(REF, "src", "include")]
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.c")
        exe = os.path.join(td, "probe")
        open(src, "w").write(prog)
        r = subprocess.run(["gcc", "-std=c11", "-w", *inc, src, "-o", exe],
                           capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-6000:])
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    table = {}
    for ln in out.splitlines():
        p = ln.split()
        if p[0] == "S":
            table[" ".join(p[1:-1])] = int(p[-1])
        else:
            table[" ".join(p[1:-2])] = [int(p[-2]), int(p[-1])]
    return table


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "ours"
    if which == "golden":
        t = run_probe("ref")
        json.dump(t, open(GOLDEN, "w"), indent=0, sort_keys=True)
        print(f"wrote {GOLDEN}: {len(t)} entries")
        return
    if which == "members":
        missing, extra, only = member_diff()
        print("missing in include/:", json.dumps(missing, indent=1))
        print("extra in include/:", json.dumps(extra, indent=1))
        print("aggregates without reference counterpart:", only)
        return
    if which == "diff":
        ours = aggregates("ours")
        common = {k: [f for f in v if f in ours[k]] for k, v in aggregates("ref").items()
                  if k in ours}
        a, b = run_probe("ours", common), run_probe("ref", common)
        bad = 0
        for k in a:
            if k in b and a[k] != b[k]:
                print(f"{k}: ours {a[k]} ref {b[k]}")
                bad += 1
        print(f"{bad} mismatches of {len(a)}")
        return
    for k, v in run_probe(which).items():
        print(k, v)


if __name__ == "__main__":
    main()
