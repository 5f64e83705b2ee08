"""Helpers shared by the tests that compare the interpreter with the literal Python oracle
(oracle/pyref): both enumerate every declared signal in O0 order; multi-dimensional circom
signals are compared under pyref's flattened naming (`a[t][k]` -> `a[32 t + k]`)."""
import itertools


def flat_walk(root):
    """(flat_name, value, how, kind) for every signal of an interpreter run, O0 order."""
    stack = [(root, "main")]
    while stack:
        inst, path = stack.pop()
        sigs = list(inst.sigs.values())
        outs = [s for s in sigs if s.kind == "out"]
        ins = [s for s in sigs if s.kind == "in"]
        mids = [s for s in sigs if s.kind == "mid"]
        if inst.public:
            ins = [s for s in ins if s.name in inst.public] + [s for s in ins if s.name not in inst.public]
        for s in outs + ins + mids:
            how = s.how
            if not s.dims:
                yield f"{path}.{s.name}", (s.vals[0] or 0), (how[0] if how else None), s.kind
            else:
                base = f"{path}.{s.name}["
                for j, v in enumerate(s.vals):
                    yield f"{base}{j}]", (v or 0), (how[j] if how else None), s.kind
        for c in reversed(inst.subs):
            stack.append((c, path + "." + c.name))


def diff(root, pyref_main, rename=None, limit=10):
    """-> (n_compared, [first differences]) between flat_walk(root) and pyref_main.walk()."""
    n = 0
    bad = []
    nbad = 0
    for a, b in itertools.zip_longest(flat_walk(root), pyref_main.walk()):
        n += 1
        an = a[0] if a else None
        if rename and an is not None:
            an = rename(an)
        if a is None or b is None or an != b[0] or a[1] != b[1]:
            nbad += 1
            if len(bad) < limit:
                bad.append((n, a, b))
    return n, nbad, bad


# ------------------------------------------------------------------ anonymous component names
# The compiler names an anonymous component `<Template>_<line>_<offset>` (+ `[i]` inside a loop);
# the kept-v1 layout (oracle/pyref, csrc/zkwg_layout.h) uses stable names instead.  The table maps
# (parent template, anonymous template, ordinal of the call site inside the parent by source
# position) -> kept-v1 name; default `anon_<Template>`.
import re

KEPT_V1_ANON = {
    ("EmailVerifier", "AssertZeroPadding", 0): "anon_AssertZeroPadding_header",
    ("EmailVerifier", "AssertZeroPadding", 1): "anon_AssertZeroPadding_body",
    ("SelectRegexReveal", "IsZero", 0): "anon_IsZero",
    ("SelectRegexReveal", "IsZero", 1): "anon_IsPrevZero",
    ("PoseidonModular", "Slice", 0): "anon_Slice",
    ("PoseidonModular", "Slice", 1): "anon_Slice",
    ("PoseidonModular", "Poseidon", 0): "anon_Poseidon_chunk",
    ("PoseidonModular", "Poseidon", 1): "anon_Poseidon_chunk",
    ("PoseidonModular", "Poseidon", 2): "anon_Poseidon_merge",
}
_ANON_RE = re.compile(r"^([A-Za-z0-9_]+?)_(\d+)_(\d+)((?:\[\d+\])*)$")


def anon_aliases(root, sources=None):
    """-> {interpreter component path prefix: kept-v1 path prefix} for every anonymous component.
    `sources`: the program's templates_src (to enumerate call sites that did not execute)."""
    out = {}

    def visit(inst, ipath, kpath):
        sites = {}
        for c in inst.subs:
            m = _ANON_RE.match(c.name)
            if m:
                sites.setdefault(m.group(1), set()).add(int(m.group(3)))
        if sources is not None and inst.tname in sources:
            for t, off in _anon_sites(sources[inst.tname][1]):
                sites.setdefault(t, set()).add(off)
        order = {t: sorted(v) for t, v in sites.items()}
        for c in inst.subs:
            m = _ANON_RE.match(c.name)
            if m:
                t, off, idx = m.group(1), int(m.group(3)), m.group(4)
                kname = KEPT_V1_ANON.get((inst.tname, t, order[t].index(off)), "anon_" + t) + idx
                out[ipath + "." + c.name + "."] = kpath + "." + kname + "."
            else:
                kname = c.name
            visit(c, ipath + "." + c.name, kpath + "." + kname)
    visit(root, "main", "main")
    return out


def _anon_sites(node):
    if type(node) is tuple:
        if node and node[0] == "anon":
            yield node[1], node[5]
        for x in node:
            yield from _anon_sites(x)
    elif type(node) is list:
        for x in node:
            yield from _anon_sites(x)


def flat_walk_kept(root, sources=None):
    """flat_walk with kept-v1 component names: (kept_name, value, how, kind), O0 order."""
    stack = [(root, "main")]
    while stack:
        inst, path = stack.pop()
        sigs = list(inst.sigs.values())
        outs = [s for s in sigs if s.kind == "out"]
        ins = [s for s in sigs if s.kind == "in"]
        mids = [s for s in sigs if s.kind == "mid"]
        if inst.public:
            ins = [s for s in ins if s.name in inst.public] + [s for s in ins if s.name not in inst.public]
        for s in outs + ins + mids:
            how = s.how
            if not s.dims:
                yield f"{path}.{s.name}", (s.vals[0] or 0), (how[0] if how else None), s.kind
            else:
                base = f"{path}.{s.name}["
                for j, v in enumerate(s.vals):
                    yield f"{base}{j}]", (v or 0), (how[j] if how else None), s.kind
        sites = {}
        for c in inst.subs:
            m = _ANON_RE.match(c.name)
            if m:
                sites.setdefault(m.group(1), set()).add(int(m.group(3)))
        if sites and sources is not None and inst.tname in sources:
            for t, off in _anon_sites(sources[inst.tname][1]):
                sites.setdefault(t, set()).add(off)
        order = {t: sorted(v) for t, v in sites.items()}
        for c in reversed(inst.subs):
            m = _ANON_RE.match(c.name) if sites else None
            if m:
                t, off, idx = m.group(1), int(m.group(3)), m.group(4)
                kname = KEPT_V1_ANON.get((inst.tname, t, order[t].index(off)), "anon_" + t) + idx
            else:
                kname = c.name
            stack.append((c, path + "." + kname))
