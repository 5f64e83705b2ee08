"""Every `file:line` citation of a reference file in the documentation, the C-ABI header and the product sources
must point inside that file (skipped where /root/reference is absent, e.g. on the GPU box)."""
import os
import re

import pytest

from conftest import ROOT

REF = "/root/reference"
DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md", "BASELINE.md", "include/zkwg.h"]
SRC_DIRS = ["zk-email-verify_amd/csrc", "zk-email-verify_amd/py/zkwg", "zk-email-verify_amd/js", "oracle/pyref", "oracle/c"]
CITE = re.compile(r"([A-Za-z0-9_./@-]+\.(?:circom|ts|md|yml|sol|json)):(\d+)(?:-(\d+))?")


@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is not present")
def test_cited_reference_lines_exist():
    index = {}
    for d, _, files in os.walk(REF):
        if "/node_modules" in d or "/.git" in d:
            continue
        for f in files:
            index.setdefault(f, []).append(os.path.join(d, f))
    lengths = {}

    def nlines(p):
        if p not in lengths:
            with open(p, "rb") as fh:
                lengths[p] = fh.read().count(b"\n") + 1
        return lengths[p]

    files = [os.path.join(ROOT, d) for d in DOCS]
    for sd in SRC_DIRS:
        for d, _, fs in os.walk(os.path.join(ROOT, sd)):
            files += [os.path.join(d, f) for f in fs if f.endswith((".h", ".hip", ".py", ".js", ".c"))]
    bad, checked = [], 0
    for path in files:
        text = open(path, errors="replace").read()
        for m in CITE.finditer(text):
            name, lo, hi = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            base = os.path.basename(name)
            cands = [p for p in index.get(base, []) if p.endswith(name.lstrip("./")) or "/" not in name]
            if not cands:
                continue                       # not a reference file (our own file, or an [EXT] package)
            checked += 1
            if not any(lo <= hi <= nlines(p) for p in cands):
                bad.append(f"{os.path.relpath(path, ROOT)}: {m.group(0)}")
    assert checked > 200
    assert not bad, bad[:20]
