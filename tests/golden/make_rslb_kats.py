#!/usr/bin/env python3
"""Extracts the RemoveSoftLineBreaks(32) known answers from the reference's own test
(packages/circuits/tests/remove-soft-line-breaks.test.ts) into tests/golden/rslb_kats.json.
Run in the build container only (the reference tree does not exist on the GPU box)."""
import json, re, sys

src = open("/root/reference/packages/circuits/tests/remove-soft-line-breaks.test.ts").read()
cases = []
for m in re.finditer(r"it\(\s*'([^']+)'.*?encoded:\s*\[(.*?)\],\s*decoded:\s*\[(.*?)\],?\s*\};.*?isValid:\s*(\d)", src, re.S):
    def arr(txt):
        out = []
        for tok in re.split(r",(?![^()]*\))", txt):
            tok = tok.split("//")[0].strip()
            if not tok:
                continue
            f = re.match(r"\.\.\.Array\((\d+)\)\.fill\((\d+)\)", tok)
            if f:
                out += [int(f.group(2))] * int(f.group(1))
            else:
                out.append(int(tok))
        return out
    # strip line comments first
    enc = arr(re.sub(r"//[^\n]*", "", m.group(2)))
    dec = arr(re.sub(r"//[^\n]*", "", m.group(3)))
    assert len(enc) == 32 and len(dec) == 32, (m.group(1), len(enc), len(dec))
    cases.append({"name": m.group(1), "encoded": enc, "decoded": dec, "isValid": int(m.group(4))})
json.dump({"source": "packages/circuits/tests/remove-soft-line-breaks.test.ts", "maxLength": 32, "cases": cases},
          open(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/rslb_kats.json", "w"), indent=0)
print(len(cases), "cases")
