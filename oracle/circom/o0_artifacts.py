"""`circom --O0`-style artefacts of a reference circuit from the interpreter (TEST INFRASTRUCTURE):
`.r1cs` (iden3 binary, every `<==` / `===` one constraint), `.sym` (labelIdx,witnessIdx,componentIdx,name),
the rename rules between the compiler's anonymous-component names and the kept-v1 names of the product, the
input used, and the complete witness (digest + values).  The product loads the `.sym` + `.r1cs` pair through
zkwg_circuit_create_full; tests compare its complete witness with the interpreter's.

    python -m oracle.circom.o0_artifacts rsa  OUT_DIR
    python -m oracle.circom.o0_artifacts ev   OUT_DIR [maxHeader maxBody]
"""
import gzip
import hashlib
import json
import os
import re
import sys

from . import ev
from .compare import KEPT_V1_ANON, _ANON_RE, _anon_sites
from .runtime import iter_signals, write_sym
from .symbolic import SymProgram


def alias_rules(root, sources):
    """[(ours, theirs)] deepest first: full kept-v1 path of an anonymous component -> the same path with the
    compiler's name for that component (the parents are renamed by their own, later rules)."""
    rules = []

    def visit(inst, kpath, depth):
        sites = {}
        for c in inst.subs:
            m = _ANON_RE.match(c.name)
            if m:
                sites.setdefault(m.group(1), set()).add(int(m.group(3)))
        if sites and inst.tname in sources:
            for t, off in _anon_sites(sources[inst.tname][1]):
                sites.setdefault(t, set()).add(off)
        order = {t: sorted(v) for t, v in sites.items()}
        seen = set()
        for c in inst.subs:
            m = _ANON_RE.match(c.name)
            if m:
                t, off, idx = m.group(1), int(m.group(3)), m.group(4)
                ktok = KEPT_V1_ANON.get((inst.tname, t, order[t].index(off)), "anon_" + t)
                ttok = c.name[:len(c.name) - len(idx)] if idx else c.name
                end = "[" if idx else "."
                key = (kpath + "." + ktok + end, kpath + "." + ttok + end)
                if key not in seen:
                    seen.add(key)
                    rules.append((depth, key))
                kname = ktok + idx
            else:
                kname = c.name
            visit(c, kpath + "." + kname, depth + 1)
    visit(root, "main", 0)
    rules.sort(key=lambda r: -r[0])
    return [k for _, k in rules]


def build(kind, out_dir, max_header=576, max_body=192, inputs=None, compress=True):
    os.makedirs(out_dir, exist_ok=True)
    paths = ev.include_paths()
    if kind == "rsa":
        prog = SymProgram(os.path.join(ev.REF_CIRCUITS, "tests/test-circuits/rsa-test.circom"), paths)
        tag = "rsa"
    elif kind == "ev":
        prog = SymProgram(None, paths)
        prog.load(os.path.join(ev.REF_CIRCUITS, "email-verifier.circom"))
        prog.main = (["pubkey"], "EmailVerifier", [("num", v) for v in (max_header, max_body, 121, 17, 0, 0, 0, 0)])
        tag = f"ev_{max_header}_{max_body}"
    else:
        raise SystemExit("kind must be rsa | ev")
    root = prog.run(inputs)
    base = os.path.join(out_dir, "o0_" + tag)
    n_wires = prog.write_r1cs(base + ".r1cs")
    with open(base + ".sym", "w") as fh:
        write_sym(root, fh)
    h = hashlib.sha256()
    h.update((1).to_bytes(32, "little"))
    sample = {}
    for i, (_, v, _, _) in enumerate(iter_signals(root, with_names=False), start=1):
        h.update(v.to_bytes(32, "little"))
        if i % 9973 == 0:
            sample[str(i)] = str(v)
    rules = alias_rules(root, prog.templates_src)
    meta = {"kind": kind, "max_header": max_header, "max_body": max_body, "n_wires": n_wires,
            "n_constraints": prog.n_constraints, "witness_sha256": h.hexdigest(), "sample": sample,
            "alias": "".join(f"{a}={b}\n" for a, b in rules),
            "inputs": {k: [str(x) for x in v] if isinstance(v, list) else str(v) for k, v in inputs.items()}}
    json.dump(meta, open(base + ".json", "w"))
    if compress:
        for ext in (".r1cs", ".sym"):
            with open(base + ext, "rb") as src, gzip.open(base + ext + ".gz", "wb", compresslevel=6) as dst:
                while True:
                    b = src.read(1 << 24)
                    if not b:
                        break
                    dst.write(b)
            os.unlink(base + ext)
    return meta


def default_inputs(kind, max_header=576, max_body=192):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path[:0] = [os.path.join(root, "tests"), os.path.join(root, "zk-email-verify_amd", "py")]
    if kind == "rsa":
        from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG, limbs
        return {"message": KAT_MSG, "signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB)}
    from test_ev_cpu import _inputs
    return _inputs(max_header, max_body, 0, index=0, body_len=60)


if __name__ == "__main__":
    kind, out = sys.argv[1], sys.argv[2]
    n, m = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (576, 192)
    meta = build(kind, out, n, m, default_inputs(kind, n, m))
    print({k: v for k, v in meta.items() if k not in ("sample", "alias", "inputs")})
