"""`python -m zkwg.generate_witness <circuit> <input.json> <witness.wtns> [device]` -- counterpart of circom's
generated `generate_witness.js` as the reference documents it (docs/zk-email-docs/UsageGuide/README.md:132-140:
`node generate_witness.js circuit.wasm input.json witness.wtns`).  <circuit> names the circuit by its template
parameters instead of a WASM file: `EmailVerifier(1024,1536,121,17,0,0,0,0)` or a JSON object / file with the
keyword arguments of zkwg.Circuit.  An input file holding a list writes <witness>_<i>.wtns per email and exits 1
if any email failed."""
import json
import os
import re
import sys

import zkwg


def circuit_kwargs(arg):
    m = re.fullmatch(r"\s*EmailVerifier\(([\d\s,]+)\)\s*", arg)
    if m:
        p = [int(x) for x in m.group(1).split(",")]
        if len(p) != 8:
            raise SystemExit("EmailVerifier takes 8 parameters")
        return dict(main_kind=zkwg.MAIN_EMAIL_VERIFIER, max_header=p[0], max_body=p[1], n=p[2], k=p[3],
                    ignore_body_hash_check=p[4], enable_header_masking=p[5], enable_body_masking=p[6],
                    remove_soft_line_breaks=p[7])
    text = open(arg).read() if os.path.exists(arg) else arg
    kw = json.loads(text)
    base = os.path.dirname(arg) if os.path.exists(arg) else "."
    if isinstance(kw.get("sym"), str) and os.path.exists(os.path.join(base, kw["sym"])):
        kw["sym"] = open(os.path.join(base, kw["sym"])).read()
    if isinstance(kw.get("r1cs"), str):
        kw["r1cs"] = open(os.path.join(base, kw["r1cs"]), "rb").read()
    return kw


def main(argv=None):
    a = sys.argv[1:] if argv is None else argv
    if len(a) < 3:
        print("Usage: python -m zkwg.generate_witness <circuit> <input.json> <witness.wtns> [device]", file=sys.stderr)
        return 2
    c = zkwg.Circuit(device=int(a[3]) if len(a) > 3 else 0, **circuit_kwargs(a[0]))
    inp = json.load(open(a[1]))
    many = isinstance(inp, list)
    wits, status = c.calculate_batch_host(b"".join(c.pack(x) for x in (inp if many else [inp])))
    wb = c.witness_bytes
    stem = re.sub(r"\.wtns$", "", a[2])
    failed = 0
    for i, st in enumerate(status):
        if st != 0:
            print(f"email {i}: Error: Assert Failed (status {st})", file=sys.stderr)
            failed += 1
            continue
        with open(f"{stem}_{i}.wtns" if many else a[2], "wb") as f:
            f.write(c.wtns(wits[i * wb:(i + 1) * wb]))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
