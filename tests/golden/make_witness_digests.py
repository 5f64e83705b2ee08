"""Generates tests/golden/witness_digests.json: SHA-256 digests of whole kept-v1 witnesses computed by the
LITERAL Python oracle (oracle/pyref) for fixed synthetic emails and the reference's RSA KAT.  The fast C
oracle and the HIP path are then checked against these committed digests (tests/test_golden.py), so a
silent drift of either is caught even where the literal oracle is too slow to run.

Run (build container, ~2 min):  python tests/golden/make_witness_digests.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.pyref import zkemail as zk, comp  # noqa: E402
from zkwg import synth, inputs  # noqa: E402


def digest(ws):
    h = hashlib.sha256()
    for v in ws:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


out = {"layout": "kept-v1", "cases": []}
for (N, M, ignore, seed, index, body_len) in [(576, 192, 0, 7, 0, 100), (576, 192, 1, 7, 1, 100), (1024, 1536, 0, 7, 2, 1024)]:
    d = synth.synthetic_dkim_result(seed, index, body_len=body_len)
    inp = inputs.generate_email_verifier_inputs_from_dkim_result(d, N, M, ignore_body_hash_check=bool(ignore))
    main = zk.EmailVerifier(N, M, 121, 17, ignore, inp, body_hash_regex=lambda m: zk.BodyHashRegexV1(N, m))
    w = comp.witness_kept(main)
    out["cases"].append({"main": "EmailVerifier", "max_header": N, "max_body": M, "ignore_body_hash_check": ignore,
                         "seed": seed, "index": index, "body_len": body_len, "W": len(w), "sha256": digest(w),
                         "pubkeyHash": str(w[1]), "shaHi": str(w[2]), "shaLo": str(w[3])})
    print(out["cases"][-1])
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "witness_digests.json"), "w"), indent=1)
