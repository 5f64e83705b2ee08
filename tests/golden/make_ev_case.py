"""tests/golden/ev_576_192_case.json: one synthetic email for EmailVerifier(576,192,121,17,0,0,0,0) as the
`CircuitInput` object generateEmailVerifierInputs returns (decimal strings), with the SHA-256 of the
kept-v1 witness bytes computed by the literal Python oracle (oracle/pyref) and its public signals.
Used by zk-email-verify_amd/js/test_gpu.js (Node -> N-API -> C-ABI -> HIP) and tests/test_generic_inputs.py.

    python tests/golden/make_ev_case.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zk-email-verify_amd", "py")]


def main():
    from oracle.pyref import zkemail as zk, comp
    from test_ev_cpu import _inputs
    N, M = 576, 192
    inp = _inputs(N, M, 0, index=3, body_len=75)
    main = zk.EmailVerifier(N, M, 121, 17, 0, inp, body_hash_regex=lambda m: zk.BodyHashRegexV1(N, m))
    wit = comp.witness_kept(main)
    blob = b"".join(int(v).to_bytes(32, "little") for v in wit)
    case = {"maxHeader": N, "maxBody": M,
            "input": {k: [str(x) for x in v] if isinstance(v, list) else str(v) for k, v in inp.items()},
            "witnessLen": len(wit), "witnessSha256": hashlib.sha256(blob).hexdigest(),
            "pubkeyHash": str(wit[1]), "shaHi": str(wit[2]), "shaLo": str(wit[3])}
    dst = os.path.join(ROOT, "tests", "golden", "ev_576_192_case.json")
    json.dump(case, open(dst, "w"))
    print("wrote", dst, os.path.getsize(dst), "bytes; W =", len(wit))


if __name__ == "__main__":
    main()
