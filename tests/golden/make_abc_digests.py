"""Golden digests of the prover's first stage (A.w | B.w | C.w, 32-byte little-endian values: the A values, then B, then C) for the
reference's own known-answer inputs (and of the kept-v1 witness of the FpMul(2,4) one), computed WITHOUT the product's device or C++ code: the combinations of zkwg.r1cs (derived from the reference
templates) evaluated in Python integers over the oracle's witness, and -- for the entries named o0_* -- the interpreter-generated
`--O0` constraint system (the file a zkey is keyed to) evaluated over the interpreter's complete witness.  Run from the repo root: python tests/golden/make_abc_digests.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def digest(cons, w, montgomery=False):
    sc = (1 << 256) if montgomery else 1
    h = hashlib.sha256()
    for j in range(3):
        for t in cons:
            h.update((sum(cf * w[k] for k, cf in t[j].items()) * sc % P).to_bytes(32, "little"))
    return h.hexdigest()


def main():
    from oracle.pyref import zkemail as zk, comp
    from zkwg import r1cs as zr
    from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG, limbs
    out = {}
    # packages/circuits/tests/rsa.test.ts:64-103
    main_rsa = zk.RSAVerifier65537(121, 17, KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB), is_main=True)
    cons = zr.rsa_main_constraints(comp.symbols_kept(main_rsa))
    w = comp.witness_kept(main_rsa)
    out["rsa_kat"] = {"constraints": len(cons), "standard": digest(cons, w), "montgomery": digest(cons, w, True)}
    # packages/circuits/tests/fp-mul.test.ts:34-46
    fp = zk.FpMul(2, 4, [1, 0, 1, 0], [0, 1, 1, 0], [1, 1, 1, 1])
    fp.is_main = True
    cons = zr.fp_mul_main_constraints(comp.symbols_kept(fp), 2, 4)
    w = comp.witness_kept(fp)
    out["fp_mul_2_4_kat"] = {"constraints": len(cons), "standard": digest(cons, w), "montgomery": digest(cons, w, True),
                             "witness_len": len(w), "witness_sha256": hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in w)).hexdigest()}
    # ---- the systems a real prover carries: the COMPILED (`--O0`) `.r1cs`, evaluated over the circom interpreter's complete
    # witness (oracle/circom executing the reference's sources; needs /root/reference), parsed by tests/r1cs_util.read_r1cs
    import gzip
    import r1cs_util as ru
    import real_email as R
    from oracle.circom import ev
    from oracle.circom.runtime import iter_signals

    def complete(prog, inp, want_sha=None):
        root = prog.run(inp)
        w = [1] + [v for _, v, _, _ in iter_signals(root, with_names=False)]
        sha = hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in w)).hexdigest()
        assert want_sha is None or sha == want_sha
        return w, sha

    # packages/circuits/tests/rsa.test.ts:64-103 on rsa-test.circom (tests/golden/o0_rsa.*)
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "o0_rsa.json")))
    w, _ = complete(ev.program("tests/test-circuits/rsa-test.circom"), meta["inputs"], meta["witness_sha256"])
    hdr, cons = ru.read_r1cs(gzip.open(os.path.join(ROOT, "tests", "golden", "o0_rsa.r1cs.gz"), "rb").read())
    out["o0_rsa_kat"] = {"constraints": len(cons), "wires": hdr["n_wires"], "standard": ru.abc_digest(cons, w), "montgomery": ru.abc_digest(cons, w, True)}
    # EmailVerifier(576,192,121,17,0,0,0,0) on the REAL test.eml (tests/real_email.py): (a) the kept-v1 system derived by
    # zkwg.r1cs over the oracle's witness, (b) the interpreter-generated O0 system (artifacts/o0_ev_576_192.r1cs, 3.13 M
    # constraints) over the interpreter's complete witness
    inp = R.ev_inputs("test_eml", 576, 192)
    main_ev = zk.EmailVerifier(576, 192, 121, 17, 0, {k: [int(x) for x in v] if isinstance(v, list) else int(v) for k, v in inp.items()},
                               body_hash_regex=lambda m: zk.BodyHashRegexV1(576, m))
    sym = comp.symbols_kept(main_ev)
    wk = comp.witness_kept(main_ev)
    hdr, cons = ru.read_r1cs(zr.email_verifier_r1cs(sym, 576, 192))
    assert hdr["n_wires"] == len(wk)
    out["ev_test_eml_576_192_kept"] = {"constraints": len(cons), "wires": len(wk), "standard": ru.abc_digest(cons, wk), "montgomery": ru.abc_digest(cons, wk, True),
                                       "witness_sha256": hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in wk)).hexdigest()}
    # all template flags at once (header / body masks, removeSoftLineBreaks = 1): the first valid synthetic input of
    # tests/test_r1cs._flag_inputs(576, 192) -- the kept-v1 system of zkwg.r1cs over the oracle's witness
    from test_r1cs import _flag_inputs
    finp, _ = _flag_inputs(576, 192, 0)
    fi = {k: [int(x) for x in v] if isinstance(v, list) else int(v) for k, v in finp.items()}
    main_f = zk.EmailVerifier(576, 192, 121, 17, 0, fi, body_hash_regex=lambda m: zk.BodyHashRegexV1(576, m),
                              enableHeaderMasking=1, enableBodyMasking=1, removeSoftLineBreaks=1)
    symf, wf = comp.symbols_kept(main_f), comp.witness_kept(main_f)
    hdr, cons = ru.read_r1cs(zr.email_verifier_r1cs(symf, 576, 192, 1, 1, 1))
    assert hdr["n_wires"] == len(wf)
    out["ev_flags_576_192_kept"] = {"constraints": len(cons), "wires": len(wf), "standard": ru.abc_digest(cons, wf), "montgomery": ru.abc_digest(cons, wf, True),
                                    "witness_sha256": hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in wf)).hexdigest()}
    art = os.path.join(ROOT, "artifacts", "o0_ev_576_192.r1cs.gz")
    if os.path.exists(art):
        w, sha = complete(ev.email_verifier(576, 192), inp)
        hdr, cons = ru.read_r1cs(gzip.open(art, "rb").read())
        assert hdr["n_wires"] == len(w)
        assert all(sum(cf * w[k] for k, cf in a) * sum(cf * w[k] for k, cf in b) % P == sum(cf * w[k] for k, cf in c) % P for a, b, c in cons)
        out["o0_ev_test_eml_576_192"] = {"constraints": len(cons), "wires": len(w), "witness_sha256": sha,
                                         "standard": ru.abc_digest(cons, w), "montgomery": ru.abc_digest(cons, w, True)}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "abc_digests.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
