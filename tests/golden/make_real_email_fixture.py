"""Generates tests/golden/real_email_digests.json: the reference's REAL-email circuit tests executed by the circom
interpreter on the reference's own test mains (needs /root/reference; run from the repo root, ~20 min):

    python tests/golden/make_real_email_fixture.py [case ...]

Inputs come from the committed fixtures (tests/real_email.py: test.eml / email-good-large.eml canonical bytes, their
signatures and the icloud.com/1a1hai modulus recovered by make_icloud_key.py), built exactly like
generateEmailVerifierInputsFromDKIMResult (packages/helpers/src/input-generators.ts:190-252).

Per case the script (1) runs the interpreter on the reference's test main, which executes every `===` / assert of
the circuit (a wrong key or canonicalisation would fail at rsa.circom:44), (2) runs the literal Python oracle and
requires every one of its signals to carry the interpreter's value, (3) records
    kept_sha256   SHA-256 of the kept-v1 witness (product order; interpreter values == pyref values)
    o0_sha256     SHA-256 of the interpreter's complete witness (every declared signal, O0 order, constant first)
    outputs       pubkeyHash, shaHi, shaLo
and (4) checks that each tamper case of email-verifier.test.ts:61-186 raises Assert Failed in the interpreter.
The C oracle (CPU) and the HIP path (GPU) are then held to these digests by tests/test_real_email.py.
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zk-email-verify_amd", "py")]

TC = "tests/test-circuits/"
MASK = lambda n: [1 if 25 < i < 50 else 0 for i in range(n)]   # email-verifier-with-*-mask.test.ts:35-37

# name -> (reference test, test main, EmailVerifier parameters, email, generator options)
CASES = {
    "ev_test_eml": ("email-verifier.test.ts:33-44", "email-verifier-test.circom", (640, 768, 0, 0, 0, 0), "test_eml", {}),
    "ev_test_eml_selector": ("email-verifier.test.ts:46-59", "email-verifier-test.circom", (640, 768, 0, 0, 0, 0), "test_eml",
                             {"sha_precompute_selector": "How are"}),
    "ev_no_body": ("email-verifier-no-body.test.ts:33-46", "email-verifier-no-body-test.circom", (640, 768, 1, 0, 0, 0), "test_eml",
                   {"ignore_body_hash_check": True}),
    "ev_body_mask": ("email-verifier-with-body-mask.test.ts:34-60", "email-verifier-with-body-mask-test.circom", (640, 768, 0, 0, 1, 0),
                     "test_eml", {"enable_body_masking": True, "body_mask": MASK(768)}),
    "ev_header_mask": ("email-verifier-with-header-mask.test.ts:34-60", "email-verifier-with-header-mask-test.circom",
                       (640, 768, 0, 1, 0, 0), "test_eml", {"enable_header_masking": True, "header_mask": MASK(640)}),
    "ev_good_large_selector": ("helpers/tests/input-generators.test.ts:39-53 through EmailVerifier(1024,1536)", None,
                               (1024, 1536, 0, 0, 0, 0), "email_good_large", {"sha_precompute_selector": "thousands"}),
    # the real test.eml at the size whose interpreter-generated `.sym` / `.r1cs` ship under artifacts/ (complete `--O0` witness on the GPU)
    "ev_test_eml_576_192": ("email-verifier.test.ts:33-44 at (576,192)", None, (576, 192, 0, 0, 0, 0), "test_eml", {}),
    "rsa_test_eml_2048": ("rsa.test.ts:27-62", "rsa-test.circom", None, "test_eml", {}),
}


def digest(vals):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


def pyref_main(params, inp):
    from oracle.pyref import zkemail as zk
    N, M, ign, hm, bm, rs = params
    iinp = {k: [int(x) for x in v] if isinstance(v, list) else int(v) for k, v in inp.items()}
    return zk.EmailVerifier(N, M, 121, 17, ign, iinp, body_hash_regex=lambda m: zk.BodyHashRegexV1(N, m),
                            enableHeaderMasking=hm, enableBodyMasking=bm)


def run_case(name):
    import real_email as R
    import zkwg
    from oracle.circom import ev
    from oracle.circom.compare import flat_walk_kept
    from oracle.circom.runtime import AssertFailed, iter_signals
    from oracle.pyref import comp
    cite, main_file, params, which, opts = CASES[name]
    t0 = time.time()
    if name == "rsa_test_eml_2048":
        d = R.dkim_result(which)
        inp = {"signature": R.ev_inputs(which)["signature"], "modulus": R.ev_inputs(which)["pubkey"], "message": R.RSA_TEST_MESSAGE}
        prog = ev.program(TC + main_file)
        root = prog.run(inp)
        from test_rsa_cpu import oracle_rsa
        kept = oracle_rsa([int(x) for x in inp["message"]], [int(x) for x in inp["signature"]], [int(x) for x in inp["modulus"]])
        c0 = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1)
        main = None
    else:
        inp = R.ev_inputs(which, params[0], params[1], **opts)
        if main_file:
            prog = ev.program(TC + main_file)
        else:
            prog = ev.email_verifier(params[0], params[1])
        root = prog.run(inp)
        main = pyref_main(params, inp)
        kept = comp.witness_kept(main)
        c0 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=params[0], max_body=params[1], ignore_body_hash_check=params[2],
                          enable_header_masking=params[3], enable_body_masking=params[4], device=-1)
    got = {}
    # email-verifier.circom:100,161 name both mask components `byteMask` (one per if-block); kept-v1 tells them apart
    ren = None
    if params and (params[3] or params[4]):
        assert not (params[3] and params[4])
        ren = ("main.byteMask.", "main.byteMask_header." if params[3] else "main.byteMask_body.")
    for k, v, _, _ in flat_walk_kept(root, prog.templates_src):
        if ren and k.startswith(ren[0]):
            k = ren[1] + k[len(ren[0]):]
        got[k] = v
    n_pyref = 0
    if main is not None:
        for nm, v, _ in main.walk():
            n_pyref += 1
            assert got[nm] == v, (name, nm)
    sym = c0.symbols()
    assert len(sym) == len(kept) == c0.W
    from_interp = [1] + [got[nm] for _, nm in sym[1:]]
    assert from_interp == kept, name            # interpreter (reference .circom) == pyref on every kept-v1 slot
    h = hashlib.sha256()
    h.update((1).to_bytes(32, "little"))
    n_o0 = 1
    for _, v, _, _ in iter_signals(root, with_names=False):
        h.update(v.to_bytes(32, "little"))
        n_o0 += 1
    rec = {"reference_test": cite, "main": main_file or f"EmailVerifier({params[0]},{params[1]},121,17,0,0,0,0) [email-verifier.circom]", "params": params,
           "email": which, "options": {k: v for k, v in opts.items() if not k.endswith("_mask")},
           "W_kept": len(kept), "kept_sha256": digest(kept), "n_o0": n_o0, "o0_sha256": h.hexdigest(),
           "pyref_signals_checked": n_pyref}
    if main is not None:
        rec["outputs"] = {"pubkeyHash": str(kept[1]), "shaHi": str(kept[2]), "shaLo": str(kept[3])}
        rec["emailHeaderLength"], rec["emailBodyLength"] = inp["emailHeaderLength"], inp.get("emailBodyLength")
        rec["bodyHashIndex"] = inp.get("bodyHashIndex")
    if name == "ev_test_eml":
        # every negative case of email-verifier.test.ts:61-186 through the reference's circuit text
        tam = []
        for label, bad in R.tamper_cases():
            try:
                ev.program(TC + main_file).run(bad)
                raise SystemExit(f"tamper case {label} was accepted by the interpreter")
            except AssertFailed as e:
                tam.append({"case": label, "interpreter": str(e)[:160]})
        rec["tamper_cases"] = tam
    rec["seconds"] = round(time.time() - t0, 1)
    return rec


def main(argv):
    dst = os.path.join(ROOT, "tests", "golden", "real_email_digests.json")
    out = json.load(open(dst)) if os.path.exists(dst) else {"layout": "kept-v1", "key": "tests/golden/icloud_1a1hai.json", "cases": {}}
    for name in (argv or list(CASES)):
        rec = run_case(name)
        out["cases"][name] = rec
        print(name, {k: v for k, v in rec.items() if k not in ("tamper_cases",)}, flush=True)
        json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
