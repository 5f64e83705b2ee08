"""The reference's REAL signed emails through every implementation.

`packages/circuits/tests/test-emails/test.eml` and `packages/helpers/tests/test-data/email-good-large.eml` are signed by
`d=icloud.com; s=1a1hai`; the reference fetches that key from DNS, the tree does not hold it.  tests/golden/make_icloud_key.py
recovers the modulus offline (gcd over the two signatures), tests/golden/make_real_email_fixture.py runs the reference's own
test mains on the real inputs through the circom interpreter (which executes every `===`), requires the literal Python
oracle to agree on every signal, and commits digests (tests/golden/real_email_digests.json).  Here:

  CPU   the recovered key verifies both signatures; the C oracle reproduces the digests; every tamper case of
        email-verifier.test.ts:61-186 fails; (with /root/reference) the interpreter reproduces a digest
  GPU   the HIP path reproduces the same digests through the C-ABI: rsa.test.ts:27-62, email-verifier.test.ts:33-59 and
        :61-186 (status 4 / "Assert Failed"), the no-body / body-mask / header-mask mains, email-good-large.eml with the
        precompute selector "thousands" from RAW canonical bytes through zk_gen_inputs, and -- where the interpreter
        generated `.sym` / `.r1cs` exist -- the COMPLETE `--O0` witness of the real email against the interpreter's
"""
import gzip
import hashlib
import json
import os

import pytest

from conftest import ROOT
import real_email as R

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "real_email_digests.json")))["cases"]
EV_CASES = [k for k in GOLD if k.startswith("ev_")]
MASK = lambda n: [1 if 25 < i < 50 else 0 for i in range(n)]


def _opts(name):
    o = dict(GOLD[name]["options"])
    N, M = GOLD[name]["params"][:2]
    if o.get("enable_body_masking"):
        o["body_mask"] = MASK(M)
    if o.get("enable_header_masking"):
        o["header_mask"] = MASK(N)
    return o


def _inputs(name):
    g = GOLD[name]
    return R.ev_inputs(g["email"], g["params"][0], g["params"][1], **_opts(name))


def _circuit(name, device, **kw):
    import zkwg
    N, M, ign, hm, bm, _ = GOLD[name]["params"]
    return zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, ignore_body_hash_check=ign, enable_header_masking=hm,
                        enable_body_masking=bm, device=device, **kw)


def test_recovered_key_verifies_both_signatures():
    n = R.icloud_modulus()
    assert n.bit_length() == 2048
    for which in ("test_eml", "email_good_large"):
        d = R.dkim_result(which)
        t = bytes.fromhex("3031300d060960864801650304020105000420") + hashlib.sha256(d["headers"]).digest()
        em = int.from_bytes(b"\x00\x01" + b"\xff" * (256 - len(t) - 3) + b"\x00" + t, "big")
        assert pow(d["signature"], 65537, n) == em
        import base64
        assert base64.b64encode(hashlib.sha256(d["body"]).digest()).decode() == d["bodyHash"]
    # rsa.test.ts:40-58: the message limbs are the header hash of test.eml
    h = int.from_bytes(hashlib.sha256(R.dkim_result("test_eml")["headers"]).digest(), "big")
    assert [str((h >> (121 * i)) & ((1 << 121) - 1)) for i in range(17)] == R.RSA_TEST_MESSAGE


@pytest.mark.parametrize("name", EV_CASES)
def test_c_oracle_reproduces_the_real_email_digests(name):
    from oracle import coracle
    g = GOLD[name]
    N, M, ign = g["params"][:3]
    wits, status, W = coracle.calculate(0, N, M, ign, [_inputs(name)])
    assert status == [0] and W == g["W_kept"]
    assert hashlib.sha256(wits[0]).hexdigest() == g["kept_sha256"]
    assert str(int.from_bytes(wits[0][32:64], "little")) == g["outputs"]["pubkeyHash"]


def test_c_oracle_rsa_2048_real_signature():
    from oracle import coracle
    inp = R.ev_inputs()
    wits, status, W = coracle.calculate(2, 0, 0, 0, [{"signature": inp["signature"], "modulus": inp["pubkey"], "message": R.RSA_TEST_MESSAGE}])
    assert status == [0]
    assert hashlib.sha256(wits[0]).hexdigest() == GOLD["rsa_test_eml_2048"]["kept_sha256"]
    bad = list(R.RSA_TEST_MESSAGE)
    bad[0] = str(int(bad[0]) + 1)       # rsa.test.ts:105-144 style negative on the real key
    _, status, _ = coracle.calculate(2, 0, 0, 0, [{"signature": inp["signature"], "modulus": inp["pubkey"], "message": bad}])
    assert status == [4]


def test_tamper_cases_fail_in_both_oracles():
    from oracle import coracle
    from oracle.pyref import zkemail as zk, comp
    labels = []
    for label, bad in R.tamper_cases():
        _, st, _ = coracle.calculate(0, 640, 768, 0, [bad])
        assert st == [4], label
        with pytest.raises(comp.AssertFailed):
            zk.EmailVerifier(640, 768, 121, 17, 0, bad, body_hash_regex=lambda m: zk.BodyHashRegexV1(640, m))
        labels.append(label)
    assert len(labels) == 6 == len(GOLD["ev_test_eml"]["tamper_cases"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/packages/circuits"), reason="/root/reference is not present")
def test_interpreter_on_the_reference_main_reproduces_the_no_body_digest():
    """email-verifier-no-body-test.circom (the smallest of the real-email mains) executed from the reference's sources:
    complete O0 digest and kept-v1 digest as committed."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_real_email_fixture as mk
    rec = mk.run_case("ev_no_body")
    for k in ("W_kept", "kept_sha256", "n_o0", "o0_sha256", "outputs"):
        assert rec[k] == GOLD["ev_no_body"][k], k


# ------------------------------------------------------------------------------------------------ GPU


@pytest.mark.gpu
@pytest.mark.parametrize("name", EV_CASES)
def test_hip_reproduces_the_real_email_digests(name):
    import zkwg
    g = GOLD[name]
    c = _circuit(name, 0)
    assert c.W == g["W_kept"]
    wc = zkwg.WitnessCalculator(c)
    inp = _inputs(name)
    b = wc.calculateBinWitness(inp)
    assert hashlib.sha256(b).hexdigest() == g["kept_sha256"]
    c.assertOut(b, {"pubkeyHash": g["outputs"]["pubkeyHash"], "shaHi": g["outputs"]["shaHi"], "shaLo": g["outputs"]["shaLo"]})
    if name == "ev_body_mask":       # email-verifier-with-body-mask.test.ts:55-58 assertOut(maskedBody)
        c.assertOut(b, {"maskedBody": [int(x) if m else 0 for x, m in zip(inp["emailBody"], MASK(768))]})
    if name == "ev_header_mask":     # email-verifier-with-header-mask.test.ts:55-58
        c.assertOut(b, {"maskedHeader": [int(x) if m else 0 for x, m in zip(inp["emailHeader"], MASK(640))]})


@pytest.mark.gpu
def test_hip_real_email_tamper_cases_assert_failed():
    """email-verifier.test.ts:61-186, each built the way the reference builds it, in ONE batch beside the good email."""
    import zkwg
    c = _circuit("ev_test_eml", 0)
    wc = zkwg.WitnessCalculator(c)
    cases = R.tamper_cases()
    wits, status = wc.calculateBatch([_inputs("ev_test_eml")] + [x for _, x in cases])
    assert status == [0] + [4] * len(cases)
    assert hashlib.sha256(wits[0]).hexdigest() == GOLD["ev_test_eml"]["kept_sha256"]
    for label, bad in cases:
        with pytest.raises(zkwg.ZkwgError, match="Assert Failed"):
            wc.calculateWitness(bad)


@pytest.mark.gpu
def test_hip_rsa_main_real_2048_bit_signature_kept_and_complete():
    """rsa.test.ts:27-62 "should verify 2048 bit rsa signature correctly" with the real key; plus the complete `--O0` witness
    of rsa-test.circom (tests/golden/o0_rsa.*: interpreter-generated `.sym` / `.r1cs`) against the interpreter's digest."""
    import zkwg
    g = GOLD["rsa_test_eml_2048"]
    inp = R.ev_inputs()
    rsa_in = {"signature": inp["signature"], "modulus": inp["pubkey"], "message": R.RSA_TEST_MESSAGE}
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0)
    b = zkwg.WitnessCalculator(c).calculateBinWitness(rsa_in)
    assert hashlib.sha256(b).hexdigest() == g["kept_sha256"]
    base = os.path.join(ROOT, "tests", "golden", "o0_rsa")
    meta = json.load(open(base + ".json"))
    co = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0, sym=gzip.open(base + ".sym.gz", "rb").read(),
                      sym_alias=meta["alias"], r1cs=gzip.open(base + ".r1cs.gz", "rb").read())
    assert co.W == g["n_o0"]
    wit, status = co.calculate_batch_host(co.pack(rsa_in))
    assert status == [0]
    assert hashlib.sha256(wit).hexdigest() == g["o0_sha256"]


@pytest.mark.gpu
def test_hip_email_good_large_from_raw_bytes_with_selector():
    """packages/helpers/tests/input-generators.test.ts:39-53 continued into the circuit: the canonical header / body bytes of
    email-good-large.eml, the real signature and key go to zk_gen_inputs (sha256Pad, generatePartialSHA with the selector
    "thousands", limb split on the device); the records equal the host mirror's and the EmailVerifier(1024,1536) witness of
    the real email equals the interpreter's / oracle's digest."""
    import zkwg
    g = GOLD["ev_good_large_selector"]
    c = _circuit("ev_good_large_selector", 0)
    d = R.dkim_result("email_good_large")
    recs, st = zkwg.generate_inputs_device(c, [d, d], selector="thousands")
    assert st == [0, 0]
    host = recs.cpu().numpy()
    inp = _inputs("ev_good_large_selector")
    assert host[0].tobytes() == c.pack(inp) == host[1].tobytes()
    body = bytes(int(x) for x in inp["emailBody"])
    assert body.startswith(b"h hundreds of thousands of blocks.")       # input-generators.test.ts:49-52
    wit, status = c.calculate_batch_host(host.tobytes())
    assert status == [0, 0]
    wb = c.witness_bytes
    assert hashlib.sha256(wit[:wb]).hexdigest() == g["kept_sha256"] and wit[wb:] == wit[:wb]


def _complete(name, tag):
    base = os.path.join(ROOT, "artifacts", f"o0_ev_{tag[0]}_{tag[1]}")
    if not os.path.exists(base + ".json"):
        pytest.skip(f"artifacts/o0_ev_{tag[0]}_{tag[1]}.* not built (needs /root/reference)")
    import zkwg
    g = GOLD[name]
    meta = json.load(open(base + ".json"))
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=tag[0], max_body=tag[1], device=0, sym=gzip.open(base + ".sym.gz", "rb").read(),
                     sym_alias=meta["alias"], r1cs=gzip.open(base + ".r1cs.gz", "rb").read())
    assert c.W == g["n_o0"]
    wit, status = c.calculate_batch_host(c.pack(_inputs(name)))
    assert status == [0]
    assert hashlib.sha256(wit).hexdigest() == g["o0_sha256"]


@pytest.mark.gpu
def test_hip_complete_o0_witness_of_the_real_test_eml():
    """Every one of the 3,113,238 signals the reference's email-verifier.circom declares at (576,192), for the REAL test.eml,
    equals the interpreter's (the product sees only the interpreter-generated `.sym` + `.r1cs` and the input record)."""
    _complete("ev_test_eml_576_192", (576, 192))


@pytest.mark.gpu
def test_hip_complete_o0_witness_of_email_good_large_at_the_default_size():
    """The same at the circuit size the reference deploys, EmailVerifier(1024,1536,121,17,0,0,0,0): 9,255,356 signals of the
    real email-good-large.eml (precompute selector "thousands")."""
    _complete("ev_good_large_selector", (1024, 1536))
