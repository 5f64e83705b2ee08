"""The reference's REAL signed emails as `DKIMVerificationResult`-shaped dicts, from committed fixtures only
(nothing here reads /root/reference, so the `-m gpu` tests can use it on the GPU box):

  test_eml           packages/circuits/tests/test-emails/test.eml (= helpers/tests/test-data/email-good.eml)
  email_good_large   packages/helpers/tests/test-data/email-good-large.eml

Both are signed by `d=icloud.com; s=1a1hai`; the reference fetches that key from DNS
(packages/helpers/src/dkim/index.ts:105-131).  tests/golden/icloud_1a1hai.json holds the modulus recovered
offline from the two signatures (tests/golden/make_icloud_key.py), both of which verify under it.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_G = os.path.join(HERE, "golden")
_cache = {}


def _load(name):
    if name not in _cache:
        _cache[name] = json.load(open(os.path.join(_G, name)))
    return _cache[name]


def icloud_modulus():
    return int(_load("icloud_1a1hai.json")["modulus_hex"], 16)


def dkim_result(which="test_eml"):
    """dict(headers, body, bodyHash, publicKey, signature): the fields of `verifyDKIMSignature`'s result that
    `generateEmailVerifierInputsFromDKIMResult` reads (input-generators.ts:190-252)."""
    key = _load("icloud_1a1hai.json")
    idx = {"test_eml": 0, "email_good_large": 1}[which]
    em = key["emails"][idx]
    body_fx = _load("test_eml.json" if idx == 0 else "email_good_large.json")
    body = bytes.fromhex(body_fx["canonical_body_hex"])
    assert len(body) == em["canonical_body_len"]
    return {"headers": bytes.fromhex(em["canonical_header_hex"]), "body": body, "bodyHash": em["bh"],
            "publicKey": int(key["modulus_hex"], 16), "signature": int(em["signature_hex"], 16)}


# packages/circuits/tests/rsa.test.ts:40-58: the message limbs of test.eml's header hash
RSA_TEST_MESSAGE = ["1156466847851242602709362303526378170", "191372789510123109308037416804949834", "7204"] + ["0"] * 14


def ev_inputs(which="test_eml", max_header=640, max_body=768, **kw):
    """generateEmailVerifierInputsFromDKIMResult(dkimResult, {maxHeadersLength, maxBodyLength, ...}) on the real email
    (email-verifier.test.ts:33-41 uses 640 / 768)."""
    from zkwg import inputs
    return inputs.generate_email_verifier_inputs_from_dkim_result(dkim_result(which), max_header, max_body, **kw)


def tamper_cases(d=None, max_header=640, max_body=768):
    """The six negative cases of packages/circuits/tests/email-verifier.test.ts:61-186 built the way the reference
    builds them -> list of (label, CircuitInput)."""
    from zkwg import inputs
    d = d or dkim_result("test_eml")
    gen = lambda dk: inputs.generate_email_verifier_inputs_from_dkim_result(dk, max_header, max_body)
    out = []
    out.append((":61-79 signature + 1", gen(dict(d, signature=d["signature"] + 1))))
    h = bytearray(d["headers"]); h[0] = 1
    out.append((":81-102 header[0] = 1", gen(dict(d, headers=bytes(h)))))
    x = gen(d); x["emailHeader"][max_header - 1] = "1"
    out.append((":104-123 header padding", x))
    b = bytearray(d["body"]); b[-1] = 1
    out.append((":125-146 body[-1] = 1", gen(dict(d, body=bytes(b)))))
    x = gen(d); x["emailBody"][max_body - 1] = "1"
    out.append((":148-166 body padding", x))
    out.append((":168-186 bodyHash + 'a'", gen(dict(d, bodyHash=d["bodyHash"] + "a"))))
    return out
