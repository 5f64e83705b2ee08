"""Host-side input generation mirror (zkwg.inputs) -- follows the reference's helpers tests
(packages/helpers/tests/input-generators.test.ts:39-63) and checks the SHA midstate precompute."""
import hashlib
import struct

import pytest

from zkwg import inputs, synth


def _finish_sha(state32: bytes, rest: bytes) -> bytes:
    st = list(struct.unpack(">8I", state32))
    for i in range(0, len(rest), 64):
        st = inputs._compress(st, rest[i:i + 64])
    return struct.pack(">8I", *st)


def test_sha256_pad_and_limbs():
    p, n = inputs.sha256_pad(b"abc", 128)
    assert n == 64 and len(p) == 128 and p[3] == 0x80 and p[56:64] == (24).to_bytes(8, "big") and p[64:] == bytes(64)
    with pytest.raises(ValueError):
        inputs.sha256_pad(b"x" * 200, 128)
    x = (1 << 2047) | 12345
    limbs = [int(v) for v in inputs.to_circom_bigint_bytes(x)]
    assert len(limbs) == 17 and sum(v << (121 * i) for i, v in enumerate(limbs)) == x and all(v < (1 << 121) for v in limbs)


def test_precompute_selector_cut_is_on_a_64_byte_boundary_and_midstate_is_right():
    # input-generators.test.ts:39-53
    d = synth.synthetic_dkim_result(1, 0, body_len=1000)
    sel = d["body"][700:712].decode()
    inp = inputs.generate_email_verifier_inputs_from_dkim_result(d, 1024, 512, sha_precompute_selector=sel)
    cut = (d["body"].find(sel.encode()) // 64) * 64
    assert cut == 640
    body_padded, padded_len = inputs.sha256_pad(d["body"], 1024 + 64)
    assert int(inp["emailBodyLength"]) == padded_len - cut
    remaining = bytes(int(b) for b in inp["emailBody"])
    assert remaining[:padded_len - cut] == body_padded[cut:padded_len]
    # midstate + remaining blocks = SHA-256(body)
    pre = bytes(int(b) for b in inp["precomputedSHA"])
    assert pre == inputs.partial_sha(d["body"][:cut])
    assert _finish_sha(pre, remaining[:padded_len - cut]) == hashlib.sha256(d["body"]).digest()


def test_bad_selector_and_too_long_remaining_body_throw():
    # input-generators.test.ts:55-63
    d = synth.synthetic_dkim_result(1, 1, body_len=600)
    with pytest.raises(ValueError, match="not found"):
        inputs.generate_email_verifier_inputs_from_dkim_result(d, 1024, 512, sha_precompute_selector="\x01nope")
    with pytest.raises(ValueError, match="longer than max"):
        inputs.generate_email_verifier_inputs_from_dkim_result(d, 1024, 256)


def test_body_hash_index_points_at_the_bh_value():
    d = synth.synthetic_dkim_result(2, 3, body_len=200)
    inp = inputs.generate_email_verifier_inputs_from_dkim_result(d, 1024, 512)
    i = int(inp["bodyHashIndex"])
    hdr = bytes(int(b) for b in inp["emailHeader"])
    assert hdr[i:i + 44].decode() == d["bodyHash"] and hdr[i - 3:i] == b"bh="


def test_input_generators_test_ts_precompute_selector_on_email_good_large():
    # packages/helpers/tests/input-generators.test.ts:39-53: selector 'thousands' on email-good-large.eml ->
    # emailBody starts at the previous 64-byte boundary: 'h hundreds of thousands of blocks.'
    # (fixture: canonical body extracted by tests/golden/make_email_good_large_fixture.py, pinned by its bh=)
    import base64
    import hashlib
    import json
    import os
    from conftest import ROOT
    from zkwg import inputs
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "email_good_large.json")))
    body = bytes.fromhex(fx["canonical_body_hex"])
    assert fx["body_hash_matches_bh"] and base64.b64encode(hashlib.sha256(body).digest()).decode() == fx["bh"]
    d = {"headers": b"x: bh=" + fx["bh"].encode(), "body": body, "bodyHash": fx["bh"], "publicKey": 1, "signature": 1}
    inp = inputs.generate_email_verifier_inputs_from_dkim_result(d, 1024, 1536, sha_precompute_selector=fx["selector"])
    got = bytes(int(b) for b in inp["emailBody"])
    assert got.startswith(fx["expected_body_prefix"].encode())
    # the midstate really is the SHA-256 state of the cut-off prefix: finishing the hash from it gives bh
    cut = body.find(got[:len(fx["expected_body_prefix"])])
    assert cut > 0 and cut % 64 == 0
    assert inputs.partial_sha(body[:cut]) == bytes(int(b) for b in inp["precomputedSHA"])
    with pytest.raises(ValueError, match='SHA precompute selector "Bla Bla" not found in cleaned body'):
        inputs.generate_email_verifier_inputs_from_dkim_result(d, 1024, 1536, sha_precompute_selector="Bla Bla")
