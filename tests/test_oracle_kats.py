"""Pins the literal Python oracle on every known answer the reference's own tests hold for this
path.  Fixture tests/golden/test_eml.json was generated from the
reference's test.eml by tests/golden/make_test_eml_fixture.py."""
import hashlib
import json
import os
import random

import pytest

from conftest import ROOT, sha_pad
from oracle.pyref import zkemail as zk, comp, bigint_func as bf

FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "test_eml.json")))


def limbs(x, n=121, k=17):
    return [(x >> (n * i)) & ((1 << n) - 1) for i in range(k)]


def test_field_modulus():
    # packages/helpers/src/constants.ts:1
    assert comp.P == 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert comp.P == 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def test_sha_test_ts_vectors():
    # packages/circuits/tests/sha.test.ts:26-41 -- Sha256Bytes(640) vs SHA-256
    for m in (b"0", b"hello world", b""):
        p, n = sha_pad(m, 640)
        c = zk.Sha256Bytes(640, list(p), n, is_main=True)
        assert int("".join(map(str, c.o)), 2).to_bytes(32, "big") == hashlib.sha256(m).digest()


def test_fp_mul_test_ts():
    # packages/circuits/tests/fp-mul.test.ts:34-46: FpMul(2,4): 17 * 20 mod 85 = 0
    assert zk.FpMul(2, 4, [1, 0, 1, 0], [0, 1, 1, 0], [1, 1, 1, 1]).o == [0, 0, 0, 0]
    # :48-64 (fp-mul-test-range-check.circom): r > p must be rejected
    with pytest.raises(comp.AssertFailed):
        zk.FpMul(2, 4, [1, 0, 1, 0], [0, 1, 1, 0], [1, 1, 1, 1], qr_override=([3, 0, 0, 0], [1, 1, 1, 1 + 0]))


def test_rsa_test_ts_1024_kat_and_negative():
    from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG
    zk.RSAVerifier65537(121, 17, KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB), is_main=True)
    with pytest.raises(comp.AssertFailed):
        zk.RSAVerifier65537(121, 17, [KAT_MSG[0] + 1] + KAT_MSG[1:], limbs(KAT_SIG), limbs(KAT_PUB), is_main=True)


def test_test_eml_header_hash_is_the_rsa_test_message():
    # rsa.test.ts:40-58 message limbs == SHA-256(relaxed-canonical signed header of test.eml) packed as
    # email-verifier.circom:74-84 does (sha[255-i] -> 121-bit limbs)
    hdr = bytes.fromhex(FIX["canonical_header_hex"])
    assert len(hdr) == 472
    digest = hashlib.sha256(hdr).digest()
    assert [str(x) for x in limbs(int.from_bytes(digest, "big"))[:3]] == FIX["rsa_test_ts_message_limbs"]
    # and through the circuit restatement: Sha256Bytes + the rsaMessage packing
    p, n = sha_pad(hdr, 640)
    sha = zk.Sha256Bytes(640, list(p), n).o
    msg = [sum(sha[255 - i] << (i % 121) for i in range(256) if i // 121 == j) for j in range(3)]
    assert [str(x) for x in msg] == FIX["rsa_test_ts_message_limbs"]


def test_test_eml_body_hash_base64_and_index():
    body = bytes.fromhex(FIX["canonical_body_hex"])
    assert body == b"Hello,\r\n\r\nHow are you?\r\n"
    assert FIX["bh"] == "7xQMDuoVVU4m0W0WRVSrVXMeGSIASsnucK9dJsrc+vU="
    assert FIX["body_hash_index"] == 363
    hdr = bytes.fromhex(FIX["canonical_header_hex"])
    # BodyHashRegex -> SelectRegexReveal -> Base64Decode on the real header reproduce the body digest,
    # and Sha256BytesPartial with an empty precompute (IV) gives the same
    hp, _ = sha_pad(hdr, 640)
    rx = zk.BodyHashRegex(640, list(hp))
    assert rx.o[0] == 1
    sel = zk.SelectRegexReveal(640, 44, rx.o[1], 363)
    assert bytes(sel.o).decode() == FIX["bh"]
    dec = zk.Base64Decode(32, sel.o)
    assert bytes(dec.o) == hashlib.sha256(body).digest()
    bp, bn = sha_pad(body, 768)
    iv = bytes.fromhex("6a09e667bb67ae853c6ef372a54ff53a510e527f9b05688c1f83d9ab5be0cd19")
    part = zk.Sha256BytesPartial(768, list(bp), bn, list(iv))
    assert int("".join(map(str, part.o)), 2).to_bytes(32, "big") == hashlib.sha256(body).digest()


def test_base64_test_ts():
    # packages/circuits/tests/base64.test.ts:21-31 lookup table, :43 invalid chars
    for ch, val in [(65, 0), (90, 25), (97, 26), (122, 51), (48, 52), (57, 61), (43, 62), (47, 63), (61, 0)]:
        assert zk.Base64Lookup(ch).o == val
    for ch in (34, 64, 91, 44):
        with pytest.raises(comp.AssertFailed):
            zk.Base64Lookup(ch)


def test_pack_bits_test_ts():
    # packages/circuits/tests/pack-bits.test.ts: PackBits(256,128) big-endian halves
    h = hashlib.sha256(b"test data").digest()
    bits = [(b >> (7 - i)) & 1 for b in h for i in range(8)]
    out = zk.PackBits(256, 128, bits).o
    assert out == [int.from_bytes(h[:16], "big"), int.from_bytes(h[16:], "big")]
    assert zk.PackBits(256, 128, [0] * 256).o == [0, 0]
    assert zk.PackBits(256, 128, [1] * 256).o == [(1 << 128) - 1] * 2


def test_select_regex_reveal_test_ts():
    # packages/circuits/tests/select-regex-reveal.test.ts:22-120, SelectRegexReveal(34, 8)
    rng = random.Random(3)
    rev = [ord(c) for c in "zk email"]
    start = rng.randrange(24)
    inp = [0] * 34
    inp[start:start + 8] = rev
    assert zk.SelectRegexReveal(34, 8, inp, start).o == rev
    inp2 = [0] * 34
    inp2[30:32] = [ord("z"), ord("k")]
    assert zk.SelectRegexReveal(34, 8, inp2, 30).o == [ord("z"), ord("k"), 0, 0, 0, 0, 0, 0]
    with pytest.raises(comp.AssertFailed):   # all zero
        zk.SelectRegexReveal(34, 8, [0] * 34, rng.randrange(34))
    s1 = 1 + rng.randrange(24)
    inp3 = [0] * 34
    inp3[s1:s1 + 8] = rev
    with pytest.raises(comp.AssertFailed):   # startIndex points at a zero
        zk.SelectRegexReveal(34, 8, inp3, s1 - 1)
    with pytest.raises(comp.AssertFailed):   # startIndex not at the start of the run
        zk.SelectRegexReveal(34, 8, inp3, s1 + 1)


def test_bigint_func_long_div_matches_integer_division():
    # lib/bigint-func.circom:169-264 literal long_div vs exact integer floor division
    rng = random.Random(11)
    for bits in (2048, 1024, 1500):
        p = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        a, b = rng.randrange(p), rng.randrange(p)
        x = limbs(a * b, 121, 34) + [0] * 66
        out = bf.long_div(121, 17, 17, x, limbs(p) + [0] * 83)
        q = sum(v << (121 * i) for i, v in enumerate(out[0][:18]))
        r = sum(v << (121 * i) for i, v in enumerate(out[1][:17]))
        assert (q, r) == divmod(a * b, p)


def test_remove_soft_line_breaks_test_ts():
    # packages/circuits/tests/remove-soft-line-breaks.test.ts: all 7 cases of RemoveSoftLineBreaks(32)
    # (fixture extracted by tests/golden/make_rslb_kats.py)
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "rslb_kats.json")))
    assert len(kat["cases"]) == 7
    for case in kat["cases"]:
        c = zk.RemoveSoftLineBreaks(kat["maxLength"], case["encoded"], case["decoded"], is_main=True)
        assert c.o == case["isValid"], case["name"]


def test_poseidon16_known_vector_and_modular_chain():
    # circomlibjs' own test vector for 16 inputs pins the regenerated t = 17 constants used by
    # PoseidonModular (utils/hash.circom:49-82); the chain mirrors helpers/src/hash.ts:19-52.
    from oracle.pyref import poseidon
    assert poseidon.poseidon_hash(list(range(1, 17))) == \
        9989051620750914585850546081941653841776809718687451684622678807385399211877
    rng = random.Random(37)
    xs = [rng.randrange(1 << 53) for _ in range(37)]   # poseidon-modular.test.ts:26-28 shape
    out = None
    for i in range(0, 37, 16):
        h = poseidon.poseidon_hash(xs[i:i + 16])
        out = h if out is None else poseidon.poseidon_hash([out, h])
    assert zk.PoseidonModular(37, xs).o == out


def test_byte_mask_test_ts():
    # packages/circuits/tests/byte-mask.test.ts:19-47 (ByteMask(10)); a non-bit mask value must be rejected
    c = zk.ByteMask(10, list(range(1, 11)), [1, 0] * 5)
    assert c.o == [1, 0, 3, 0, 5, 0, 7, 0, 9, 0]
    with pytest.raises(comp.AssertFailed):
        zk.ByteMask(10, list(range(1, 11)), [1, 2, 1, 0, 1, 0, 1, 0, 1, 0])
