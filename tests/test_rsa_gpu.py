"""GPU parity: main = RSAVerifier65537(121,17) (reference: packages/circuits/tests/rsa.test.ts,
tests/test-circuits/rsa-test.circom) -- HIP witness vs the literal Python oracle, bit-exact."""
import hashlib

import pytest

from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG, limbs, oracle_rsa

pytestmark = pytest.mark.gpu


def _calc():
    import zkwg
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0)
    return c, zkwg.WitnessCalculator(c)


def test_rsa_1024_kat():
    # rsa.test.ts:64-103 "should verify 1024 bit rsa signature correctly"
    import zkwg
    c, wc = _calc()
    w = wc.calculateWitness({"signature": [str(x) for x in limbs(KAT_SIG)],
                             "modulus": [str(x) for x in limbs(KAT_PUB)],
                             "message": [str(x) for x in KAT_MSG]})
    assert w == oracle_rsa(KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB))


def test_rsa_wrong_message_assert_failed():
    # rsa.test.ts:105-144 "should fail when verifying with an incorrect signature"
    import zkwg
    c, wc = _calc()
    m2 = list(KAT_MSG)
    m2[0] += 1
    with pytest.raises(zkwg.ZkwgError, match="Assert Failed"):
        wc.calculateWitness({"signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB), "message": m2})


def test_rsa_2048_batch_bit_exact():
    import zkwg
    from zkwg.synth import test_key, pkcs1_sign_digest
    c, wc = _calc()
    key = test_key()
    inputs, expect = [], []
    for i in range(6):
        digest = hashlib.sha256(b"zkwg synthetic header %d" % i).digest()
        sig = pkcs1_sign_digest(key, digest)
        msg = limbs(int.from_bytes(digest, "big"))
        if i == 4:
            sig ^= 1  # tampered signature
        inputs.append({"signature": limbs(sig), "modulus": limbs(key["n"]), "message": msg})
        expect.append((msg, limbs(sig), limbs(key["n"])))
    wits, status = wc.calculateBatch(inputs)
    assert status == [0, 0, 0, 0, 4, 0]
    for i, (wb, ex) in enumerate(zip(wits, expect)):
        if status[i] == 0:
            assert zkwg.witness_ints(wb) == oracle_rsa(*ex)
