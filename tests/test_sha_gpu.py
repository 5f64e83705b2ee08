"""GPU parity: main = Sha256Bytes(N) (reference: packages/circuits/tests/sha.test.ts,
tests/test-circuits/sha-test.circom) -- HIP witness vs the literal Python oracle, bit-exact."""
import hashlib
import random

import pytest

from conftest import sha_pad

pytestmark = pytest.mark.gpu


def _oracle_witness(N, padded, n):
    from oracle.pyref import zkemail as zk, comp
    main = zk.Sha256Bytes(N, list(padded), n, is_main=True)
    return comp.witness_kept(main)


def _run(N, msgs):
    import zkwg
    c = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=N, max_body=0, device=0)
    wc = zkwg.WitnessCalculator(c)
    inputs = []
    for m in msgs:
        p, n = sha_pad(m, N)
        inputs.append({"paddedIn": [str(b) for b in p], "paddedInLength": str(n)})
    wits, status = wc.calculateBatch(inputs)
    return c, wits, status


def test_sha_test_ts_vectors_640():
    # sha.test.ts:26-41: "0", "hello world", "" hashed with Sha256Bytes(640)
    import zkwg
    msgs = [b"0", b"hello world", b""]
    c, wits, status = _run(640, msgs)
    assert status == [0, 0, 0]
    for m, wb in zip(msgs, wits):
        w = zkwg.witness_ints(wb)
        assert w[0] == 1
        digest_bits = w[1:257]
        dig = int("".join(str(b) for b in digest_bits), 2).to_bytes(32, "big")
        assert dig == hashlib.sha256(m).digest()
        p, n = sha_pad(m, 640)
        assert w == _oracle_witness(640, p, n)


def test_sha_random_lengths_bit_exact():
    import zkwg
    rng = random.Random(1234)
    N = 256
    msgs = [bytes(rng.randrange(256) for _ in range(L)) for L in (0, 1, 55, 56, 63, 64, 119, 120, 183, 200, 247)]
    c, wits, status = _run(N, msgs)
    assert status == [0] * len(msgs)
    for m, wb in zip(msgs, wits):
        p, n = sha_pad(m, N)
        assert zkwg.witness_ints(wb) == _oracle_witness(N, p, n)


def test_sha_bad_length_is_assert_failed():
    import zkwg
    N = 128
    c = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=N, max_body=0, device=0)
    wc = zkwg.WitnessCalculator(c)
    p, n = sha_pad(b"abc", N)
    for bad in (n + 1, 0, N + 64):
        with pytest.raises(zkwg.ZkwgError, match="Assert Failed"):
            wc.calculateWitness({"paddedIn": list(p), "paddedInLength": bad})
    # and the oracle agrees
    from oracle.pyref import zkemail as zk, comp
    for bad in (n + 1, 0, N + 64):
        with pytest.raises(comp.AssertFailed):
            zk.Sha256Bytes(N, list(p), bad, is_main=True)


def test_wtns_container():
    import zkwg
    c, wits, status = _run(128, [b"abc"])
    blob = c.wtns(wits[0])
    assert blob[:4] == b"wtns" and int.from_bytes(blob[4:8], "little") == 2
    assert int.from_bytes(blob[8:12], "little") == 2
    assert int.from_bytes(blob[24:28], "little") == 32
    assert int.from_bytes(blob[28:60], "little") == zkwg.FIELD_MODULUS
    assert int.from_bytes(blob[60:64], "little") == c.W
    assert blob[76:] == wits[0]
