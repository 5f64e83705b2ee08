"""Pins the fast C oracle (oracle/c/zkwg_oracle.c) to the literal Python oracle."""
import pytest

from test_ev_cpu import _inputs, _oracle_ev
from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG, limbs
from conftest import sha_pad


def ints(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


@pytest.mark.parametrize("ignore", [1, 0])
def test_c_oracle_email_verifier_matches_python(ignore):
    from oracle import coracle
    from oracle.pyref import comp
    N, M = 576, 192
    ins = [_inputs(N, M, ignore, index=i, body_len=30 + 21 * i) for i in range(3)]
    wits, status, W = coracle.calculate(0, N, M, ignore, ins)
    assert status == [0, 0, 0]
    for inp, wb in zip(ins, wits):
        assert ints(wb) == comp.witness_kept(_oracle_ev(N, M, ignore, inp))


def test_c_oracle_rsa_kat_and_negative():
    from oracle import coracle
    from oracle.pyref import zkemail as zk, comp
    good = {"message": KAT_MSG, "signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB)}
    bad = dict(good, message=[KAT_MSG[0] + 1] + KAT_MSG[1:])
    wits, status, W = coracle.calculate(2, 0, 0, 0, [good, bad])
    assert status == [0, 4]
    main = zk.RSAVerifier65537(121, 17, KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB), is_main=True)
    assert ints(wits[0]) == comp.witness_kept(main)


def test_c_oracle_sha_main():
    from oracle import coracle
    from oracle.pyref import zkemail as zk, comp
    N = 192
    ins, mains = [], []
    for m in (b"", b"0", b"hello world", bytes(range(100))):
        p, n = sha_pad(m, N)
        ins.append({"paddedIn": list(p), "paddedInLength": n})
        mains.append(zk.Sha256Bytes(N, list(p), n, is_main=True))
    wits, status, W = coracle.calculate(1, N, 0, 0, ins)
    assert status == [0] * 4
    for wb, main in zip(wits, mains):
        assert ints(wb) == comp.witness_kept(main)


def test_c_oracle_tamper_status():
    from oracle import coracle
    import copy
    N, M = 576, 192
    good = _inputs(N, M, 0, index=2, body_len=60)
    b1 = copy.deepcopy(good); b1["emailBody"][0] = str(int(b1["emailBody"][0]) ^ 1)
    b2 = copy.deepcopy(good); b2["emailHeader"][int(good["emailHeaderLength"]) + 1] = "1"
    _, status, _ = coracle.calculate(0, N, M, 0, [good, b1, b2], want_witness=False)
    assert status == [0, 4, 4]
