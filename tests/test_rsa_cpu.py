"""CPU tests of the RSA path: the product's wave-collective RSA core (host build) + segment
table against the literal Python oracle, on the reference's own RSA KATs
(packages/circuits/tests/rsa.test.ts:64-144)."""
import ctypes as C

import pytest

import hosttest
from zkwg._lib import Config, MAIN_RSA_VERIFIER


def limbs(x, n=121, k=17):
    return [(x >> (n * i)) & ((1 << n) - 1) for i in range(k)]


KAT_SIG = 102386562682221859025549328916727857389789009840935140645361501981959969535413501251999442013082353139290537518086128904993091119534674934202202277050635907008004079788691412782712147797487593510040249832242022835902734939817209358184800954336078838331094308355388211284440290335887813714894626653613586546719
KAT_PUB = 106773687078109007595028366084970322147907086635176067918161636756354740353674098686965493426431314019237945536387044259034050617425729739578628872957481830432099721612688699974185290306098360072264136606623400336518126533605711223527682187548332314997606381158951535480830524587400401856271050333371205030999
KAT_MSG = [1156466847851242602709362303526378170, 191372789510123109308037416804949834, 7204] + [0] * 14


def run_host_rsa(message, sig, mod, wave=False):
    """wave=False: zkwg_rsa_core.h, the phase-sequential restatement; wave=True: zkwg_rsa_wave.h, the algorithm the device
    runs (ballot look-ahead, shuffles, lane-parallel Knuth D, safegcd), on the 64-fiber wavefront of tests/native/wavesim.h"""
    lib = hosttest.load()
    cfg = Config(MAIN_RSA_VERIFIER, 0, 0, 121, 17, 0, 0, 0, 0, 0)
    h = lib.ht_create(C.byref(cfg))
    assert h
    rec = (C.c_uint8 * lib.ht_in_stride(h))()

    def put(field, vals):
        off = lib.ht_in_off(h, field)
        for i, v in enumerate(vals):
            rec[off + 16 * i:off + 16 * i + 16] = list(int(v).to_bytes(16, "little"))

    put(3, mod); put(4, sig); put(5, message)
    bits = (C.c_uint64 * lib.ht_img_bits(h))()
    small = (C.c_uint32 * lib.ht_img_small(h))()
    frv = (C.c_uint8 * (32 * lib.ht_img_fr(h)))()
    small[lib.ht_m_one(h)] = 1
    if wave:
        n = C.c_uint64()
        ok = hosttest.load_wave().wt_run_rsa(C.byref(cfg), rec, None, bits, small, frv, C.byref(n))
        assert 1000 < n.value < 10_000_000      # it really went through the cross-lane exchanges
    else:
        ok = lib.ht_run_rsa(h, rec, None, bits, small, frv)
    wit = hosttest.expand(lib, h, rec, bits, small, frv)
    lib.ht_destroy(h)
    return ok, wit


def oracle_rsa(message, sig, mod):
    from oracle.pyref import zkemail as zk, comp
    main = zk.RSAVerifier65537(121, 17, message, sig, mod, is_main=True)
    return comp.witness_kept(main)


def test_rsa_1024_kat_host_core_matches_oracle():
    ok, wit = run_host_rsa(KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB))
    assert ok == 1
    assert wit == oracle_rsa(KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB))


def test_rsa_wrong_message_fails():
    from oracle.pyref import comp
    m2 = list(KAT_MSG)
    m2[0] += 1
    ok, _ = run_host_rsa(m2, limbs(KAT_SIG), limbs(KAT_PUB))
    assert ok == 0
    with pytest.raises(comp.AssertFailed):
        oracle_rsa(m2, limbs(KAT_SIG), limbs(KAT_PUB))


def test_rsa_2048_synthetic_key():
    from zkwg.synth import test_key, pkcs1_sign_digest
    import hashlib
    key = test_key()
    digest = hashlib.sha256(b"zkwg synthetic header").digest()
    sig = pkcs1_sign_digest(key, digest)
    msg = limbs(int.from_bytes(digest, "big"))
    ok, wit = run_host_rsa(msg, limbs(sig), limbs(key["n"]))
    assert ok == 1
    assert wit == oracle_rsa(msg, limbs(sig), limbs(key["n"]))


def test_safegcd_inverse_matches_fermat():
    # zkwg_fr_inv.h (Bernstein-Yang division steps, 30-bit limbs): the per-lane inversion of zk_rsa
    import random
    lib = hosttest.load()
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    rng = random.Random(1)
    cases = [0, 1, 2, 3, P - 1, P - 2, (P + 1) // 2, 1 << 253, (1 << 121) - 1]
    cases += [rng.randrange(P) for _ in range(1500)] + [rng.randrange(1 << 122) for _ in range(300)]
    cases += [P - rng.randrange(1, 1 << 122) for _ in range(300)]
    for x in cases:
        a = (C.c_uint8 * 32)(*x.to_bytes(32, "little"))
        o = (C.c_uint8 * 32)()
        lib.ht_fr_inv_by(a, o)
        assert int.from_bytes(bytes(o), "little") == (pow(x, P - 2, P) if x else 0), x


def test_device_rsa_algorithm_on_a_simulated_wavefront_matches_oracle():
    """csrc/zkwg_rsa_wave.h -- what zk_rsa runs on the GPU, not the phase-sequential restatement -- compiled for the host on
    a 64-fiber wavefront (tests/native/wavesim.h: ballots, readlane, shuffles as exchange points): the reference's 1,024-bit
    KAT (packages/circuits/tests/rsa.test.ts:64-103; exercises long_div's k-- path and RSAPad's leading zeros), its
    wrong-message negative (:105-144) and a 2,048-bit synthetic signature, every kept signal against the oracle."""
    import hashlib
    from zkwg.synth import test_key, pkcs1_sign_digest
    ok, wit = run_host_rsa(KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB), wave=True)
    assert ok == 1 and wit == oracle_rsa(KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB))
    m2 = list(KAT_MSG)
    m2[0] += 1
    ok, _ = run_host_rsa(m2, limbs(KAT_SIG), limbs(KAT_PUB), wave=True)
    assert ok == 0
    key = test_key()
    digest = hashlib.sha256(b"zkwg synthetic header").digest()
    sig = pkcs1_sign_digest(key, digest)
    msg = limbs(int.from_bytes(digest, "big"))
    ok, wit = run_host_rsa(msg, limbs(sig), limbs(key["n"]), wave=True)
    assert ok == 1 and wit == oracle_rsa(msg, limbs(sig), limbs(key["n"]))
    # and it agrees with the phase-sequential core signal for signal
    ok2, wit2 = run_host_rsa(msg, limbs(sig), limbs(key["n"]))
    assert ok2 == 1 and wit2 == wit
