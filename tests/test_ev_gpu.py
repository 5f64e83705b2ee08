"""GPU parity: main = EmailVerifier(maxHeader, maxBody, 121, 17, ignoreBodyHashCheck, 0, 0, 0)
(reference: packages/circuits/tests/email-verifier.test.ts, email-verifier-no-body.test.ts) --
HIP witness vs the literal Python oracle, bit-exact; tamper cases must give "Assert Failed"."""
import copy

import pytest

from test_ev_cpu import _inputs, _oracle_ev

pytestmark = pytest.mark.gpu


def _circuit(N, M, ignore):
    import zkwg
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, ignore_body_hash_check=ignore, device=0)
    return c, zkwg.WitnessCalculator(c)


@pytest.mark.parametrize("ignore", [1, 0])
def test_ev_small_batch_bit_exact(ignore):
    import zkwg
    from oracle.pyref import comp
    N, M = 576, 192
    c, wc = _circuit(N, M, ignore)
    ins = [_inputs(N, M, ignore, index=i, body_len=40 + 13 * i) for i in range(5)]
    wits, status = wc.calculateBatch(ins)
    assert status == [0] * 5
    for inp, wb in zip(ins, wits):
        main = _oracle_ev(N, M, ignore, inp)
        w = zkwg.witness_ints(wb)
        assert w == comp.witness_kept(main)
        # outputs first: pubkeyHash, shaHi, shaLo (email-verifier.test.ts:188-207 checks w[1])
        assert (w[1], w[2], w[3]) == main.o
        # the same through the circom_tester surface: assertOut(witness, {pubkeyHash: ...}) (:204-206)
        c.assertOut(w, {"pubkeyHash": main.o[0], "shaHi": str(main.o[1])})
        c.assertOut(wb, {"shaLo": main.o[2], "pubkey": inp["pubkey"]})
        with pytest.raises(AssertionError, match=r"main\.shaHi: expected 5"):
            c.assertOut(w, {"shaHi": 5})
        with pytest.raises(AssertionError, match="Output variable not defined: main.nope"):
            c.assertOut(w, {"nope": 0})


def test_ev_tamper_cases_assert_failed():
    # email-verifier.test.ts:61-186: invalid signature / tampered header / tampered body /
    # non-zero padding / wrong bodyHashIndex each make calculateWitness throw "Assert Failed"
    import zkwg
    from oracle.pyref import comp
    N, M = 576, 192
    c, wc = _circuit(N, M, 0)
    good = _inputs(N, M, 0, index=3, body_len=77)

    def mutated(fn):
        x = copy.deepcopy(good)
        fn(x)
        return x

    def set_(key, idx, val):
        def f(x):
            x[key][idx] = str(val)
        return f

    hlen, blen = int(good["emailHeaderLength"]), int(good["emailBodyLength"])
    bad = [
        mutated(set_("signature", 0, int(good["signature"][0]) ^ 1)),          # :61-79
        mutated(set_("emailHeader", 0, int(good["emailHeader"][0]) ^ 1)),      # :81-102 header tampered
        mutated(set_("emailBody", 0, int(good["emailBody"][0]) ^ 1)),          # :125-146 body tampered
        mutated(set_("emailHeader", hlen + 1, 1)),                             # :104-123 padding not zero
        mutated(set_("emailBody", blen + 1, 1)),                               # :148-166 body padding not zero
        mutated(lambda x: x.__setitem__("bodyHashIndex", str(int(x["bodyHashIndex"]) + 1))),  # :168-186
    ]
    wits, status = wc.calculateBatch([good] + bad)
    assert status[0] == 0
    assert status[1:] == [4] * len(bad)
    for b in bad:
        with pytest.raises(comp.AssertFailed):
            _oracle_ev(N, M, 0, b)
        with pytest.raises(zkwg.ZkwgError, match="Assert Failed"):
            wc.calculateWitness(b)


def test_ev_default_size_one_email_bit_exact():
    # BASELINE.json configs[0]: EmailVerifier(1024, 1536, 121, 17, 0,0,0,0), one email
    import zkwg
    from oracle.pyref import comp
    N, M = 1024, 1536
    c, wc = _circuit(N, M, 0)
    inp = _inputs(N, M, 0, index=11, body_len=1024)
    w = wc.calculateWitness(inp)
    main = _oracle_ev(N, M, 0, inp)
    assert w == comp.witness_kept(main)


def test_ev_with_sha_precompute_selector_bit_exact():
    # email-verifier-with-*-sha-precompute-selector tests: the body prefix is hashed on the host,
    # the circuit starts from the midstate (Sha256BytesPartial with a non-IV preHash)
    import zkwg
    from zkwg import synth, inputs
    from oracle.pyref import comp
    N, M = 576, 320   # remaining body pads to 256 bytes; the length must stay below maxBodyLength
    c, wc = _circuit(N, M, 0)
    d = synth.synthetic_dkim_result(8, 1, body_len=700)
    sel = d["body"][520:532].decode()
    inp = inputs.generate_email_verifier_inputs_from_dkim_result(d, N, M, sha_precompute_selector=sel)
    assert inp["precomputedSHA"] != [str(b) for b in bytes.fromhex("6a09e667bb67ae853c6ef372a54ff53a510e527f9b05688c1f83d9ab5be0cd19")]
    w = wc.calculateWitness(inp)
    assert w == comp.witness_kept(_oracle_ev(N, M, 0, inp))


@pytest.mark.parametrize("variant", ["g16", "lane", "wave"])
def test_pubkey_hash_kernels_agree_with_the_oracle(variant, monkeypatch):
    """PoseidonLarge(121,17) -> Poseidon(9) (utils/hash.circom:15-39) has three device kernels: 16 lanes per email
    (zk_poseidon9_g16, the default from 1,024 emails), one lane per email (round 2's) and one wavefront per email
    (small batches).  Each is forced here on a 7-email batch with different keys; the 420 S-box signals and
    pubkeyHash sit inside the witness that is compared with the oracle."""
    import zkwg
    from oracle.pyref import comp
    if variant == "g16":
        monkeypatch.setenv("ZKWG_POS_WAVE_BELOW", "0")
    elif variant == "lane":
        monkeypatch.setenv("ZKWG_POS_WAVE_BELOW", "0")
        monkeypatch.setenv("ZKWG_POS_LANE", "1")
    from oracle.pyref import poseidon as pos
    N, M = 576, 192
    c, wc = _circuit(N, M, 1)
    ins = [_inputs(N, M, 1, index=10 + i, body_len=50) for i in range(7)]
    for i, inp in enumerate(ins[3:], 1):    # emails 3..6: other moduli (the signature no longer verifies; the hash is still defined)
        inp["pubkey"] = [str((int(v) + 977 * i * (k + 1)) % (1 << 121)) for k, v in enumerate(inp["pubkey"])]
    wits, status = wc.calculateBatch(ins)
    hashes = set()
    for n, (inp, wb, st) in enumerate(zip(ins, wits, status)):
        w = zkwg.witness_ints(wb)
        limbs = [int(v) for v in inp["pubkey"]]
        merged = [limbs[2 * k] + (limbs[2 * k + 1] << 121) for k in range(8)] + [limbs[16]]
        assert w[1] == pos.poseidon_hash(merged), (variant, n)      # pubkeyHash is the first output
        hashes.add(w[1])
        if n < 3:
            assert st == 0 and w == comp.witness_kept(_oracle_ev(N, M, 1, inp))   # incl. the 420 S-box signals
        else:
            assert st == 4
    assert len(hashes) == 5
