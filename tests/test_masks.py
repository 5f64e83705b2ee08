"""Flag variants enableHeaderMasking / enableBodyMasking (email-verifier.circom:97-105,158-166;
reference tests email-verifier-with-header-mask.test.ts, email-verifier-with-body-mask.test.ts)."""
import pytest


def _masked_inputs(N, M, index=0):
    from test_ev_cpu import _inputs
    inp = _inputs(N, M, 0, index=index, body_len=90)
    inp["headerMask"] = [1 if 25 < i < 50 else 0 for i in range(N)]   # the mask of the reference's test
    inp["bodyMask"] = [1 if i % 3 == 0 else 0 for i in range(M)]
    return inp


def _oracle(N, M, inp):
    from oracle.pyref import zkemail as zk
    return zk.EmailVerifier(N, M, 121, 17, 0, inp, body_hash_regex=lambda m: zk.BodyHashRegexV1(N, m),
                            enableHeaderMasking=1, enableBodyMasking=1)


def test_mask_layout_and_c_oracle_match_literal_oracle():
    import zkwg
    from oracle import coracle
    from oracle.pyref import comp
    N, M = 576, 192
    inp = _masked_inputs(N, M)
    main = _oracle(N, M, inp)
    sym = comp.symbols_kept(main)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=-1, enable_header_masking=1, enable_body_masking=1)
    assert c.W == len(sym) and c.symbols() == sym and c.n_public == 3 + N + M + 17
    wits, status, W = coracle.calculate(0, N, M, 0, [inp])
    assert status == [0] and W == c.W
    w = [int.from_bytes(wits[0][i:i + 32], "little") for i in range(0, len(wits[0]), 32)]
    assert w == comp.witness_kept(main)
    # maskedHeader / maskedBody are main outputs right after shaLo
    hdr = [int(x) for x in inp["emailHeader"]]
    assert w[4:4 + N] == [hdr[i] if 25 < i < 50 else 0 for i in range(N)]


@pytest.mark.gpu
def test_masks_on_gpu_bit_exact_and_non_binary_mask_rejected():
    import copy
    import zkwg
    from oracle.pyref import comp
    N, M = 576, 192
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, enable_header_masking=1, enable_body_masking=1)
    wc = zkwg.WitnessCalculator(c)
    inp = _masked_inputs(N, M, index=2)
    w = wc.calculateWitness(inp)
    assert w == comp.witness_kept(_oracle(N, M, inp))
    hdr = [int(x) for x in inp["emailHeader"]]
    assert w[4:4 + N] == [hdr[i] if 25 < i < 50 else 0 for i in range(N)]   # assertOut(maskedHeader)
    bad = copy.deepcopy(inp)
    bad["bodyMask"][5] = 2                                                    # AssertBit fails
    with pytest.raises(zkwg.ZkwgError, match="Assert Failed"):
        wc.calculateWitness(bad)
