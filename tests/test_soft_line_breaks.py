"""Flag variant removeSoftLineBreaks (email-verifier.circom:148-156, helpers/remove-soft-line-breaks.circom;
reference tests remove-soft-line-breaks.test.ts -- its 7 known answers are in test_oracle_kats.py -- and
email-verifier-with-soft-line-breaks.test.ts)."""
import pytest

N, M = 576, 384


def _inputs(index=0, body_len=300):
    from zkwg import synth, inputs as gen
    d = synth.synthetic_dkim_result(11, index, body_len, soft_breaks=True)
    inp = gen.generate_email_verifier_inputs_from_dkim_result(d, N, M, remove_soft_line_breaks_flag=True)
    return inp


def _oracle(inp):
    from oracle.pyref import zkemail as zk
    return zk.EmailVerifier(N, M, 121, 17, 0, inp, body_hash_regex=lambda m: zk.BodyHashRegexV1(N, m),
                            removeSoftLineBreaks=1)


def test_soft_line_break_layout_and_c_oracle_match_literal_oracle():
    import zkwg
    from oracle import coracle
    from oracle.pyref import comp
    inp = _inputs()
    assert bytes(int(b) for b in inp["emailBody"]).count(b"=\r\n") >= 1
    main = _oracle(inp)
    sym = comp.symbols_kept(main)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=-1, remove_soft_line_breaks=1)
    assert c.W == len(sym) and c.symbols() == sym and c.n_public == 3 + 17
    wits, status, W = coracle.calculate(0, N, M, 0, [inp])
    assert status == [0] and W == c.W
    assert wits[0] == b"".join(v.to_bytes(32, "little") for v in comp.witness_kept(main))
    # a wrong decoded body: the RLC comparison fails -> Assert Failed in both tiers
    bad = dict(inp)
    bad["decodedEmailBodyIn"] = list(inp["decodedEmailBodyIn"])
    bad["decodedEmailBodyIn"][7] = str(int(bad["decodedEmailBodyIn"][7]) ^ 1)
    with pytest.raises(comp.AssertFailed):
        _oracle(bad)
    assert coracle.calculate(0, N, M, 0, [bad], want_witness=False)[1] == [4]


def test_helper_remove_soft_line_breaks():
    # input-generators.ts:107-126 doc example + position map
    from zkwg import inputs as gen
    clean, pmap = gen.remove_soft_line_breaks(bytes([72, 101, 108, 108, 111, 61, 13, 10, 87, 111, 114, 108, 100]))
    assert list(clean) == [72, 101, 108, 108, 111, 87, 111, 114, 108, 100, 0, 0, 0]
    assert pmap == {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 8, 6: 9, 7: 10, 8: 11, 9: 12}
    assert gen.get_adjusted_selector(b"Hel=\r\nlo", "Hello", *gen.remove_soft_line_breaks(b"Hel=\r\nlo")) == "Hel=\r\nlo"


@pytest.mark.gpu
def test_soft_line_breaks_on_gpu_bit_exact():
    import zkwg
    from oracle import coracle
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, remove_soft_line_breaks=1)
    wc = zkwg.WitnessCalculator(c)
    inps = [_inputs(i, 150 + 40 * i) for i in range(5)]
    assert sum(bytes(int(b) for b in i["emailBody"]).count(b"=\r\n") for i in inps) >= 3
    bad = dict(inps[4])
    bad["decodedEmailBodyIn"] = list(bad["decodedEmailBodyIn"])
    bad["decodedEmailBodyIn"][3] = "65" if bad["decodedEmailBodyIn"][3] != "65" else "66"
    inps.append(bad)
    recs = b"".join(c.pack(i) for i in inps)
    wit, status = c.calculate_batch_host(recs)
    owit, ostatus, W = coracle.calculate(0, N, M, 0, inps, threads=4)
    assert W == c.W
    assert status == ostatus == [0, 0, 0, 0, 0, 4]
    wb = c.witness_bytes
    for i in range(5):
        got, exp = wit[i * wb:(i + 1) * wb], owit[i]
        if got != exp:
            first = next(k for k in range(c.W) if got[32 * k:32 * k + 32] != exp[32 * k:32 * k + 32])
            raise AssertionError(f"email {i}: first differing slot {first} ({dict(c.symbols()).get(first)})")
    # the reference-shaped entry point on one email
    w = wc.calculateWitness(inps[0])
    assert b"".join(v.to_bytes(32, "little") for v in w) == owit[0]
    with pytest.raises(zkwg.ZkwgError, match="Assert Failed"):
        wc.calculateWitness(bad)


@pytest.mark.gpu
def test_constant_chunks_give_the_image_of_hashed_chunks(monkeypatch):
    """zkwg_kernels_rslb.hip "constant chunks": the all-zero 16-byte chunks of the padded halves take the precomputed signals of
    Poseidon(16)(0, ..., 0) instead of a lane of zk_rslb_chunks.  Bodies from a few bytes (nearly every chunk constant) to the longest the
    circuit takes (the encoded half has none); same witnesses as with ZKWG_RSLB_CONST_CHUNKS=0 (every unit hashed) and as the C oracle"""
    import zkwg
    from oracle import coracle
    inps = [_inputs(i, n) for i, n in enumerate((5, 17, 160, 250, 318))]
    recs = None
    wits = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("ZKWG_RSLB_CONST_CHUNKS", flag)
        c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, remove_soft_line_breaks=1)
        recs = recs or b"".join(c.pack(i) for i in inps)
        wits[flag], status = c.calculate_batch_host(recs)
        assert status == [0] * len(inps)
        c.close()
    assert wits["1"] == wits["0"]
    owit, ostatus, W = coracle.calculate(0, N, M, 0, inps, threads=4)
    assert ostatus == [0] * len(inps) and b"".join(owit) == wits["1"]


@pytest.mark.gpu
def test_standard_form_pipeline_fits_the_scratch_without_montgomery_copies():
    """zkwg_scratch_bytes_standard (include/zkwg.h): prepare + expand into a buffer that ends where the Montgomery-copy area would begin --
    half the bytes for this circuit -- leave the guard behind it untouched and give the witnesses of the full-size buffer and the C oracle"""
    import torch
    import zkwg
    from oracle import coracle
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, remove_soft_line_breaks=1)
    inps = [_inputs(i, 90 + 50 * i) for i in range(4)]
    n = len(inps)
    std, full = c.scratch_bytes(n, montgomery=False), c.scratch_bytes(n)
    assert 256 < std < 0.6 * full and c.scratch_bytes(n, montgomery=True) == full
    dev = torch.device("cuda:0")
    d_in = torch.frombuffer(bytearray(b"".join(c.pack(i) for i in inps)), dtype=torch.uint8).view(n, c.in_stride).to(dev)
    st = torch.cuda.current_stream()
    outs = []
    for size in (std, full):
        d_scr = torch.full((size + 4096,), 0xA5, dtype=torch.uint8, device=dev)
        d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        d_out = torch.empty(n * c.witness_bytes, dtype=torch.uint8, device=dev)
        c.prepare_device(d_in, n, d_st, d_scr, st)
        c.expand_device(d_in, n, d_scr, 0, n, d_out, st)
        torch.cuda.synchronize()
        assert d_st.tolist() == [0] * n
        assert bool((d_scr[size:] == 0xA5).all())
        outs.append(bytes(d_out.cpu().numpy()))
    assert outs[0] == outs[1]
    owit, ostatus, W = coracle.calculate(0, N, M, 0, inps, threads=4)
    assert ostatus == [0] * n and b"".join(owit) == outs[0]


@pytest.mark.gpu
def test_soft_line_break_edge_patterns_on_gpu():
    """Signed emails whose bodies put "=\\r\\n" where the circuit's index arithmetic has its corners
    (helpers/remove-soft-line-breaks.circom:47-91 and the reference's own cases in remove-soft-line-breaks.test.ts:
    at the beginning, at the end, consecutive, incomplete sequences), every slot against the C oracle."""
    import zkwg
    from zkwg import synth, inputs as gen
    from oracle import coracle
    bodies = [
        b"=\r\nhello\r\n",                                  # soft break at the very beginning
        b"hello=\r\n",                                       # ... at the very end (next byte is the 0x80 pad)
        b"ab=\r\n=\r\n=\r\ncd\r\n",                          # consecutive
        b"a==\r\nb=\rx=\ny\r=\n=\r\r\n=\r\n\r\n",            # incomplete sequences around real ones
        b"=" * 40 + b"\r\n" + b"x=\r" * 10 + b"\n\r\n",         # runs of '=' and "=\r"
        (b"0123456789" * 7 + b"=\r\n") * 4 + b"end\r\n",       # QP-style long lines
        b"no soft breaks at all\r\n",
    ]
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, remove_soft_line_breaks=1)
    inps = []
    for i, b in enumerate(bodies):
        d = synth.synthetic_dkim_result(21, i, body=b)
        inps.append(gen.generate_email_verifier_inputs_from_dkim_result(d, N, M, remove_soft_line_breaks_flag=True))
    # the reference helper must produce the compaction the circuit accepts
    recs = b"".join(c.pack(i) for i in inps)
    wit, status = c.calculate_batch_host(recs)
    owit, ostatus, W = coracle.calculate(0, N, M, 0, inps, threads=4)
    assert status == ostatus == [0] * len(bodies)
    wb = c.witness_bytes
    for i in range(len(bodies)):
        assert wit[i * wb:(i + 1) * wb] == owit[i], f"body {i}"
    # decoded shifted by one byte / a soft break left in place: rejected by both
    bad = []
    for i in (0, 2, 5):
        b2 = dict(inps[i])
        dec = list(b2["decodedEmailBodyIn"])
        b2["decodedEmailBodyIn"] = ["0"] + dec[:-1]
        bad.append(b2)
        b3 = dict(inps[i])
        b3["decodedEmailBodyIn"] = list(b3["emailBody"])
        bad.append(b3)
    _, st = c.calculate_batch_host(b"".join(c.pack(i) for i in bad), want_witness=False)
    assert st == coracle.calculate(0, N, M, 0, bad, threads=4, want_witness=False)[1] == [4] * len(bad)


def test_all_flags_together_layout_and_oracle_tiers():
    # enableHeaderMasking + enableBodyMasking + removeSoftLineBreaks in one circuit: input / component order
    import zkwg
    from oracle import coracle
    from oracle.pyref import zkemail as zk, comp
    inp = dict(_inputs(3, 200))
    inp["headerMask"] = [1 if 10 < i < 90 else 0 for i in range(N)]
    inp["bodyMask"] = [i % 2 for i in range(M)]
    main = zk.EmailVerifier(N, M, 121, 17, 0, inp, body_hash_regex=lambda m: zk.BodyHashRegexV1(N, m),
                            enableHeaderMasking=1, enableBodyMasking=1, removeSoftLineBreaks=1)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=-1, enable_header_masking=1,
                     enable_body_masking=1, remove_soft_line_breaks=1)
    assert c.symbols() == comp.symbols_kept(main)
    assert len(c.pack(inp)) == c.in_stride
    wits, status, W = coracle.calculate(0, N, M, 0, [inp])
    assert status == [0] and W == c.W
    assert wits[0] == b"".join(v.to_bytes(32, "little") for v in comp.witness_kept(main))


@pytest.mark.gpu
def test_all_flags_together_on_gpu():
    import zkwg
    from oracle import coracle
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, enable_header_masking=1,
                     enable_body_masking=1, remove_soft_line_breaks=1)
    inps = []
    for i in range(3):
        inp = dict(_inputs(5 + i, 180 + 30 * i))
        inp["headerMask"] = [1 if (i + 3) * 7 < k < 200 else 0 for k in range(N)]
        inp["bodyMask"] = [(k + i) % 2 for k in range(M)]
        inps.append(inp)
    wit, status = c.calculate_batch_host(b"".join(c.pack(i) for i in inps))
    owit, ostatus, W = coracle.calculate(0, N, M, 0, inps, threads=3)
    assert W == c.W and status == ostatus == [0, 0, 0]
    wb = c.witness_bytes
    assert [wit[i * wb:(i + 1) * wb] == owit[i] for i in range(3)] == [True] * 3


def test_device_merge_chain_on_a_simulated_wavefront_matches_the_oracle():
    """csrc/zkwg_rslb_wave.h -- the body of the kernel zk_rslb_merge: the Poseidon(2) merge chain of PoseidonModular
    (utils/hash.circom:76-80) with 4 lanes per email, Montgomery-form state and lane 3 converting the S-box signals -- compiled
    for the host on the 64-fiber wavefront of tests/native/wavesim.h (the workgroup barrier is an exchange point).  20 emails
    (one full wavefront of 16 + a partial one), 4 chunk digests each: r and all 3 x 243 S-box signals per email against the
    oracle's textbook Poseidon(2)."""
    import ctypes as C
    import random
    import hosttest
    from oracle.pyref import poseidon
    from oracle.pyref.comp import witness_kept
    lib = hosttest.load_wave()
    rng = random.Random(11)
    n, nch = 20, 4
    P16, P2 = 612, 243
    off = lambda c: 0 if c == 0 else P16 + (c - 1) * (P16 + P2)          # zk_rs_chunk_off
    f_chunk, f_hash = 5, 16
    img_fr = f_hash + off(nch - 1) + P16 + P2 + 3
    dig = [[rng.randrange(poseidon.P) for _ in range(nch)] for _ in range(n)]
    buf = bytearray(n * img_fr * 32)
    for e in range(n):
        for c in range(nch):
            o = 32 * (e * img_fr + f_chunk + c)
            buf[o:o + 32] = dig[e][c].to_bytes(32, "little")
    raw = (C.c_uint8 * len(buf)).from_buffer(buf)
    ex = C.c_uint64()
    assert lib.wt_run_rslb_merge(n, nch, raw, img_fr, f_chunk, f_hash, C.byref(ex)) == 0
    assert ex.value > 500           # barriers = exchange points: the lanes really ran in lockstep
    get = lambda e, i: int.from_bytes(buf[32 * (e * img_fr + i):32 * (e * img_fr + i) + 32], "little")
    for e in range(n):
        out = dig[e][0]
        for c in range(1, nch):
            comp = poseidon.Poseidon(2, [out, dig[e][c]])
            kept = witness_kept(comp)[1:]
            assert len(kept) == P2
            base = f_hash + off(c) + P16
            assert [get(e, base + i) for i in range(P2)] == kept, (e, c)
            out = comp.o
        assert get(e, f_chunk) == out                                     # r, where zk_rslb_scan reads it
        assert [get(e, f_chunk + c) for c in range(1, nch)] == dig[e][1:]  # the other digests are untouched
