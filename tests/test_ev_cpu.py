"""CPU tests for the EmailVerifier main: layout/symbol table vs the oracle, host builds of the
Poseidon core and the BodyHashRegex scanner vs the oracle / Python `re`."""
import ctypes as C
import random

import hosttest


def _oracle_ev(N, M, ignore, inp):
    from oracle.pyref import zkemail as zk
    return zk.EmailVerifier(N, M, 121, 17, ignore, inp, body_hash_regex=lambda m: zk.BodyHashRegexV1(N, m))


def _inputs(N, M, ignore, index=0, body_len=100):
    from zkwg import synth, inputs
    d = synth.synthetic_dkim_result(7, index, body_len=body_len)
    return inputs.generate_email_verifier_inputs_from_dkim_result(d, N, M, ignore_body_hash_check=bool(ignore))


def test_ev_layout_matches_oracle_with_and_without_body():
    import zkwg
    from oracle.pyref import comp
    for ignore in (1, 0):
        N, M = 576, 192
        main = _oracle_ev(N, M, ignore, _inputs(N, M, ignore))
        sym = comp.symbols_kept(main)
        c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, ignore_body_hash_check=ignore, device=-1)
        assert c.W == len(sym)
        got = c.symbols()
        assert got == sym
        assert c.n_public == 20


def test_poseidon_host_core_matches_oracle():
    from oracle.pyref import zkemail as zk, comp
    from zkwg.synth import test_key
    lib = hosttest.load()
    n = test_key()["n"]
    limbs = [(n >> (121 * i)) & ((1 << 121) - 1) for i in range(17)]
    buf = b"".join(x.to_bytes(16, "little") for x in limbs)
    out = (C.c_uint8 * (32 * 420))()
    h = (C.c_uint8 * 32)()
    lib.ht_poseidon(buf, out, h)
    pl = zk.PoseidonLarge(121, 17, limbs)
    kept = [v for _, v, k in pl.walk("x") if comp.is_kept(k)]
    got = [int.from_bytes(bytes(out[32 * i:32 * i + 32]), "little") for i in range(420)]
    assert got == kept
    assert int.from_bytes(bytes(h), "little") == pl.o


def test_poseidon_known_vectors():
    # poseidon([1,2]) is the globally known circomlib vector; [1,2,3,4] is circomlibjs' own test
    from oracle.pyref import poseidon
    assert poseidon.poseidon_hash([1, 2]) == 7853200120776062878684798364095072458815029376092732009249414926327459813530
    assert poseidon.poseidon_hash([1, 2, 3, 4]) == 18821383157269793795438455681495246036402687001665670618754263018637548127333


def test_dfa_circuit_agrees_with_python_re_on_fuzzed_headers():
    """The DFA circuit (zkwg v1) and the interface-level restatement (Python `re`) reveal the same bytes."""
    from oracle.pyref import zkemail as zk
    from zkwg import synth
    rng = random.Random(9)
    N = 320
    for i in range(8):
        hdr = bytearray(synth.synthetic_dkim_result(4, i, body_len=60)["headers"][-300:])
        if i % 3 == 1:
            hdr[rng.randrange(len(hdr))] = rng.randrange(256)
        if i % 3 == 2:
            hdr = b"\r\ndkim-signature:a=b; bh=QUJD; x\r\n" + hdr[:250]
        msg = list(bytes(hdr[:N]).ljust(N, b"\0"))
        a, b = zk.BodyHashRegexV1(N, msg), zk.BodyHashRegex(N, msg)
        assert a.o == b.o


def test_regex_scanner_vs_python_re():
    from oracle.pyref import zkemail as zk
    lib = hosttest.load()
    rng = random.Random(5)
    from zkwg import synth
    N = 576
    cases = []
    for i in range(40):
        hdr = bytearray(synth.synthetic_dkim_result(3, i, body_len=80)["headers"])
        mode = i % 8
        if mode == 1:
            hdr[rng.randrange(len(hdr))] = rng.randrange(256)
        elif mode == 2:
            k = hdr.find(b"bh=")
            hdr[k + 10] = ord(";")
        elif mode == 3:
            hdr = hdr.replace(b"dkim-signature:", b"dkim-signaturE:")
        elif mode == 4:
            hdr = hdr.replace(b"; bh=", b";bh=")
        elif mode == 5:
            hdr = b"dkim-signature:a=b; bh=QUJD; x" + hdr[:300]
        elif mode == 6:
            hdr = bytearray(rng.randrange(256) for _ in range(400))
        cases.append(bytes(hdr[:N]).ljust(N, b"\0"))
    for ci, msg in enumerate(cases):
        rev = (C.c_uint32 * N)()
        own = (C.c_uint32 * (2 * (N + 1) + 3 * N))()
        n = lib.ht_regex_scan(msg, N, rev, own)
        o = zk.BodyHashRegex(N, list(msg))           # interface semantics (Python `re`)
        assert (1 if n else 0) == o.o[0]
        assert list(rev) == o.o[1]
        if ci % 10 == 0:
            v1 = zk.BodyHashRegexV1(N, list(msg))    # the DFA circuit restated literally
            assert list(rev) == v1.o[1] and (1 if n else 0) == v1.o[0]


def test_poseidon_sparse_host_core_matches_oracle():
    # the product's sparse-partial-round Poseidon (t = 3 and t = 17, PoseidonModular's two arities)
    # against the textbook rounds of the oracle: hash and every kept Sigma signal
    import ctypes as C
    import random
    from oracle.pyref import poseidon
    from oracle.pyref.comp import witness_kept
    lib = hosttest.load()
    rng = random.Random(5)
    for t in (3, 17):
        for trial in range(3):
            xs = [rng.randrange(256) for _ in range(t - 1)] if trial == 0 else [rng.randrange(poseidon.P) for _ in range(t - 1)]
            if trial == 2:
                xs = [poseidon.P - 1] * (t - 1)       # the largest inputs: the lazy ranges of zkwg_poseidon29.h at their widest
            comp = poseidon.Poseidon(t - 1, xs)
            kept = witness_kept(comp)[1:]
            n = 3 * (8 * t + poseidon.N_ROUNDS_P[t - 2])
            assert len(kept) == n
            inp = b"".join(x.to_bytes(32, "little") for x in xs)
            emit = C.create_string_buffer(32 * n)
            h = C.create_string_buffer(32)
            # 4 x 64-bit words | 9 x 29-bit limbs (zk_rslb_chunks' evaluator) in each of its variants
            fns = [lambda *a: lib.ht_poseidon_sparse(*a)] + [(lambda *a, v=v: lib.ht_poseidon29(a[0], v, *a[1:])) for v in ((0, 2, 3, 7) if t == 3 else range(8))]
            for fn in fns:
                emit = C.create_string_buffer(32 * n)
                h = C.create_string_buffer(32)
                assert fn(t, inp, emit, h) == 0
                assert int.from_bytes(h.raw, "little") == comp.o
                got = [int.from_bytes(emit.raw[32 * i:32 * i + 32], "little") for i in range(n)]
                assert got == kept
    # every column addition, operand limb and packed result of the limb-form runs above stayed inside the ranges its header argues
    assert lib.ht_p29_violations() == 0
