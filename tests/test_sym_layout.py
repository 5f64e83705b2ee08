"""`.sym`-driven witness order (include/zkwg.h zkwg_circuit_create_sym; SURVEY.md 8b3 "layout source =
builtin or path to .sym", 8c5): the compiled circuit's own `.sym` decides the witness index of every
signal.  No circom artifact exists offline, so the files here are synthetic -- the kept-v1 names with
shuffled blocks, eliminated signals and unrelated label indices (tests/hosttest.py synthetic_sym)."""
import ctypes as C

import pytest

import hosttest


def test_sym_layout_rsa_main_on_the_host_core():
    # product schedule builder + RSA core (host build) + Python mirror of zk_expand, `.sym` order
    import zkwg
    from zkwg._lib import Config, MAIN_RSA_VERIFIER
    from test_rsa_cpu import KAT_MSG, KAT_SIG, KAT_PUB, limbs, oracle_rsa
    lib = hosttest.load()
    c0 = zkwg.Circuit(MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1)
    text, dst = hosttest.synthetic_sym(c0.symbols(), c0.n_public, seed=3)
    cfg = Config(MAIN_RSA_VERIFIER, 0, 0, 121, 17, 0, 0, 0, 0, 0)
    h = lib.ht_create_sym(C.byref(cfg), text.encode(), len(text))
    assert h
    exp = hosttest.apply_sym(oracle_rsa(KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB)), dst)
    assert lib.ht_W(h) == len(exp) < c0.W
    rec = (C.c_uint8 * lib.ht_in_stride(h))()
    for field, vals in ((3, limbs(KAT_PUB)), (4, limbs(KAT_SIG)), (5, KAT_MSG)):
        off = lib.ht_in_off(h, field)
        for i, v in enumerate(vals):
            rec[off + 16 * i:off + 16 * i + 16] = list(int(v).to_bytes(16, "little"))
    bits = (C.c_uint64 * lib.ht_img_bits(h))()
    small = (C.c_uint32 * lib.ht_img_small(h))()
    frv = (C.c_uint8 * (32 * lib.ht_img_fr(h)))()
    small[lib.ht_m_one(h)] = 1
    assert lib.ht_run_rsa(h, rec, None, bits, small, frv) == 1
    assert hosttest.expand(lib, h, rec, bits, small, frv) == exp
    lib.ht_destroy(h)


def test_sym_layout_handle_symbols_and_errors():
    import zkwg
    c0 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1)
    sym0 = c0.symbols()
    text, dst = hosttest.synthetic_sym(sym0, c0.n_public, seed=5)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1, sym=text)
    kept = [(d, sym0[s][1]) for s, d in enumerate(dst) if d is not None]
    assert c.W == len(kept) < c0.W
    assert c.symbols() == sorted(kept)
    assert c.n_public == c0.n_public and c.symbols()[1][1] == "main.pubkeyHash"
    # identity file -> identical layout
    ident = "".join(f"{s},{s},0,{n}\n" for s, n in sym0[1:])
    assert zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1, sym=ident).symbols() == sym0
    # a signal the schedule cannot produce, a hole, a duplicate index
    with pytest.raises(zkwg.ZkwgError, match="not produced by this schedule .first: main.sha.compression.x"):
        zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1, sym=ident + "9,99,0,main.sha.compression.x\n")
    with pytest.raises(zkwg.ZkwgError, match="is not covered"):
        zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1, sym=ident.replace(f"\n7,7,0,", "\n7,-1,0,", 1))
    two = ident.split("\n")
    a = two[10].split(",", 3)
    two[11] = ",".join([two[11].split(",", 3)[0], a[1], "0", two[11].split(",", 3)[3]])
    with pytest.raises(zkwg.ZkwgError, match="assigned to two signals|is not covered"):
        zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1, sym="\n".join(two))
    # rename rules for compiler-generated component names
    theirs = ident.replace("main.anon_Sha256Bytes.", "main.Sha256Bytes_66_2570.")
    with pytest.raises(zkwg.ZkwgError, match="not produced"):
        zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1, sym=theirs)
    c3 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1, sym=theirs,
                      sym_alias="main.anon_Sha256Bytes.=main.Sha256Bytes_66_2570.\n")
    assert c3.W == c0.W and any(n.startswith("main.Sha256Bytes_66_2570.") for _, n in c3.symbols())


@pytest.mark.gpu
def test_sym_layout_on_gpu_bit_exact():
    import zkwg
    from oracle import coracle
    from test_ev_cpu import _inputs
    N, M = 576, 192
    c0 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=-1)
    inps = [_inputs(N, M, 0, index=i, body_len=60 + 20 * i) for i in range(3)]
    owit, ostatus, W = coracle.calculate(0, N, M, 0, inps, threads=3)
    assert ostatus == [0, 0, 0] and W == c0.W
    for seed, drop, blk in ((1, 0.01, 400), (2, 0.0, 3), (3, 0.3, 50000)):
        text, dst = hosttest.synthetic_sym(c0.symbols(), c0.n_public, seed=seed, drop_frac=drop, max_block=blk)
        c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, sym=text)
        recs = b"".join(c.pack(i) for i in inps)
        wit, status = c.calculate_batch_host(recs)
        assert status == [0, 0, 0]
        wb = c.witness_bytes
        for e in range(3):
            src = owit[e]
            exp = bytearray(wb)
            for s, d in enumerate(dst):
                if d is not None:
                    exp[32 * d:32 * d + 32] = src[32 * s:32 * s + 32]
            assert wit[e * wb:(e + 1) * wb] == bytes(exp), (seed, e)
        # .wtns written in the file's order
        blob = zkwg.WitnessCalculator(c).calculateWTNSBin(inps[0])
        assert blob[:4] == b"wtns" and int.from_bytes(blob[60:64], "little") == c.W and len(blob) == 76 + 32 * c.W
