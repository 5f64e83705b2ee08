"""main = FpMul(n, k) with generic (small) parameters -- packages/circuits/tests/test-circuits/fp-mul-test.circom:5
(`component main = FpMul(2, 4)`), lib/fp.circom:16-81 -- through the product: the reference's own known-answer test
(tests/fp-mul.test.ts:34-46: 17 * 20 mod 85 = 0) and random cases for several (n, k), every kept signal against the
literal Python oracle; host build of the core here, the HIP kernel in the gpu test."""
import ctypes as C
import hashlib
import json
import os
import random

import pytest

import hosttest
from zkwg._lib import Config, MAIN_FP_MUL

PARAMS = [(2, 4), (3, 5), (8, 4), (15, 4), (31, 2), (1, 17), (3, 17)]
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abc_digests.json")))["fp_mul_2_4_kat"]


def _sha(values):
    return hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in values)).hexdigest()


def chunks(x, n, k):
    return [(x >> (n * i)) & ((1 << n) - 1) for i in range(k)]


def oracle_fpmul(n, k, a, b, p):
    from oracle.pyref import zkemail as zk, comp
    main = zk.FpMul(n, k, a, b, p)
    main.is_main = True
    return comp.witness_kept(main), comp.symbols_kept(main)


def run_host(n, k, a, b, p):
    lib = hosttest.load()
    cfg = Config(MAIN_FP_MUL, 0, 0, n, k, 0, 0, 0, 0, 0)
    h = lib.ht_create(C.byref(cfg))
    assert h
    rec = (C.c_uint8 * lib.ht_in_stride(h))()
    for field, vals in ((3, a), (4, b), (5, p)):
        off = lib.ht_in_off(h, field)
        for i, v in enumerate(vals):
            rec[off + 16 * i:off + 16 * i + 16] = list(int(v).to_bytes(16, "little"))
    bits = (C.c_uint64 * lib.ht_img_bits(h))()
    small = (C.c_uint32 * lib.ht_img_small(h))()
    frv = (C.c_uint8 * (32 * lib.ht_img_fr(h)))()
    st = lib.ht_run_fpmul(h, rec, bits, small, frv)
    wit = hosttest.expand(lib, h, rec, bits, small, frv)
    lib.ht_destroy(h)
    return st, wit


def random_case(rng, n, k):
    top = 1 << (n * k)
    p = rng.randrange(1, top)
    kind = rng.randrange(4)
    if kind == 0:
        a, b = rng.randrange(p), rng.randrange(p)
    elif kind == 1:
        a, b = p - 1, p - 1
    elif kind == 2:
        a, b = 0, rng.randrange(p)
    else:
        a, b = rng.randrange(p), 1
    return chunks(a, n, k), chunks(b, n, k), chunks(p, n, k)


def test_reference_kat_fp_mul_2_4_host_core():
    # tests/fp-mul.test.ts:34-46
    a, b, p = [1, 0, 1, 0], [0, 1, 1, 0], [1, 1, 1, 1]
    st, wit = run_host(2, 4, a, b, p)
    want, sym = oracle_fpmul(2, 4, a, b, p)
    assert st == 0 and wit == want
    out = [wit[s] for s, nm in sym if nm.startswith("main.out[")]
    assert out == [0, 0, 0, 0]
    assert _sha(wit) == GOLD["witness_sha256"] and len(wit) == GOLD["witness_len"]      # committed digest of the oracle's witness


def test_layout_names_match_the_oracle_walk():
    import zkwg
    for n, k in PARAMS:
        c = zkwg.Circuit(zkwg.MAIN_FP_MUL, max_header=0, max_body=0, n=n, k=k, device=-1)
        _, sym = oracle_fpmul(n, k, chunks(3, n, k), chunks(2, n, k), chunks(5, n, k))
        assert [(s, nm) for s, nm in c.symbols()] == sym, (n, k)
        assert c.n_public == 0
    for n, k in ((121, 17), (2, 1), (2, 18), (32, 2), (4, 16)):       # not offered: numbers beyond machine words / k out of range
        with pytest.raises(zkwg.ZkwgError):
            zkwg.Circuit(zkwg.MAIN_FP_MUL, max_header=0, max_body=0, n=n, k=k, device=-1)


@pytest.mark.parametrize("n,k", PARAMS)
def test_random_cases_host_core_matches_oracle(n, k):
    rng = random.Random(1000 * n + k)
    for _ in range(12):
        a, b, p = random_case(rng, n, k)
        st, wit = run_host(n, k, a, b, p)
        want, _ = oracle_fpmul(n, k, a, b, p)
        assert st == 0 and wit == want, (n, k, a, b, p)


def test_constraint_system_holds_for_the_oracle_witness_and_catches_a_flip():
    """zkwg.r1cs.fp_mul_main_constraints (the generic-parameter FpMul of lib/fp.circom as R1CS over the layout): every
    oracle witness satisfies it (host checker); flipping any single kept signal is caught."""
    import zkwg
    lib = hosttest.load()
    for n, k in ((2, 4), (15, 4), (3, 17)):
        c = zkwg.Circuit(zkwg.MAIN_FP_MUL, max_header=0, max_body=0, n=n, k=k, device=-1)
        cs = zkwg.WitnessCalculator(c).constraint_system()
        assert cs.n_wires == c.W and cs.n_pub_out == k
        rng = random.Random(5 * n + k)
        for _ in range(4):
            a, b, p = random_case(rng, n, k)
            want, _ = oracle_fpmul(n, k, a, b, p)
            blob = b"".join(int(v).to_bytes(32, "little") for v in want)
            assert lib.ht_r1cs_first_bad(cs.data, len(cs.data), blob) == -1          # (host build of the check core)
        missed = []
        names = dict(c.symbols())
        for slot in range(1, c.W):
            w2 = list(want)
            w2[slot] = (w2[slot] + 1) % (1 << 253)
            if lib.ht_r1cs_first_bad(cs.data, len(cs.data), b"".join(int(v).to_bytes(32, "little") for v in w2)) == -1:
                missed.append(names[slot])
        # what the template itself leaves free: the carry that is declared but never assigned (bigint.circom:76) and the
        # IsZero inverse of a zero difference (in * inv = 1 - out holds for any inv when in = 0)
        for nm in missed:
            assert nm == f"main.tCheck.carry[{2 * k - 2}]" or nm.endswith(".isz.inv"), (n, k, nm)
        assert len(missed) <= 1 + k


def test_inputs_the_template_rejects():
    from oracle.pyref import comp
    # p = 0: long_div divides by zero in the reference; a quotient that does not fit k chunks (a, b >= p) violates tCheck
    assert run_host(2, 4, [1, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0])[0] == 4
    st, _ = run_host(2, 4, [3, 3, 3, 3], [3, 3, 3, 3], [1, 0, 0, 0])
    assert st == 4
    with pytest.raises((comp.AssertFailed, AssertionError, ZeroDivisionError, IndexError)):
        oracle_fpmul(2, 4, [3, 3, 3, 3], [3, 3, 3, 3], [1, 0, 0, 0])
    assert run_host(2, 4, [4, 0, 0, 0], [1, 0, 0, 0], [1, 1, 1, 1])[0] == 4       # a chunk that does not fit n bits


@pytest.mark.gpu
def test_fp_mul_on_the_gpu_matches_oracle_and_constraints():
    """zk_fpmul_small + zk_expand through the C ABI: the reference KAT and random cases of every parameter pair in one
    batch each; statuses, every witness value, and Montgomery-form output."""
    import torch
    import zkwg
    dev = torch.device("cuda", 0)
    for n, k in PARAMS:
        c = zkwg.Circuit(zkwg.MAIN_FP_MUL, max_header=0, max_body=0, n=n, k=k, device=0)
        rng = random.Random(77 * n + k)
        cases = [random_case(rng, n, k) for _ in range(70)]
        if (n, k) == (2, 4):
            cases[0] = ([1, 0, 1, 0], [0, 1, 1, 0], [1, 1, 1, 1])
        bad = len(cases) - 1
        cases[bad] = (chunks(5 % (1 << n * k), n, k), chunks(1, n, k), chunks(0, n, k))       # p = 0
        recs = b"".join(c.pack({"a": a, "b": b, "p": p}) for a, b, p in cases)
        nb = len(cases)
        d_in = torch.frombuffer(bytearray(recs), dtype=torch.uint8).to(dev)
        d_out = torch.full((nb * c.witness_bytes,), 0xA5, dtype=torch.uint8, device=dev)
        d_status = torch.zeros(nb, dtype=torch.int32, device=dev)
        d_scr = torch.empty(c.scratch_bytes(nb), dtype=torch.uint8, device=dev)
        c.calculate_batch_device(d_in, nb, d_out, d_status, d_scr)
        torch.cuda.synchronize()
        st = d_status.cpu().tolist()
        assert st == [0] * bad + [4], (n, k, st)
        raw = d_out.cpu().numpy().tobytes()
        for e, (a, b, p) in enumerate(cases[:bad]):
            w = raw[e * c.witness_bytes:(e + 1) * c.witness_bytes]
            got = [int.from_bytes(w[32 * i:32 * i + 32], "little") for i in range(c.W)]
            want, _ = oracle_fpmul(n, k, a, b, p)
            assert got == want, (n, k, e)
            if (n, k) == (2, 4) and e == 0:
                assert _sha(got) == GOLD["witness_sha256"]
        # the device witnesses satisfy the layout's constraint system (checked on the device), all but the rejected one
        cs = zkwg.WitnessCalculator(c).constraint_system()
        assert cs.first_violations_device(d_out, bad, c.witness_bytes) == [None] * bad
        # Montgomery-form output = value * 2^256 mod r
        d_m = torch.empty(nb * c.witness_bytes, dtype=torch.uint8, device=dev)
        c.prepare_device(d_in, nb, d_status, d_scr, torch.cuda.current_stream())
        c.expand_montgomery_device(d_in, nb, d_scr, 0, nb, d_m, torch.cuda.current_stream())
        torch.cuda.synchronize()
        P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
        m0 = d_m[:c.witness_bytes].cpu().numpy().tobytes()
        w0 = raw[:c.witness_bytes]
        for i in range(c.W):
            assert int.from_bytes(m0[32 * i:32 * i + 32], "little") == int.from_bytes(w0[32 * i:32 * i + 32], "little") * (1 << 256) % P
