"""`.r1cs` reader + `checkConstraints` (include/zkwg.h zkwg_r1cs_*).  No compiled circuit exists
offline, so the constraint systems are written by tests/r1cs_util.py in the iden3 binary format: circomlib-style
Num2Bits, a chain of Poseidon S-boxes, and large random satisfiable systems; expected results come from the
pure-Python evaluator in the same file."""
import ctypes as C

import json
import os

import pytest

import hosttest
import r1cs_util as ru


def _blob(w):
    return b"".join(int(v).to_bytes(32, "little") for v in w)


def test_r1cs_reader_and_check_core_on_the_host():
    import zkwg
    lib = hosttest.load()
    nw, cons, w = ru.num2bits_system(8, 0xA5)
    for header_last in (False, True):
        data = ru.write_r1cs(nw, cons, n_pub_out=0, n_pub_in=1, n_prv_in=0, header_last=header_last)
        r = zkwg.R1cs(data, device=-1)   # parse only
        assert (r.n_wires, r.n_constraints, r.n_pub_in, r.n_labels) == (nw, len(cons), 1, nw)
        with pytest.raises(zkwg.ZkwgError):
            r.first_violations(_blob(w))   # no device: the check has no CPU fallback
    # the same check core (zkwg_r1cs.h), host build
    for (nw, cons, w) in (ru.num2bits_system(8, 0xA5), ru.sigma_chain_system(12345, 20)):
        data = ru.write_r1cs(nw, cons)
        assert lib.ht_r1cs_first_bad(data, len(data), _blob(w)) == -1
        for k in (1, nw // 2, nw - 1):
            w2 = list(w)
            w2[k] = (w2[k] + 1) % ru.P
            exp = ru.first_violation(cons, w2)
            assert exp is not None and lib.ht_r1cs_first_bad(data, len(data), _blob(w2)) == exp
    cons, w = ru.random_system(7, 300, 2000)
    data = ru.write_r1cs(300, cons)
    assert ru.first_violation(cons, w) is None and lib.ht_r1cs_first_bad(data, len(data), _blob(w)) == -1
    w2 = list(w); w2[17] = (w2[17] + 5) % ru.P
    assert lib.ht_r1cs_first_bad(data, len(data), _blob(w2)) == ru.first_violation(cons, w2)
    # malformed files
    for bad in (b"r1cx" + data[4:], data[:40], data[:12] + b"\xff" * 12 + data[24:]):
        with pytest.raises(zkwg.ZkwgError):
            zkwg.R1cs(bad, device=-1)
    # a non-reduced witness value is a violation, not an accident
    w3 = list(w); w3[17] = w[17] + ru.P if w[17] + ru.P < (1 << 256) else w[17]
    if w3[17] != w[17]:
        exp = next(i for i, (a, b, c) in enumerate(cons) if 17 in a or 17 in b or 17 in c)
        assert lib.ht_r1cs_first_bad(data, len(data), _blob(w3)) == exp


@pytest.mark.gpu
def test_check_constraints_on_gpu():
    import torch
    import zkwg
    nw, cons, w = ru.sigma_chain_system(987654321, 200)
    r = zkwg.R1cs(ru.write_r1cs(nw, cons), device=0)
    r.checkConstraints(w)                                   # circom_tester-shaped: silent when satisfied
    bad = list(w); bad[5] = (bad[5] + 1) % ru.P
    with pytest.raises(zkwg.ZkwgError, match="Constraint doesn't match"):
        r.checkConstraints(bad)
    with pytest.raises(zkwg.ZkwgError, match="Invalid witness length"):
        r.checkConstraints(w[:-1])
    # a large random system, a batch of witnesses with one corruption each, host and device entry points
    NW, M = 5000, 60000
    cons, w = ru.random_system(11, NW, M)
    r = zkwg.R1cs(ru.write_r1cs(NW, cons), device=0)
    wits, exp = [], []
    import random
    rng = random.Random(3)
    for e in range(6):
        we = list(w)
        if e:
            k = rng.randrange(1, NW)
            we[k] = (we[k] + 1 + rng.randrange(100)) % ru.P
        wits.append(_blob(we))
        exp.append(ru.first_violation(cons, we))
    assert exp[0] is None
    assert r.first_violations(b"".join(wits)) == exp
    stride = 32 * NW + 64                                    # padded stride
    d = torch.zeros(6 * stride, dtype=torch.uint8, device="cuda:0")
    for e in range(6):
        d[e * stride:e * stride + 32 * NW] = torch.frombuffer(bytearray(wits[e]), dtype=torch.uint8).to("cuda:0")
    assert r.first_violations_device(d, 6, stride) == exp


@pytest.mark.gpu
def test_check_constraints_of_real_device_witnesses():
    """The device witness against constraint systems derived independently of the witness kernels:
    (1) EmailVerifier: the complete Poseidon(9) pubkey-hash block; (2) Sha256Bytes main: booleanity of every
    slot that the segment table types as a bit (Num2Bits / xor / comparator outputs: 97 % of the witness)."""
    import torch
    import zkwg
    from zkwg._lib import Config, MAIN_SHA256_BYTES
    from test_ev_cpu import _inputs
    from conftest import sha_pad
    N, M = 576, 192
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    from zkwg import r1cs as zr
    cons = zr.poseidon9_constraints(c.symbols())
    assert len(cons) == 3 * 140 + 1
    r = zkwg.R1cs(ru.write_r1cs(c.W, cons, n_pub_out=3, n_pub_in=17), device=0)
    recs = b"".join(c.pack(_inputs(N, M, 0, index=i, body_len=80)) for i in range(3))
    d_in = torch.frombuffer(bytearray(recs), dtype=torch.uint8).to("cuda:0")
    d_out = torch.empty(3 * c.witness_bytes, dtype=torch.uint8, device="cuda:0")
    d_status = torch.zeros(3, dtype=torch.int32, device="cuda:0")
    d_scr = torch.empty(c.scratch_bytes(3), dtype=torch.uint8, device="cuda:0")
    c.calculate_batch_device(d_in, 3, d_out, d_status, d_scr)
    torch.cuda.synchronize()
    assert d_status.cpu().tolist() == [0, 0, 0]
    assert r.first_violations_device(d_out, 3, c.witness_bytes) == [None, None, None]
    sl = dict((n, s) for s, n in c.symbols())["main.anon_PoseidonLarge.anon_Poseidon.pEx.sigmaP[30].in4"]
    d_out[c.witness_bytes + 32 * sl] ^= 1                                   # flip one bit of witness 1
    assert r.first_violations_device(d_out, 3, c.witness_bytes) == [None, 3 * (4 * 10 + 30) + 1, None]

    # (2) booleanity over every bit-typed slot of a Sha256Bytes(128) witness
    lib = hosttest.load()
    cfg = Config(MAIN_SHA256_BYTES, 128, 0, 121, 17, 0, 0, 0, 0, 0)
    h = lib.ht_create(C.byref(cfg))
    segs = lib.ht_segs(h)
    bit_slots = []
    for i in range(lib.ht_nsegs(h)):
        s = segs[i]
        if s.type in (2, 3, 4, 5, 9, 11, 14):
            bit_slots.extend(range(s.slot, s.slot + s.nslots))
    W = lib.ht_W(h)
    lib.ht_destroy(h)
    assert len(bit_slots) > 0.95 * W
    cs = zkwg.Circuit(MAIN_SHA256_BYTES, max_header=128, max_body=0, device=0)
    assert cs.W == W
    rb = zkwg.R1cs(ru.write_r1cs(W, [({k: 1}, {k: 1, 0: ru.P - 1}, {}) for k in bit_slots]), device=0)
    p, n = sha_pad(b"hello world", 128)
    w = zkwg.WitnessCalculator(cs).calculateWitness({"paddedIn": list(p), "paddedInLength": n})
    rb.checkConstraints(w)
    k = len(bit_slots) // 2
    w[bit_slots[k]] = 2
    with pytest.raises(zkwg.ZkwgError, match=f"constraint {k}\\)"):
        rb.checkConstraints(w)


def test_derived_constraint_systems_hold_for_the_oracle_witnesses():
    # zkwg.r1cs: COMPLETE constraint systems of the two small mains, derived from the
    # reference / circomlib template definitions with aliases resolved onto the kept wires: every wire is
    # constrained (except the 17 declared-but-unassigned carry[32] of CheckCarryToZero) and the oracle's
    # witnesses satisfy all of them
    import zkwg
    from zkwg import r1cs as zr
    from conftest import sha_pad
    from oracle import coracle
    from test_rsa_cpu import KAT_MSG, KAT_SIG, KAT_PUB, limbs, oracle_rsa
    c = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=128, max_body=0, device=-1)
    cons = zr.sha256_main_constraints(c.symbols(), 128)
    assert len(cons) == 65742
    used = set()
    for a, b, cc in cons:
        used |= set(a) | set(b) | set(cc)
    assert used == set(range(c.W))
    for msg in (b"", b"hello world", b"x" * 100):
        p, n = sha_pad(msg, 128)
        w, st, W = coracle.calculate(1, 128, 0, 0, [{"paddedIn": list(p), "paddedInLength": n}])
        wi = [int.from_bytes(w[0][32 * i:32 * i + 32], "little") for i in range(W)]
        assert st == [0] and ru.first_violation(cons, wi) is None
    wi[30000] ^= 1
    assert ru.first_violation(cons, wi) is not None
    cr = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1)
    rcons = zr.rsa_main_constraints(cr.symbols())
    assert len(rcons) == 191654
    used = set()
    for a, b, cc in rcons:
        used |= set(a) | set(b) | set(cc)
    names = dict(cr.symbols())
    assert sorted(names[s] for s in set(range(cr.W)) - used) == sorted(
        [f"main.bigPow.doublers[{i}].tCheck.carry[32]" for i in range(16)] + ["main.bigPow.adder.tCheck.carry[32]"])
    w = oracle_rsa(KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB))      # packages/circuits/tests/rsa.test.ts:64-103
    assert ru.first_violation(rcons, w) is None
    w[100000] = (w[100000] + 1) % ru.P
    assert ru.first_violation(rcons, w) is not None
    # the module's own Poseidon constants / interpolation matrix against the oracle's
    from oracle.pyref import poseidon, bigint_func as bf
    Cc, M = zr.poseidon_constants(10)
    Co, Mo = poseidon.constants(10)
    assert Cc == Co and M == Mo
    T = zr.RsaBuilder({}, "x").interp_matrix(33)
    for x in (0, 7, 32):
        col = bf.poly_interp(33, [1 if y == x else 0 for y in range(33)])
        assert [T[i][x] for i in range(33)] == [v % ru.P for v in col[:33]] and not any(v % ru.P for v in col[33:])


def test_complete_email_verifier_constraint_system_holds_for_the_oracle_witness():
    # zkwg.r1cs: EmailVerifier(576,192): 753,807 constraints, every wire constrained except the 17 unassigned
    # carry[32]; the oracle witness of a valid email satisfies all of them, the .r1cs file parses back
    import zkwg
    from zkwg import r1cs as zr
    from test_ev_cpu import _inputs
    from oracle import coracle
    N, M = 576, 192
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=-1)
    cons = zr.email_verifier_constraints(c.symbols(), N, M)
    assert len(cons) == 753807
    used = set()
    for a, b, cc in cons:
        used |= set(a) | set(b) | set(cc)
    names = dict(c.symbols())
    assert sorted(names[s] for s in set(range(c.W)) - used) == sorted(
        [f"main.rsaVerifier.bigPow.doublers[{i}].tCheck.carry[32]" for i in range(16)] + ["main.rsaVerifier.bigPow.adder.tCheck.carry[32]"])
    w, st, W = coracle.calculate(0, N, M, 0, [_inputs(N, M, 0, index=1, body_len=100)])
    wi = [int.from_bytes(w[0][32 * i:32 * i + 32], "little") for i in range(W)]
    assert st == [0] and ru.first_violation(cons, wi) is None
    r = zkwg.R1cs(zr.email_verifier_r1cs(c.symbols(), N, M), device=-1)
    assert (r.n_wires, r.n_constraints, r.n_pub_out, r.n_pub_in, r.n_prv_in) == (c.W, 753807, 3, 17, N + 1 + 17 + 1 + 32 + M + 1)


def _device_witnesses(c, inputs):
    import torch
    n = len(inputs)
    recs = b"".join(c.pack(i) for i in inputs)
    d_in = torch.frombuffer(bytearray(recs), dtype=torch.uint8).to("cuda:0")
    # poison the output first: a slot the kernels failed to write would read as 0xA5A5... (>= r) and be reported
    d_out = torch.full((n * c.witness_bytes,), 0xA5, dtype=torch.uint8, device="cuda:0")
    d_status = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device="cuda:0")
    c.calculate_batch_device(d_in, n, d_out, d_status, d_scr)
    torch.cuda.synchronize()
    assert d_status.cpu().tolist() == [0] * n
    return d_out


def _flip_and_expect(r, cons, c, d_out, n, e, slot):
    """flip the lowest bit of wire `slot` of witness e: that witness alone must be rejected, at the constraint the
    pure-Python evaluator names"""
    d_out[e * c.witness_bytes + 32 * slot] ^= 1
    we = d_out[e * c.witness_bytes:(e + 1) * c.witness_bytes].cpu().numpy().tobytes()
    exp = ru.first_violation(cons, [int.from_bytes(we[32 * i:32 * i + 32], "little") for i in range(c.W)])
    assert exp is not None
    assert r.first_violations_device(d_out, n, c.witness_bytes) == [exp if i == e else None for i in range(n)]
    d_out[e * c.witness_bytes + 32 * slot] ^= 1   # restore


@pytest.mark.gpu
def test_check_constraints_complete_sha_and_rsa_mains_on_device_witnesses():
    """`checkConstraints` with the complete derived systems of `Sha256Bytes(128)` (65,742 constraints) and
    `RSAVerifier65537(121,17)` (191,654) on witnesses straight from the device kernels."""
    import zkwg
    from zkwg import r1cs as zr
    from conftest import sha_pad
    from zkwg import synth
    import hashlib
    from test_rsa_cpu import KAT_MSG, KAT_SIG, KAT_PUB, limbs
    c = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=128, max_body=0, device=0)
    cons = zr.sha256_main_constraints(c.symbols(), 128)
    r = zkwg.R1cs(ru.write_r1cs(c.W, cons, n_pub_out=256, n_pub_in=129), device=0)
    inputs = []
    for m in (b"", b"abc", b"hello world", bytes(range(64)), b"q" * 119):
        p, n = sha_pad(m, 128)
        inputs.append({"paddedIn": list(p), "paddedInLength": n})
    d_out = _device_witnesses(c, inputs)
    assert r.first_violations_device(d_out, 5, c.witness_bytes) == [None] * 5
    _flip_and_expect(r, cons, c, d_out, 5, 2, dict((n, s) for s, n in c.symbols())["main.sha.sha256compression[1].t1[20].ch.out[7]"])

    cr = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0)
    rcons = zr.rsa_main_constraints(cr.symbols())
    rr = zkwg.R1cs(ru.write_r1cs(cr.W, rcons, n_pub_out=0, n_pub_in=17), device=0)
    key = synth.test_key()                                           # 2048-bit synthetic key, PKCS#1 v1.5 signature
    dig = hashlib.sha256(b"zkwg").digest()
    sig = synth.pkcs1_sign_digest(key, dig)
    msg = limbs(int.from_bytes(dig, "big"))
    inputs = [{"message": KAT_MSG, "signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB)},
              {"message": msg, "signature": limbs(sig), "modulus": limbs(key["n"])}]
    d_out = _device_witnesses(cr, inputs)
    assert rr.first_violations_device(d_out, 2, cr.witness_bytes) == [None, None]
    _flip_and_expect(rr, rcons, cr, d_out, 2, 1, dict((n, s) for s, n in cr.symbols())["main.bigPow.doublers[7].tCheck.carry[11]"])


@pytest.mark.gpu
def test_check_constraints_email_verifier_on_device_witnesses():
    """EmailVerifier(576,192) device witnesses against the COMPLETE constraint system of the kept-v1 layout
    (zkwg.r1cs, 753,807 constraints derived from the circuit definition)."""
    import zkwg
    from zkwg import r1cs as zr
    from test_ev_cpu import _inputs
    N, M = 576, 192
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    cons = zr.email_verifier_constraints(c.symbols(), N, M)
    assert len(cons) == 753807
    r = zkwg.R1cs(ru.write_r1cs(c.W, cons, n_pub_out=3, n_pub_in=17), device=0)
    n = 4
    d_out = _device_witnesses(c, [_inputs(N, M, 0, index=i, body_len=40 + 20 * i) for i in range(n)])
    assert r.first_violations_device(d_out, n, c.witness_bytes) == [None] * n
    names = dict((nm, s) for s, nm in c.symbols())
    _flip_and_expect(r, cons, c, d_out, n, 3, names["main.anon_Sha256BytesPartial.sha.sha256compression[2].suma[63].out[5]"])
    _flip_and_expect(r, cons, c, d_out, n, 1, names["main.rsaVerifier.bigPow.adder.v_pq_r[9]"])
    _flip_and_expect(r, cons, c, d_out, n, 0, names["main.anon_BodyHashRegex.and[7][200].out"])
    _flip_and_expect(r, cons, c, d_out, n, 2, names["main.anon_Base64Decode.translate[3][1].sum_az"])


def _flag_inputs(N, M, index_from=0):
    """a valid all-flags input whose body contains a soft line break"""
    from zkwg import synth, inputs as gen
    for index in range(index_from, index_from + 50):
        d = synth.synthetic_dkim_result(31, index, 100, soft_breaks=True)
        if b"=\r\n" in d["body"]:
            inp = gen.generate_email_verifier_inputs_from_dkim_result(d, N, M, remove_soft_line_breaks_flag=True)
            inp["headerMask"] = [1 if 10 < i < 90 else 0 for i in range(N)]
            inp["bodyMask"] = [(i + index) % 2 for i in range(M)]
            return inp, index
    raise AssertionError("no soft break generated")


def test_constraint_system_without_body_hash_check_holds_for_the_oracle_witness():
    # ignoreBodyHashCheck = 1 (tests/test-circuits/email-verifier-no-body-test.circom)
    import zkwg
    from zkwg import r1cs as zr
    from test_ev_cpu import _inputs
    from oracle import coracle
    N = 576
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=0, device=-1, ignore_body_hash_check=1)
    cons = zr.email_verifier_constraints(c.symbols(), N, 0, ignore_body_hash_check=1)
    used = set()
    for a, b, cc in cons:
        used |= set(a) | set(b) | set(cc)
    assert len(set(range(c.W)) - used) == 17
    w, st, W = coracle.calculate(0, N, 0, 1, [_inputs(N, 192, 1, index=0, body_len=60)])
    wi = [int.from_bytes(w[0][32 * i:32 * i + 32], "little") for i in range(W)]
    assert st == [0] and W == c.W and ru.first_violation(cons, wi) is None


def test_constraint_system_with_all_template_flags_holds_for_the_oracle_witness():
    # enableHeaderMasking + enableBodyMasking + removeSoftLineBreaks (PoseidonModular, RLC sums): complete system
    import zkwg
    from zkwg import r1cs as zr
    from oracle import coracle
    N, M = 576, 192
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=-1, enable_header_masking=1,
                     enable_body_masking=1, remove_soft_line_breaks=1)
    cons = zr.email_verifier_constraints(c.symbols(), N, M, 1, 1, 1)
    used = set()
    for a, b, cc in cons:
        used |= set(a) | set(b) | set(cc)
    names = dict(c.symbols())
    assert all("tCheck.carry[32]" in names[s] for s in set(range(c.W)) - used) and len(set(range(c.W)) - used) == 17
    inp, _ = _flag_inputs(N, M)
    w, st, W = coracle.calculate(0, N, M, 0, [inp])
    wi = [int.from_bytes(w[0][32 * i:32 * i + 32], "little") for i in range(W)]
    assert st == [0] and W == c.W and ru.first_violation(cons, wi) is None


@pytest.mark.gpu
def test_check_constraints_all_template_flags_on_device_witnesses():
    import zkwg
    from zkwg import r1cs as zr
    N, M = 576, 192
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, enable_header_masking=1,
                     enable_body_masking=1, remove_soft_line_breaks=1)
    cons = zr.email_verifier_constraints(c.symbols(), N, M, 1, 1, 1)
    r = zkwg.R1cs(zr.write_r1cs(c.W, cons, n_pub_out=3 + N + M, n_pub_in=17), device=0)
    inps, nxt = [], 0
    for _ in range(3):
        inp, idx = _flag_inputs(N, M, nxt)
        inps.append(inp)
        nxt = idx + 1
    d_out = _device_witnesses(c, inps)
    assert r.first_violations_device(d_out, 3, c.witness_bytes) == [None] * 3
    names = dict((nm, s) for s, nm in c.symbols())
    _flip_and_expect(r, cons, c, d_out, 3, 1, names["main.qpEncodingChecker.rHasher.anon_Poseidon_merge[9].pEx.sigmaP[20].in2"])
    _flip_and_expect(r, cons, c, d_out, 3, 2, names["main.qpEncodingChecker.sumDec[50]"])


@pytest.mark.gpu
def test_witness_calculator_check_constraints_like_circom_tester():
    # the reference's test idiom: witness = await circuit.calculateWitness(input); await circuit.checkConstraints(witness)
    import zkwg
    from conftest import sha_pad
    wc = zkwg.WitnessCalculator(zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=128, max_body=0, device=0))
    p, n = sha_pad(b"hello world", 128)
    w = wc.calculateWitness({"paddedIn": list(p), "paddedInLength": n})
    wc.checkConstraints(w)
    w[40000] ^= 1
    with pytest.raises(zkwg.ZkwgError, match="Constraint doesn't match"):
        wc.checkConstraints(w)


@pytest.mark.gpu
def test_single_wire_corruptions_are_all_detected():
    """Sanity check against under-constrained wires in the derived system: in a valid EmailVerifier(576,192) device
    witness change ONE wire (+1) -- 384 wires sampled over the whole witness plus the first/last wires of every
    top-level block -- and check each of the 400-odd corrupted copies on the device: every one must violate a
    constraint.  Exempt by the circuit's own definition: the 17 declared-but-unassigned carry[32], and `IsZero.inv`
    when the tested value is zero (circomlib: `inv <-- in != 0 ? 1/in : 0; out <== -in*inv + 1` leaves inv free there)."""
    import random
    import torch
    import zkwg
    from zkwg import r1cs as zr
    from test_ev_cpu import _inputs
    N, M = 576, 192
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    sym = c.symbols()
    cons = zr.email_verifier_constraints(sym, N, M)
    r = zkwg.R1cs(zr.write_r1cs(c.W, cons, n_pub_out=3, n_pub_in=17), device=0)
    base = _device_witnesses(c, [_inputs(N, M, 0, index=2, body_len=90)])
    exempt = {s for s, n in sym if n.endswith("tCheck.carry[32]")}
    rng = random.Random(123)
    picks = set(rng.sample(range(1, c.W), 384))
    prev_top = None
    for s, n in sym[1:]:                      # block boundaries: first wire of every top-level component / array
        top = n.split(".")[1].split("[")[0] if "." in n else n
        if top != prev_top:
            picks.update({s, max(1, s - 1)})
            prev_top = top
    picks = sorted(picks - exempt)
    k = len(picks)
    wb = c.witness_bytes
    d = base.view(1, wb).repeat(k, 1).contiguous()
    idx = torch.arange(k, device="cuda:0")
    off = torch.tensor([32 * s for s in picks], device="cuda:0")
    low = d[idx, off].to(torch.int32)
    d[idx, off] = ((low + 1) % 256).to(torch.uint8)          # +1 on the low byte (no carry needed for a change)
    got = r.first_violations_device(d.view(-1), k, wb)
    missed = [dict(sym)[picks[i]] for i in range(k) if got[i] is None]
    assert all(n.endswith(".inv") for n in missed), [n for n in missed if not n.endswith(".inv")][:10]
    # ... and those really are inverses of a zero: the matching `.out` wire of each is 1
    w0 = base.cpu().numpy().tobytes()
    slot_of = {n: s for s, n in sym}
    for n in missed:
        o = slot_of[n[:-4] + ".out"]
        assert int.from_bytes(w0[32 * o:32 * o + 32], "little") == 1, n
    assert len(missed) < k // 4


@pytest.mark.gpu
def test_prover_first_stage_evaluations_on_device_witnesses():
    """zkwg_r1cs_evaluate_device: A.w, B.w, C.w of every constraint -- what `groth16.prove` computes first from the
    witness (second half of fullProve, packages/helpers/src/chunked-zkey.ts:80) -- on device-resident witnesses, in
    standard form and, for a witness in Montgomery form, in Montgomery form (no conversion pass)."""
    import random
    import torch
    import zkwg
    R = 1 << 256
    # (1) a random system against Python integers, both forms
    NW, M = 3000, 20000
    cons, w = ru.random_system(21, NW, M)
    r = zkwg.R1cs(ru.write_r1cs(NW, cons), device=0)
    ev = lambda d, ww: sum(c * ww[k] for k, c in d.items()) % ru.P
    exp = [[ev(t[j], w) for t in cons] for j in range(3)]
    stride = 32 * NW
    d_std = torch.frombuffer(bytearray(_blob(w)), dtype=torch.uint8).to("cuda:0")
    d_mont = torch.frombuffer(bytearray(_blob([v * R % ru.P for v in w])), dtype=torch.uint8).to("cuda:0")
    for d, scale, mont in ((d_std, 1, False), (d_mont, R, True), (d_mont, R, False)):
        out = bytes(r.evaluate_device(d, 1, stride, montgomery=mont).cpu().numpy().tobytes())
        got = [int.from_bytes(out[32 * i:32 * i + 32], "little") for i in range(3 * M)]
        assert got == [v * scale % ru.P for j in range(3) for v in exp[j]]
    # (2) real witnesses: RSAVerifier65537 main, kept-v1 constraint system, witness written in Montgomery form by the
    # fused expand; a * b = c holds for every constraint of every email (in Montgomery form: aR * bR = cR * R)
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0)
    cs = zkwg.WitnessCalculator(c).constraint_system()
    from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG, limbs
    rec = c.pack({"signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB), "message": KAT_MSG})
    n = 3
    d_in = torch.frombuffer(bytearray(rec * n), dtype=torch.uint8).to("cuda:0")
    d_status = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device="cuda:0")
    d_wit = torch.empty(n * c.witness_bytes, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream()
    c.prepare_device(d_in, n, d_status, d_scr, s)
    c.expand_montgomery_device(d_in, n, d_scr, 0, n, d_wit, s)
    abc = cs.evaluate_device(d_wit, n, c.witness_bytes, s, montgomery=True)
    torch.cuda.synchronize()
    assert d_status.cpu().tolist() == [0] * n
    m = cs.n_constraints
    raw = abc[1].cpu().numpy().tobytes()
    rng = random.Random(2)
    for i in [0, 1, m - 1] + [rng.randrange(m) for _ in range(300)]:
        a, b, cc = (int.from_bytes(raw[32 * (j * m + i):32 * (j * m + i) + 32], "little") for j in range(3))
        assert a * b % ru.P == cc * R % ru.P, i
    assert bytes(abc[0].cpu().numpy().tobytes()) == raw == bytes(abc[2].cpu().numpy().tobytes())
    assert any(raw[32 * i:32 * i + 32] != bytes(32) for i in range(m))


@pytest.mark.gpu
@pytest.mark.parametrize("main", ["rsa", "email", "email_flags"])
def test_prover_first_stage_from_the_compact_image(main):
    """zkwg_circuit_attach_r1cs + zkwg_expand_abc_device: A.w | B.w | C.w written from the prepared image (descriptors +
    integer / field rows, no 32-byte witness read) are byte-identical to zkwg_r1cs_evaluate_device on the expanded
    witness, in standard and in Montgomery form; the witness of the same handle is unchanged by the attachment."""
    import torch
    import zkwg
    dev = torch.device("cuda", 0)
    if main == "rsa":
        mk = lambda: zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0)
        from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG, limbs
        c0 = mk()
        rec = c0.pack({"signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB), "message": KAT_MSG})
        n = 3
        d_in = torch.frombuffer(bytearray(rec * n), dtype=torch.uint8).to(dev)
    elif main == "email_flags":
        # header / body masking and removeSoftLineBreaks = 1 (the merge chain on its side stream): the row kernels then run
        # from the expand call
        mk = lambda: zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=0, enable_header_masking=1,
                                  enable_body_masking=1, remove_soft_line_breaks=1)
        c0 = mk()
        inps, nxt = [], 0
        for _ in range(3):
            inp, idx = _flag_inputs(576, 192, nxt)
            inps.append(inp)
            nxt = idx + 1
        n = len(inps)
        d_in = torch.frombuffer(bytearray(b"".join(c0.pack(i) for i in inps)), dtype=torch.uint8).to(dev)
    else:
        from zkwg import synth
        mk = lambda: zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=0)
        c0 = mk()
        n = 11
        recs, _ = synth.packed_batch(c0, seed=0xABC, n=n, body_len=60)
        d_in = torch.frombuffer(bytearray(recs), dtype=torch.uint8).to(dev)
    cs = zkwg.WitnessCalculator(c0).constraint_system()
    s = torch.cuda.current_stream()

    def witnesses(c, mont):
        d_status = torch.zeros(n, dtype=torch.int32, device=dev)
        d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device=dev)
        d_wit = torch.empty(n * c.witness_bytes, dtype=torch.uint8, device=dev)
        c.prepare_device(d_in, n, d_status, d_scr, s)
        (c.expand_montgomery_device if mont else c.expand_device)(d_in, n, d_scr, 0, n, d_wit, s)
        torch.cuda.synchronize()
        assert d_status.cpu().tolist() == [0] * n
        return d_wit, d_scr

    c1 = mk()
    before = c1.scratch_bytes(n)
    c1.attach_r1cs(cs)
    assert c1.abc_bytes == 96 * cs.n_constraints and c1.scratch_bytes(n) > before
    with pytest.raises(zkwg.ZkwgError):
        c1.attach_r1cs(cs)                     # one system per handle
    for mont in (False, True):
        w0, _ = witnesses(c0, mont)
        w1, scr = witnesses(c1, mont)
        assert torch.equal(w0, w1)
        want = cs.evaluate_device(w0, n, c0.witness_bytes, s, montgomery=mont)
        got = torch.full((n, c1.abc_bytes + 64), 0xEE, dtype=torch.uint8, device=dev)   # (padded stride: the tail stays untouched)
        c1.expand_abc_device(d_in, n, scr, 0, n, got, s, montgomery=mont, out_stride=c1.abc_bytes + 64)
        torch.cuda.synchronize()
        assert torch.equal(got[:, :c1.abc_bytes], want)
        assert bool((got[:, c1.abc_bytes:] == 0xEE).all())
        if main == "rsa":
            # the committed digest of the reference's 1,024-bit RSA known answer (tests/golden/abc_digests.json: Python integers
            # over the oracle's witness, no product code involved)
            import hashlib
            gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abc_digests.json")))["rsa_kat"]
            assert hashlib.sha256(got[0, :c1.abc_bytes].cpu().numpy().tobytes()).hexdigest() == gold["montgomery" if mont else "standard"]
        if main == "email_flags":
            # all template flags (masks + removeSoftLineBreaks): digest from Python integers over the ORACLE's witness of the first
            # input (tests/golden/make_abc_digests.py "ev_flags_576_192_kept"), not from zk_r1cs_eval
            import hashlib
            gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abc_digests.json")))["ev_flags_576_192_kept"]
            assert cs.n_constraints == gold["constraints"]
            if not mont:
                assert hashlib.sha256(w0[:c0.witness_bytes].cpu().numpy().tobytes()).hexdigest() == gold["witness_sha256"]
            assert hashlib.sha256(got[0, :c1.abc_bytes].cpu().numpy().tobytes()).hexdigest() == gold["montgomery" if mont else "standard"]
        # a sub-range of the batch
        part = torch.empty((2, c1.abc_bytes), dtype=torch.uint8, device=dev)
        c1.expand_abc_device(d_in, n, scr, 1, 2, part, s, montgomery=mont)
        torch.cuda.synchronize()
        assert torch.equal(part, want[1:3])
        if not mont:
            # the same from a host copy of the device-prepared image (zkwg_expand_abc_host: only the image crosses PCIe)
            import ctypes as C
            h_scr = bytearray(scr.cpu().numpy().tobytes())
            h_ptr = (C.c_uint8 * len(h_scr)).from_buffer(h_scr)
            recs_h = bytes(d_in.cpu().numpy().tobytes())
            host = c1.expand_abc_host(recs_h, n, h_ptr, 1, 2, rows_on_host=False)
            assert host == bytes(want[1:3].cpu().numpy().tobytes())
            if main == "rsa":
                host2 = c1.expand_abc_host(recs_h, n, h_ptr, 0, 1, rows_on_host=True)     # rows recomputed on the host: same bytes
                assert host2 == bytes(want[0:1].cpu().numpy().tobytes())
    assert any(int(x) for x in want[0, :4096].cpu().tolist())


def test_attach_r1cs_host_logic():
    """zkwg_circuit_attach_r1cs on a layout-only handle: the tables are built (the image grows by the row results), a
    second system, a system over another layout and a fully numbered handle are refused, and without a device the
    expansion itself reports NO_DEVICE instead of falling back to anything."""
    import zkwg
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1)
    cs = zkwg.WitnessCalculator(c).constraint_system()
    assert c.abc_bytes == 0
    lay0 = c.image_layout(4)
    c.attach_r1cs(cs)
    lay1 = c.image_layout(4)
    assert c.abc_bytes == 96 * cs.n_constraints
    assert lay1["small_words"] > lay0["small_words"] and lay1["fr_elems"] > lay0["fr_elems"] and lay1["bits_words"] == lay0["bits_words"]
    assert c.scratch_bytes(4) == lay1["total_bytes"]
    with pytest.raises(zkwg.ZkwgError):
        c.attach_r1cs(cs)
    other = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=64, max_body=0, device=-1)
    with pytest.raises(zkwg.ZkwgError):
        other.attach_r1cs(cs)                      # wire count of another layout
    with pytest.raises(zkwg.ZkwgError):
        other.attach_r1cs(b"not an r1cs file")
    rc = c.lib.zkwg_expand_abc_device(c.h, 1, 1, 256, 0, 1, 0, 1, c.abc_bytes, None)
    assert rc == -3        # ZKWG_RC_NO_DEVICE (include/zkwg.h)


def _abc_on_the_host(c, cons, rec, run_core):
    """A.w | B.w | C.w through zkwg_expand_abc_host with the row tables evaluated on the host, from the image the host
    build of the compute core leaves; returns (got, want-from-oracle-free Python evaluation over `witness`)."""
    import ctypes as C
    lay = c.image_layout(1)
    raw = (C.c_uint8 * (lay["total_bytes"] + 256))()
    base = (-C.addressof(raw)) % 256
    at = lambda off: C.c_void_p(C.addressof(raw) + base + off)
    small = (C.c_uint32 * lay["small_words"]).from_address(at(lay["off_small"]).value)
    run_core(at(lay["off_bits"]), at(lay["off_small"]), at(lay["off_fr"]), small)
    out = c.expand_abc_host(rec, 1, C.c_void_p(C.addressof(raw) + base), 0, 1, rows_on_host=True)
    m = len(cons)
    assert len(out) == 96 * m
    return [int.from_bytes(out[32 * i:32 * i + 32], "little") for i in range(3 * m)]


def test_prover_first_stage_tables_evaluated_on_the_host():
    """zk_o0_build's tables for an attached constraint system (descriptors, run-compressed terms, one-word / two-word
    integer rows, chains, field rows, pre-decoded slots) checked without a GPU: zkwg_expand_abc_host evaluates them with the
    kernels' own decode functions (csrc/zkwg_o0_dec.h) over the image the host build of the RSA / FpMul core produces;
    expected = the combinations evaluated in Python integers over the oracle's witness."""
    import ctypes as C
    import zkwg
    from zkwg import r1cs as zr
    from zkwg._lib import Config, MAIN_RSA_VERIFIER, MAIN_FP_MUL
    from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG, limbs, oracle_rsa
    lib = hosttest.load()
    ev = lambda d, w: sum(cf * w[k] for k, cf in d.items()) % ru.P

    GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abc_digests.json")))

    def check(c, cons, rec, run_core, witness, golden=None):
        c.attach_r1cs(zr.write_r1cs(c.W, cons))
        got = _abc_on_the_host(c, cons, rec, run_core)
        want = [ev(t[j], witness) for j in range(3) for t in cons]
        bad = [i for i in range(len(want)) if got[i] != want[i]]
        assert not bad, (len(bad), bad[:5])
        if golden is not None:       # tests/golden/abc_digests.json (make_abc_digests.py: Python integers over the oracle's witness)
            import hashlib
            assert hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in got)).hexdigest() == GOLD[golden]["standard"]
            assert GOLD[golden]["constraints"] == len(cons)
        assert any(v > (1 << 200) for v in want) and any(0 < v < 1000 for v in want)     # negative / field values and small ones

    # RSAVerifier65537(121,17): the reference's 1,024-bit known answer (rsa.test.ts:64-103)
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1)
    cons = zr.rsa_main_constraints(c.symbols())
    rec = c.pack({"signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB), "message": KAT_MSG})
    h = lib.ht_create(C.byref(Config(MAIN_RSA_VERIFIER, 0, 0, 121, 17, 0, 0, 0, 0, 0)))

    def rsa_core(bits, small_p, frv, small):
        small[lib.ht_m_one(h)] = 1
        assert lib.ht_run_rsa(h, rec, None, bits, small_p, frv) == 1
    check(c, cons, rec, rsa_core, oracle_rsa(KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB)), golden="rsa_kat")
    lib.ht_destroy(h)
    # FpMul(3,17) and the reference's FpMul(2,4) known answer (fp-mul.test.ts:34-46)
    from test_fpmul import chunks, oracle_fpmul
    for n, k, a, b, p in ((2, 4, [1, 0, 1, 0], [0, 1, 1, 0], [1, 1, 1, 1]), (3, 17, chunks(2 ** 50 - 3, 3, 17), chunks(2 ** 49 + 12345, 3, 17), chunks(2 ** 51 - 129, 3, 17))):
        c = zkwg.Circuit(zkwg.MAIN_FP_MUL, max_header=0, max_body=0, n=n, k=k, device=-1)
        cons = zr.fp_mul_main_constraints(c.symbols(), n, k)
        rec = c.pack({"a": a, "b": b, "p": p})
        h = lib.ht_create(C.byref(Config(MAIN_FP_MUL, 0, 0, n, k, 0, 0, 0, 0, 0)))
        check(c, cons, rec, lambda bits, small_p, frv, small: lib.ht_run_fpmul(h, rec, bits, small_p, frv), oracle_fpmul(n, k, a, b, p)[0],
              golden="fp_mul_2_4_kat" if (n, k) == (2, 4) else None)
        lib.ht_destroy(h)


def test_numbered_handle_takes_the_compilers_r1cs_for_the_prover_stage_on_the_host():
    """a zkey is keyed to the COMPILED `.r1cs` (chunked-zkey.ts:80-84), whose wires are the file's --
    aliases, constants and linear signals included.  zkwg_circuit_attach_r1cs on a numbered handle (zkwg_circuit_create_full)
    substitutes every wire of every combination by its kept-v1 source(s) and builds the same descriptor / row tables.
    Checked without a GPU on rsa-test.circom's interpreter-generated `.sym` + `.r1cs` (205,713 wires, 208,463 constraints):
    the tables evaluated by the kernels' decode functions over the host-built image against Python integers over the complete
    witness -- whose digest is the interpreter's -- with the file parsed by an independent reader (tests/r1cs_util.read_r1cs)."""
    import ctypes as C
    import gzip
    import hashlib
    import zkwg
    from zkwg._lib import Config, MAIN_RSA_VERIFIER
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "o0_rsa")
    meta = json.load(open(base + ".json"))
    sym, r1cs = gzip.open(base + ".sym.gz", "rb").read(), gzip.open(base + ".r1cs.gz", "rb").read()
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1, sym=sym, sym_alias=meta["alias"], r1cs=r1cs)
    lay0 = c.image_layout(1)
    lib = hosttest.load()
    h = lib.ht_create(C.byref(Config(MAIN_RSA_VERIFIER, 0, 0, 121, 17, 0, 0, 0, 0, 0)))
    rec = c.pack(meta["inputs"])

    def image():
        lay = c.image_layout(1)
        raw = (C.c_uint8 * (lay["total_bytes"] + 256))()
        b0 = (-C.addressof(raw)) % 256
        at = lambda off: C.c_void_p(C.addressof(raw) + b0 + off)
        (C.c_uint32 * lay["small_words"]).from_address(at(lay["off_small"]).value)[lib.ht_m_one(h)] = 1
        assert lib.ht_run_rsa(h, rec, None, at(lay["off_bits"]), at(lay["off_small"]), at(lay["off_fr"])) == 1
        return raw, at
    raw, at = image()
    full = c.expand_full_host(rec, 1, at(0), 0, 1)
    assert hashlib.sha256(full).hexdigest() == meta["witness_sha256"]          # = the interpreter's complete witness
    w = [int.from_bytes(full[32 * i:32 * i + 32], "little") for i in range(c.W)]
    hdr, cons = ru.read_r1cs(r1cs)
    assert hdr["n_wires"] == c.W and len(cons) == meta["n_constraints"]
    assert all(sum(cf * w[k] for k, cf in a) * sum(cf * w[k] for k, cf in b) % ru.P == sum(cf * w[k] for k, cf in cc) % ru.P for a, b, cc in cons)
    c.attach_r1cs(r1cs)
    assert c.abc_bytes == 96 * len(cons)
    lay1 = c.image_layout(1)
    assert lay1["small_words"] >= lay0["small_words"] and lay1["fr_elems"] > lay0["fr_elems"]
    raw, at = image()                                                          # (the image layout grew)
    out = c.expand_abc_host(rec, 1, at(0), 0, 1, rows_on_host=True)
    assert hashlib.sha256(out).hexdigest() == ru.abc_digest(cons, w)
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abc_digests.json")))
    assert gold["o0_rsa_kat"]["standard"] == hashlib.sha256(out).hexdigest()
    with pytest.raises(zkwg.ZkwgError):
        c.attach_r1cs(r1cs)                                                    # one system per handle
    lib.ht_destroy(h)
    # a file with another wire count is refused
    c2 = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1, sym=sym, sym_alias=meta["alias"], r1cs=r1cs)
    with pytest.raises(zkwg.ZkwgError):
        c2.attach_r1cs(zkwg.WitnessCalculator(zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1)).constraint_system())


def _abc_device_digests(c, rec, n=2):
    """prepare + zkwg_expand_abc_device for n copies of one record -> {form: sha256 of email 0's A.w|B.w|C.w}; all copies equal"""
    import hashlib
    import torch
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream()
    d_in = torch.frombuffer(bytearray(rec * n), dtype=torch.uint8).to(dev)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device=dev)
    c.prepare_device(d_in, n, d_status, d_scr, s)
    out = {}
    for mont in (False, True):
        got = torch.empty((n, c.abc_bytes), dtype=torch.uint8, device=dev)
        c.expand_abc_device(d_in, n, d_scr, 0, n, got, s, montgomery=mont)
        torch.cuda.synchronize()
        assert d_status.cpu().tolist() == [0] * n
        assert torch.equal(got[0], got[n - 1])
        out["montgomery" if mont else "standard"] = hashlib.sha256(got[0].cpu().numpy().tobytes()).hexdigest()
    return out


@pytest.mark.gpu
def test_prover_first_stage_of_the_real_test_eml_against_independent_digests():
    """A.w | B.w | C.w of EmailVerifier(576,192) -- the headline of the image path -- for the REAL test.eml
    (tests/real_email.py) against tests/golden/abc_digests.json["ev_test_eml_576_192_kept"]: the 753,807 combinations evaluated
    in Python integers over the ORACLE's witness (make_abc_digests.py), not against zk_r1cs_eval.  Also: zkwg_calculate_batch
    before and after the attachment (the cached staging buffers of the host path must follow the grown image)."""
    import hashlib
    import real_email as R
    import zkwg
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abc_digests.json")))["ev_test_eml_576_192_kept"]
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=0)
    rec = c.pack(R.ev_inputs("test_eml", 576, 192))
    w0, st0 = c.calculate_batch_host(rec * 3)                 # host path first: its device staging buffers get cached
    assert st0 == [0] * 3 and hashlib.sha256(w0[:c.witness_bytes]).hexdigest() == gold["witness_sha256"]
    before = c.scratch_bytes(3)
    cs = zkwg.WitnessCalculator(c).constraint_system()
    assert cs.n_constraints == gold["constraints"]
    c.attach_r1cs(cs)
    assert c.scratch_bytes(3) > before
    w1, st1 = c.calculate_batch_host(rec * 3)                 # ... and again with the larger image layout
    assert st1 == [0] * 3 and w1 == w0
    got = _abc_device_digests(c, rec)
    assert got["standard"] == gold["standard"] and got["montgomery"] == gold["montgomery"]


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["rsa", "ev"])
def test_prover_first_stage_on_the_compiled_r1cs_from_the_image(which):
    """VERDICT r3 item 2b on the device: the numbered handle built from the interpreter-generated `.sym` + `.r1cs` takes that same
    `.r1cs` as its constraint system; A.w | B.w | C.w of all its constraints (208,463 for rsa-test.circom; 3,131,414 for
    EmailVerifier(576,192) on the REAL test.eml) stream from the image and equal the digests computed in Python integers over
    the INTERPRETER's complete witness (tests/golden/abc_digests.json o0_*), standard and Montgomery form."""
    import gzip
    import real_email as R
    import zkwg
    gold_all = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abc_digests.json")))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if which == "rsa":
        base, gold = os.path.join(root, "tests", "golden", "o0_rsa"), gold_all["o0_rsa_kat"]
        mk = dict(main_kind=zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0)
    else:
        base, gold = os.path.join(root, "artifacts", "o0_ev_576_192"), gold_all["o0_ev_test_eml_576_192"]
        mk = dict(main_kind=zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192)
        if not os.path.exists(base + ".json"):
            pytest.skip("artifacts/o0_ev_576_192.* not built (needs /root/reference)")
    meta = json.load(open(base + ".json"))
    r1cs = gzip.open(base + ".r1cs.gz", "rb").read()
    c = zkwg.Circuit(mk.pop("main_kind"), device=0, sym=gzip.open(base + ".sym.gz", "rb").read(), sym_alias=meta["alias"], r1cs=r1cs, **mk)
    assert c.W == gold["wires"]
    c.attach_r1cs(r1cs)
    assert c.abc_bytes == 96 * gold["constraints"]
    rec = c.pack(meta["inputs"] if which == "rsa" else R.ev_inputs("test_eml", 576, 192))
    got = _abc_device_digests(c, rec)
    assert got["standard"] == gold["standard"] and got["montgomery"] == gold["montgomery"]
    if which == "ev":
        import hashlib
        wit, status = c.calculate_batch_host(rec)             # the complete witness of the same handle is unchanged by the attachment
        assert status == [0] and hashlib.sha256(wit).hexdigest() == gold["witness_sha256"]
