"""Prover stages 1-3 end to end: inputs -> witness -> A.w | B.w | C.w -> H evaluations -> the five
multi-exponentiations -> pi_a, pi_b, pi_c, all on the device, judged by the PINNED verifier (oracle/pyref/bn254_pairing.py, which
accepts the reference's own proof packages/rust-verifier/tests/data/proof_of_twitter) under a key made from a known trapdoor
(oracle/pyref/groth16.py).  Each multi-exponentiation is also compared with (sum_i k_i s_i) G computed in Python from the
trapdoor's discrete logarithms.  Reference call site: packages/helpers/src/chunked-zkey.ts:80-84."""
import ctypes as C
import random

import pytest

from oracle.pyref import bn254_g1 as G1
from oracle.pyref import bn254_g2 as G2
from oracle.pyref import bn254_pairing as P
from oracle.pyref import groth16 as G

R = G.R
Q = G1.Q


def _mont1(p):
    return bytes(64) if p is None else ((p[0] << 256) % Q).to_bytes(32, "little") + ((p[1] << 256) % Q).to_bytes(32, "little")


def _mont2(p):
    return bytes(128) if p is None else b"".join(((v << 256) % Q).to_bytes(32, "little") for v in (p[0][0], p[0][1], p[1][0], p[1][1]))


def test_assemble_on_the_host_equals_the_oracle():
    """zkwg_groth16_assemble (host arithmetic of zkwg_g1.h / zkwg_g2.h) against Python group operations; no GPU"""
    from zkwg import _lib
    lib = _lib.load()
    rng = random.Random(3)
    k = {n: rng.randrange(1, R) for n in ("a", "b", "c", "h", "alpha", "beta", "delta", "r", "s")}
    k["c"] = 0                                                     # a sum at infinity
    g1 = lambda x: G1.mul(x, G1.G)
    g2 = lambda x: G2.mul(x, G2.G2)
    pa, pb, pc = (C.c_uint8 * 64)(), (C.c_uint8 * 128)(), (C.c_uint8 * 64)()
    rc = lib.zkwg_groth16_assemble(_mont1(g1(k["a"])), _mont1(g1(k["b"])), _mont2(g2(k["b"])), _mont1(g1(k["c"])), _mont1(g1(k["h"])),
                                   _mont1(g1(k["alpha"])), _mont1(g1(k["beta"])), _mont2(g2(k["beta"])), _mont1(g1(k["delta"])), _mont2(g2(k["delta"])),
                                   k["r"].to_bytes(32, "little"), k["s"].to_bytes(32, "little"), pa, pb, pc)
    assert rc == 0
    i = lambda b, j: int.from_bytes(bytes(b)[32 * j:32 * j + 32], "little")
    sa = (k["alpha"] + k["a"] + k["r"] * k["delta"]) % R
    sb = (k["beta"] + k["b"] + k["s"] * k["delta"]) % R
    sc = (k["c"] + k["h"] + k["s"] * sa + k["r"] * sb - k["r"] * k["s"] * k["delta"]) % R
    assert (i(pa, 0), i(pa, 1)) == g1(sa)
    assert ((i(pb, 0), i(pb, 1)), (i(pb, 2), i(pb, 3))) == g2(sb)
    assert (i(pc, 0), i(pc, 1)) == g1(sc)
    # a point off the curve is refused
    bad = bytearray(_mont1(g1(5)))
    bad[0] ^= 1
    assert lib.zkwg_groth16_assemble(bytes(bad), _mont1(g1(1)), _mont2(g2(1)), _mont1(g1(1)), _mont1(g1(1)), _mont1(g1(1)), _mont1(g1(1)), _mont2(g2(1)),
                                     _mont1(g1(1)), _mont2(g2(1)), bytes(32), bytes(32), pa, pb, pc) != 0


@pytest.mark.gpu
def test_gpu_fixed_base_and_g2_sums_equal_the_oracle():
    import torch
    import zkwg
    from zkwg import prover
    rng = random.Random(8)
    ks = [0, 1, 2, R - 1, 1 << 253] + [rng.randrange(R) for _ in range(27)]
    d1 = prover.fixed_base(0, 1, ks)
    d2 = prover.fixed_base(0, 2, ks)
    raw1, raw2 = bytes(d1.cpu().numpy()), bytes(d2.cpu().numpy())
    for j, kk in enumerate(ks):
        assert prover.point_from_montgomery(raw1[64 * j:64 * j + 64]) == G1.mul(kk, G1.G), j
        assert prover.point_from_montgomery(raw2[128 * j:128 * j + 128]) == G2.mul(kk, G2.G2), j
    # a G2 sum over those bases: repeated bases, a base at infinity (index 0), scalars 0 / 1 / r - 1 among random ones; both scalar forms
    n = 700
    idx = [rng.randrange(len(ks)) for _ in range(n)]
    bases = b"".join(raw2[128 * j:128 * j + 128] for j in idx)
    sc = [rng.choice([0, 1, 1, R - 1, rng.randrange(R), rng.randrange(256)]) for _ in range(n)]
    dev = torch.device("cuda", 0)
    d_b = torch.frombuffer(bytearray(bases), dtype=torch.uint8).to(dev)
    m = prover._DeviceMsm(d_b, 2, 0, window_bits=7)
    d_w = torch.empty(m.work_bytes() + 256, dtype=torch.uint8, device=dev)
    d_w = d_w[(-d_w.data_ptr()) % 256:]
    want = G2.mul(sum(s * ks[j] for s, j in zip(sc, idx)) % R, G2.G2)
    for mont in (False, True):
        enc = [(s << 256) % R if mont else s for s in sc]
        d_s = torch.frombuffer(bytearray(b"".join(int(s).to_bytes(32, "little") for s in enc)), dtype=torch.uint8).to(dev)
        for ones_apart in (False, True):
            assert prover.point_from_montgomery(m.run(d_s.data_ptr(), mont, ones_apart, d_w)) == want, (mont, ones_apart)


def _prove_case(N, M, records_of, n_emails=2, check_sums=True, slots=3):
    """EmailVerifier(N, M): toy key from the kept-v1 system, device proofs of the batch's emails, checks"""
    import torch
    import zkwg
    from zkwg import prover
    from zkwg import r1cs as zr
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    sym = c.symbols()
    cons = zr.email_verifier_constraints(sym, N, M)
    n_public = 3 + 17
    full = zr.append_public_rows(cons, n_public)
    data = zr.write_r1cs(len(sym), full, n_pub_out=3, n_pub_in=17, n_prv_in=N + 1 + 17 + 1 + 32 + M + 1)
    key = G.setup(c.W, n_public, cons, seed=11)
    assert key.m + n_public + 1 == len(full)
    pk = prover.ProvingKey.from_scalars(0, n_public, key.power, key.a_tau, key.b_tau, key.c_key[n_public + 1:], key.h_key, key.alpha, key.beta, key.delta)
    pv = prover.Prover(c, data, len(full), pk)
    recs = records_of(c, n_emails)
    dev = torch.device("cuda", 0)
    d_in = torch.frombuffer(bytearray(recs), dtype=torch.uint8).to(dev)
    d_status = torch.zeros(n_emails, dtype=torch.int32, device=dev)
    d_scratch = torch.empty(c.scratch_bytes(n_emails), dtype=torch.uint8, device=dev)
    c.prepare_device(d_in, n_emails, d_status, d_scratch)
    torch.cuda.synchronize()
    assert d_status.tolist() == [0] * n_emails
    vk = G.vkey_json(key)
    rng = random.Random(N)
    singles = []
    for e in range(n_emails):
        r, s = rng.randrange(R), rng.randrange(R)
        proof = pv.prove_prepared(d_in, n_emails, d_scratch, e, r, s)
        w = zkwg.witness_ints(bytes(pv.d_wit.cpu().numpy()))
        if check_sums:
            sc = G.prove_scalars(key, cons, w, r, s)
            # every sum of the device against its discrete logarithm
            for name, k_ in (("a", "a"), ("b1", "b"), ("c", "c"), ("h", "h")):
                assert prover.point_from_montgomery(pv.last_sums[name]) == G1.mul(sc[k_], G1.G), name
            assert prover.point_from_montgomery(pv.last_sums["b2"]) == G2.mul(sc["b"], G2.G2)
            assert proof["pi_a"] == G1.mul(sc["pi_a"], G1.G) and proof["pi_c"] == G1.mul(sc["pi_c"], G1.G) and proof["pi_b"] == G2.mul(sc["pi_b"], G2.G2)
        pub = [str(w[i]) for i in range(1, n_public + 1)]
        pj = prover.Prover.proof_json(proof)
        assert P.groth16_verify(vk, pub, pj)
        if e == 0:
            bad = list(pub)
            bad[2] = str((int(bad[2]) + 1) % R)
            assert not P.groth16_verify(vk, bad, pj)
        singles.append((r, s, proof))
    # the same proofs from the one-call prover: E emails per launch series, rolling contexts (non-consecutive and repeated indices too)
    order = list(range(n_emails)) * 2 + list(range(n_emails - 1, -1, -1))
    batch = pv.prove_batch(d_in, n_emails, d_scratch, order, [singles[e][:2] for e in order], slots=slots)
    assert batch == [singles[e][2] for e in order]
    if not check_sums:
        return
    # inputs -> proofs in one call (zkwg_prover_prove_batch), with a tampered email in the batch: no proof for it, the others unchanged
    bad = bytearray(recs[:c.in_stride])
    bad[c.lib.zkwg_input_offset(c.h, zkwg._lib.IN_SIGNATURE)] ^= 1
    status, proofs = pv.prove_records(recs + bytes(bad), [(r, s) for r, s, _ in singles] + [(1, 2)], slots=slots)
    assert status == [0] * n_emails + [4] and proofs[-1] is None and proofs[:n_emails] == [p for _, _, p in singles]


@pytest.mark.gpu
def test_gpu_proofs_of_synthetic_emails_verify_under_the_pinned_verifier():
    from zkwg import synth
    _prove_case(576, 192, lambda c, n: synth.packed_batch(c, seed=5, n=n, body_len=100)[0])


@pytest.mark.gpu
def test_gpu_proof_of_the_reference_test_eml_verifies_under_the_pinned_verifier():
    """the reference's real email (packages/circuits/tests/test-emails/test.eml, icloud signature) at the size of its own test main,
    EmailVerifier(640, 768, ...) (email-verifier.test.ts:33-44)"""
    import real_email

    def records(c, n):
        return c.pack(real_email.ev_inputs("test_eml", 640, 768)) * n
    _prove_case(640, 768, records, n_emails=1)


@pytest.mark.gpu
def test_gpu_proof_at_the_headline_circuit_verifies_under_the_pinned_verifier():
    """EmailVerifier(1024, 1536, 121, 17, 0, 0, 0, 0) -- BASELINE.json's circuit: W = 1,776,821 wires, 1,814,506 rows, the 2^21 domain -- under
    a key with a known trapdoor: the device proofs of two synthetic 1 KB-body emails (one at a time, and through the batched one-call
    prover) are ACCEPTED by the pinned verifier (VERDICT r5 missing #5: no proof, not even one sum, had been checked at this size)"""
    from zkwg import synth
    _prove_case(1024, 1536, lambda c, n: synth.packed_batch(c, seed=5, n=n, body_len=1024)[0], n_emails=2, check_sums=False, slots=2)


@pytest.mark.gpu
def test_gpu_prover_from_the_zkey_alone_equals_the_r1cs_path():
    """`groth16.prove(zkey, wtns)` takes the zkey and nothing else (fullProve(input, wasm, zkey), chunked-zkey.ts:80-84): the rows of A and B
    come from its section 4, C.w = A.w o B.w.  A key written by zkey.write_zkey WITH that section -> zkwg_prover_create_zkey -> proofs
    byte-identical (same r, s) to the ones of the .r1cs path, accepted by the pinned verifier; a tampered email gets no proof."""
    import torch
    import zkwg
    from zkwg import prover, synth, zkey
    from zkwg import r1cs as zr
    N, M, n_public, n = 576, 192, 20, 3
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    sym = c.symbols()
    cons = zr.email_verifier_constraints(sym, N, M)
    full = zr.append_public_rows(cons, n_public)
    key = G.setup(c.W, n_public, cons, seed=31)
    pk = prover.ProvingKey.from_scalars(0, n_public, key.power, key.a_tau, key.b_tau, key.c_key[n_public + 1:], key.h_key, key.alpha, key.beta, key.delta)
    down = lambda t: bytes(t.cpu().numpy())
    pts = {"alpha1": pk.alpha1, "beta1": pk.beta1, "beta2": pk.beta2, "gamma2": _mont2(G2.mul(key.gamma, G2.G2)), "delta1": pk.delta1, "delta2": pk.delta2}
    ic = b"".join(_mont1(G1.mul(x, G1.G)) for x in key.ic)
    coeffs = [(m, j, w, v % R) for j, row in enumerate(full) for m in (0, 1) for w, v in row[m].items() if v % R]
    zbytes = zkey.write_zkey(c.W, n_public, key.n, pts, ic, down(pk.d_a), down(pk.d_b1), down(pk.d_b2), down(pk.d_c), down(pk.d_h), coeffs)
    del coeffs
    recs, _ = synth.packed_batch(c, seed=8, n=n, body_len=100)
    bad = bytearray(recs[:c.in_stride])
    bad[c.lib.zkwg_input_offset(c.h, zkwg._lib.IN_SIGNATURE)] ^= 1
    rng = random.Random(4)
    bl = [(rng.randrange(R), rng.randrange(R)) for _ in range(n + 1)]
    # the .r1cs path
    data = zr.write_r1cs(len(sym), full, n_pub_out=3, n_pub_in=17, n_prv_in=N + 1 + 17 + 1 + 32 + M + 1)
    st1, want = prover.Prover(c, data, len(full), pk).prove_records(recs + bytes(bad), bl, slots=4)
    del pk
    torch.cuda.empty_cache()
    # the zkey alone, on a fresh handle
    c2 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    pz = prover.Prover.from_zkey(c2, zbytes, slots=4)
    assert c2.abc_bytes == 96 * len(full)
    st2, got = pz.prove_records(recs + bytes(bad), bl, slots=4)
    assert st1 == st2 == [0] * n + [4] and got == want and got[-1] is None
    wit, _ = c2.calculate_batch_host(recs)
    vk = G.vkey_json(key)
    for e in range(n):
        w = zkwg.witness_ints(wit[e * c2.witness_bytes:(e + 1) * c2.witness_bytes])
        assert P.groth16_verify(vk, [str(w[i]) for i in range(1, n_public + 1)], prover.Prover.proof_json(got[e]))
    # a file whose section 4 is missing or whose sizes are off is refused
    z = zkey.read_zkey(zbytes)
    h = C.c_void_p()
    short = zkey.write_zkey(c.W, n_public, key.n, pts, ic, z["a"], z["b1"], z["b2"], z["c"], z["h"], [])
    c3 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    assert c3.lib.zkwg_prover_create_zkey(c3.h, 0, short, len(short), 2, C.byref(h)) != 0
    assert c3.lib.zkwg_prover_create_zkey(c3.h, 0, zbytes[:-100], len(zbytes) - 100, 2, C.byref(h)) != 0


def test_zkey_round_trip():
    """zkwg.zkey: a groth16 .zkey written section by section reads back field by field (layout restated from snarkjs [EXT]; the day
    a real file exists it goes through the same reader)"""
    from zkwg import zkey
    rng = random.Random(2)
    n_vars, n_public, domain = 12, 3, 16
    blob = lambda n: bytes(rng.randrange(256) for _ in range(n))
    pts = {"alpha1": blob(64), "beta1": blob(64), "beta2": blob(128), "gamma2": blob(128), "delta1": blob(64), "delta2": blob(128)}
    sec = {"ic": blob(64 * 4), "a": blob(64 * 12), "b1": blob(64 * 12), "b2": blob(128 * 12), "c": blob(64 * 8), "h": blob(64 * 16)}
    coeffs = [(0, 0, 1, 5), (1, 0, 2, R - 1), (0, 9, 0, 1), (0, 10, 1, 1)]
    data = zkey.write_zkey(n_vars, n_public, domain, pts, sec["ic"], sec["a"], sec["b1"], sec["b2"], sec["c"], sec["h"], coeffs)
    z = zkey.read_zkey(data)
    assert (z["n_vars"], z["n_public"], z["domain_size"], z["power"]) == (12, 3, 16, 4)
    assert all(z[k] == v for k, v in pts.items()) and all(z[k] == v for k, v in sec.items())
    assert z["coeffs"] == coeffs
    with pytest.raises(ValueError):
        zkey.read_zkey(b"zkex" + data[4:])
    with pytest.raises(ValueError):
        zkey.read_zkey(data[:-70])          # the last sections cut off


@pytest.mark.gpu
def test_gpu_node_host_proves_through_the_addon(tmp_path):
    """the Node host (zk-email-verify_amd/js/prove.js -> zkwg.js Prover(circuit, zkey) -> N-API addon -> zkwg_prover_create_zkey /
    zkwg_prover_prove_batch): input.json + a .zkey written from a toy key, NOTHING else -> proof.json / public signals that the pinned
    verifier accepts -- the whole of `groth16.fullProve(input, wasm, zkey)` (packages/helpers/src/chunked-zkey.ts:80-84) behind the
    reference's host language"""
    import json
    import os
    import shutil
    import subprocess
    import torch
    import zkwg
    from conftest import ROOT
    from zkwg import prover, zkey
    from zkwg import r1cs as zr
    js = os.path.join(ROOT, "zk-email-verify_amd", "js")
    if shutil.which("node") is None or not os.path.exists(os.path.join(js, "zkwg_addon.node")):
        pytest.skip("node or the built addon is missing")
    kase = json.load(open(os.path.join(ROOT, "tests", "golden", "ev_576_192_case.json")))
    N, M = kase["maxHeader"], kase["maxBody"]
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    sym = c.symbols()
    cons = zr.email_verifier_constraints(sym, N, M)
    n_public = 20
    full = zr.append_public_rows(cons, n_public)
    key = G.setup(c.W, n_public, cons, seed=21)
    pk = prover.ProvingKey.from_scalars(0, n_public, key.power, key.a_tau, key.b_tau, key.c_key[n_public + 1:], key.h_key, key.alpha, key.beta, key.delta)
    down = lambda t: bytes(t.cpu().numpy())
    pts = {"alpha1": pk.alpha1, "beta1": pk.beta1, "beta2": pk.beta2, "gamma2": _mont2(G2.mul(key.gamma, G2.G2)), "delta1": pk.delta1, "delta2": pk.delta2}
    ic = b"".join(_mont1(G1.mul(x, G1.G)) for x in key.ic)
    coeffs = [(m, j, w, v % R) for j, row in enumerate(full) for m in (0, 1) for w, v in row[m].items() if v % R]
    (tmp_path / "c.zkey").write_bytes(zkey.write_zkey(c.W, n_public, key.n, pts, ic, down(pk.d_a), down(pk.d_b1), down(pk.d_b2), down(pk.d_c), down(pk.d_h), coeffs))
    del coeffs
    del pk, c
    torch.cuda.empty_cache()
    bad = dict(kase["input"], signature=[str(int(kase["input"]["signature"][0]) ^ 1)] + list(kase["input"]["signature"][1:]))
    (tmp_path / "in.json").write_text(json.dumps([kase["input"], bad]))
    env = dict(os.environ)
    r = subprocess.run(["node", os.path.join(js, "prove.js"), f"EmailVerifier({N},{M},121,17,0,0,0,0)", str(tmp_path / "in.json"), str(tmp_path / "c.zkey"),
                        str(tmp_path / "out.json")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 1 and "1 proof(s), 1 failed email(s)" in r.stdout, r.stdout + r.stderr
    out = json.load(open(tmp_path / "out.json"))
    assert out[1]["status"] == 4 and out[1]["proof"] is None
    assert out[0]["publicSignals"][0] == str(int(kase["pubkeyHash"]))
    assert P.groth16_verify(G.vkey_json(key), out[0]["publicSignals"], out[0]["proof"])
