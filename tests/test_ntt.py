"""Prover stage 2 (SURVEY.md 8f4 "next"): batched BN254-Fr transforms and the H evaluations
a(x) b(x) - c(x) on the odd coset that `snarkjs.groth16.prove` computes after A.w | B.w | C.w (reference call site
packages/helpers/src/chunked-zkey.ts:80-84; algorithm of snarkjs / ffjavascript restated in oracle/pyref/ntt.py [EXT], parity
unpinned against the real packages: no vector exists offline).  Oracle = Python integers: O(n^2) transforms on small domains,
the polynomial identity (barycentric evaluation of the interpolant at coset points) on large ones."""
import random

import pytest

R = 1 << 256


def test_oracle_is_self_consistent():
    from oracle.pyref import ntt
    rng = random.Random(1)
    assert pow(ntt.root(20), 1 << 20, ntt.P) == 1 and pow(ntt.root(20), 1 << 19, ntt.P) == ntt.P - 1
    assert ntt.coset_inc(20) == ntt.root(21) and ntt.coset_inc(28) == 25
    for power in (3, 5):
        n = 1 << power
        a, b, c = ([rng.randrange(ntt.P) for _ in range(n - 2)] for _ in range(3))
        assert ntt.ifft(ntt.fft(a + [0, 0])) == a + [0, 0]
        h = ntt.h_evaluations(a, b, c, power)
        for k in (0, 1, n - 1):
            assert h[k] == (ntt.coset_eval_direct(a, power, k) * ntt.coset_eval_direct(b, power, k) - ntt.coset_eval_direct(c, power, k)) % ntt.P


def test_library_exports_the_transform_entry_points_and_refuses_without_a_device():
    import ctypes as C
    from zkwg import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.zkwg_ntt_create(-1, 12, C.byref(h)) == 0          # tables only (layout-only plan)
    assert lib.zkwg_ntt_domain(h) == 4096 and lib.zkwg_ntt_work_bytes(h, 2) == 2 * 3 * 4096 * 36     # limb form between the passes: 16 + 16 + 4 bytes per element
    assert lib.zkwg_ntt_transform_device(h, 256, 1, 0, None) == -3   # ZKWG_RC_NO_DEVICE: no CPU fallback
    lib.zkwg_ntt_destroy(h)
    assert lib.zkwg_ntt_create(-1, 1, C.byref(h)) != 0 and lib.zkwg_ntt_create(-1, 40, C.byref(h)) != 0


def _to_dev(torch, vals, mont=True):
    from oracle.pyref import ntt
    b = b"".join(((v * R % ntt.P) if mont else v).to_bytes(32, "little") for v in vals)
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).to("cuda:0")


def _from_dev(t, mont=True):
    from oracle.pyref import ntt
    raw = t.cpu().numpy().tobytes()
    rinv = pow(R, ntt.P - 2, ntt.P)
    return [int.from_bytes(raw[i:i + 32], "little") * (rinv if mont else 1) % ntt.P for i in range(0, len(raw), 32)]


@pytest.mark.gpu
@pytest.mark.parametrize("power", [2, 3, 6, 8, 9, 10, 11, 13, 19])
def test_transforms_match_the_oracle(power):
    """Fr.fft / Fr.ifft (natural order in and out) for several polynomials at once: one row pass (<= 2^10), one and two column
    passes above; small domains against the O(n^2) transform, large ones at random output indices."""
    import torch
    import zkwg
    from oracle.pyref import ntt
    rng = random.Random(power)
    n, polys = 1 << power, (4 if power <= 13 else 2)
    xs = [[rng.randrange(ntt.P) if rng.random() < 0.9 else rng.randrange(3) for _ in range(n)] for _ in range(polys)]
    plan = zkwg.Ntt(power)
    d = _to_dev(torch, [v for x in xs for v in x])
    plan.transform_device(d, polys)
    torch.cuda.synchronize()
    got = _from_dev(d)
    w = ntt.root(power)
    for q, x in enumerate(xs):
        if power <= 6:
            assert got[q * n:(q + 1) * n] == ntt.fft(x)
        else:
            for k in [0, 1, n // 2, n - 1] + [rng.randrange(n) for _ in range(3)]:
                wk = pow(w, k, ntt.P)
                acc = 0
                for v in reversed(x):          # Horner at w^k
                    acc = (acc * wk + v) % ntt.P
                assert got[q * n + k] == acc, (q, k)
    plan.transform_device(d, polys, inverse=True)
    torch.cuda.synchronize()
    assert _from_dev(d) == [v for x in xs for v in x]


@pytest.mark.gpu
@pytest.mark.parametrize("power,m", [(4, 13), (8, 256), (12, 3001), (14, 16384)])
def test_h_evaluations_match_the_oracle(power, m):
    """groth16_prove.js: ifft -> times inc^i -> fft of A.w, B.w, C.w (m constraints, zero-padded to the domain), then a b - c;
    3 emails at once with a padded record stride."""
    import torch
    import zkwg
    from oracle.pyref import ntt
    rng = random.Random(100 + power)
    n, emails = 1 << power, 3
    abc = [[[rng.randrange(ntt.P) for _ in range(m)] for _ in range(3)] for _ in range(emails)]
    stride = 96 * m + 64
    buf = bytearray(emails * stride)
    for e in range(emails):
        for j in range(3):
            for i, v in enumerate(abc[e][j]):
                o = e * stride + 32 * (j * m + i)
                buf[o:o + 32] = (v * R % ntt.P).to_bytes(32, "little")
    d_abc = torch.frombuffer(buf, dtype=torch.uint8).to("cuda:0")
    plan = zkwg.Ntt(power)
    d_work = torch.empty(plan.work_bytes(emails), dtype=torch.uint8, device="cuda:0")
    d_out = torch.zeros(emails * 32 * n, dtype=torch.uint8, device="cuda:0")
    plan.h_evaluations_device(d_abc, stride, m, emails, d_work, d_out)
    torch.cuda.synchronize()
    got = _from_dev(d_out)
    for e in range(emails):
        a, b, c = abc[e]
        if power <= 8:
            assert got[e * n:(e + 1) * n] == ntt.h_evaluations(a, b, c, power)
        else:
            for k in [0, n - 1, rng.randrange(n)]:
                want = (ntt.coset_eval_direct(a, power, k) * ntt.coset_eval_direct(b, power, k) - ntt.coset_eval_direct(c, power, k)) % ntt.P
                assert got[e * n + k] == want, (e, k)


@pytest.mark.gpu
@pytest.mark.parametrize("power", [7, 14, 21])
def test_transforms_hold_their_value_bounds_on_extreme_inputs(power):
    """the butterflies keep values in limb form across the stages of a pass (round 6): every element r - 1 makes every sum path as large
    as it can be (a DIF stage pair grows a value 4 x): fft of the constant r - 1 is (-n, 0, 0, ...), and the round trip returns the input;
    alternating 0 / r - 1 does the same for the difference paths"""
    import torch
    import zkwg
    from oracle.pyref import ntt
    n = 1 << power
    plan = zkwg.Ntt(power)
    top = (ntt.P - 1) * R % ntt.P
    raw = top.to_bytes(32, "little") * n + (bytes(32) + top.to_bytes(32, "little")) * (n // 2)
    d = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to("cuda:0")
    plan.transform_device(d, 2)
    torch.cuda.synchronize()
    got = d.cpu().numpy().tobytes()
    val = lambda q, k: int.from_bytes(got[32 * (q * n + k):32 * (q * n + k) + 32], "little") * pow(R, ntt.P - 2, ntt.P) % ntt.P
    assert val(0, 0) == (-n) % ntt.P and all(val(0, k) == 0 for k in (1, 2, n // 2, n - 1))
    assert val(1, 0) == (-(n // 2)) % ntt.P and val(1, n // 2) == (n // 2) % ntt.P and val(1, 1) == 0 and val(1, n - 1) == 0
    plan.transform_device(d, 2, inverse=True)
    torch.cuda.synchronize()
    assert d.cpu().numpy().tobytes() == raw


@pytest.mark.gpu
def test_h_evaluations_on_the_headline_domain():
    """2^21 points, 1,814,506 rows (EmailVerifier(1024,1536)'s system with its public rows): three passes of seven stages per transform;
    two emails, spot checks against the oracle's direct evaluation on the coset"""
    import numpy as np
    import torch
    import zkwg
    from oracle.pyref import ntt
    power, m, emails = 21, 1814506, 2
    n = 1 << power
    rng = np.random.default_rng(21)
    # small values (a witness is mostly bits) with a sprinkling of full-size ones: the direct evaluation stays affordable in Python
    vals = rng.integers(0, 3, size=(emails, 3, m), dtype=np.int64)
    big = {}
    prng = random.Random(5)
    for e in range(emails):
        for j in range(3):
            for i in prng.sample(range(m), 400):
                big[(e, j, i)] = prng.randrange(ntt.P)
    words = np.zeros((emails, 3 * m, 4), dtype="<u8")
    r1, r2 = R % ntt.P, 2 * R % ntt.P
    lut = np.array([[0, 0, 0, 0]] + [[(x >> (64 * k)) & (2 ** 64 - 1) for k in range(4)] for x in (r1, r2)], dtype="<u8")
    words[:] = lut[vals.reshape(emails, 3 * m)]
    for (e, j, i), v in big.items():
        x = v * R % ntt.P
        words[e, j * m + i] = [(x >> (64 * k)) & (2 ** 64 - 1) for k in range(4)]
    d_abc = torch.from_numpy(words.view(np.uint8).reshape(-1).copy()).to("cuda:0")
    plan = zkwg.Ntt(power)
    d_work = torch.empty(plan.work_bytes(emails), dtype=torch.uint8, device="cuda:0")
    d_out = torch.zeros(emails * 32 * n, dtype=torch.uint8, device="cuda:0")
    plan.h_evaluations_device(d_abc, 96 * m, m, emails, d_work, d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().tobytes()
    rinv = pow(R, ntt.P - 2, ntt.P)
    for e in range(emails):
        polys = []
        for j in range(3):
            p = [int(x) for x in vals[e, j]]
            for (ee, jj, i), v in big.items():
                if ee == e and jj == j:
                    p[i] = v
            polys.append(p)
        for k in (0, n - 1 - 7 * e):
            want = (ntt.coset_eval_direct(polys[0], power, k) * ntt.coset_eval_direct(polys[1], power, k) - ntt.coset_eval_direct(polys[2], power, k)) % ntt.P
            o = 32 * (e * n + k)
            assert int.from_bytes(got[o:o + 32], "little") * rinv % ntt.P == want, (e, k)


@pytest.mark.gpu
def test_h_evaluations_of_the_real_test_eml_from_the_image():
    """End of the device-side chain this repo covers: inputs of the REAL test.eml -> prepare -> A.w | B.w | C.w of
    EmailVerifier(576,192)'s 753,807 constraints straight from the image (Montgomery form) -> H evaluations on the 2^20 domain.
    Because the witness satisfies the system, a b - c vanishes on the domain itself, so h(x) = (a b - c)(x) / (x^n - 1) is a
    polynomial: checked through the oracle's barycentric evaluation of a, b, c at two coset points."""
    import torch
    import real_email as RE
    import zkwg
    from oracle.pyref import ntt
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=0)
    cs = zkwg.WitnessCalculator(c).constraint_system()
    c.attach_r1cs(cs)
    m = cs.n_constraints
    rec = c.pack(RE.ev_inputs("test_eml", 576, 192))
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream()
    d_in = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    d_scr = torch.empty(c.scratch_bytes(1), dtype=torch.uint8, device=dev)
    c.prepare_device(d_in, 1, d_status, d_scr, s)
    d_abc = torch.empty(c.abc_bytes, dtype=torch.uint8, device=dev)
    c.expand_abc_device(d_in, 1, d_scr, 0, 1, d_abc, s, montgomery=True)
    plan = zkwg.Ntt(20)
    d_work = torch.empty(plan.work_bytes(1), dtype=torch.uint8, device=dev)
    d_out = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    plan.h_evaluations_device(d_abc, c.abc_bytes, m, 1, d_work, d_out)
    torch.cuda.synchronize()
    assert d_status.cpu().tolist() == [0]
    vals = _from_dev(d_abc)
    a, b, cc = vals[:m], vals[m:2 * m], vals[2 * m:]
    assert all((a[i] * b[i] - cc[i]) % ntt.P == 0 for i in range(0, m, 997))       # the witness satisfies its constraints
    got = _from_dev(d_out)
    for k in (1, (1 << 20) - 3):
        want = (ntt.coset_eval_direct(a, 20, k) * ntt.coset_eval_direct(b, 20, k) - ntt.coset_eval_direct(cc, 20, k)) % ntt.P
        assert got[k] == want and want != 0
