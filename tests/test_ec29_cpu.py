"""The arithmetic the multi-exponentiation kernels run since round 6 -- BN254 Fq in lazy 9 x 29-bit limbs (csrc/zkwg_fq29.h), the G1 / G2
XYZZ formulas over it (csrc/zkwg_ec29.h; G2's lane-pair dot products computed for both halves) and the batched kernel bodies
(csrc/zkwg_msm_core.h) executed thread by thread -- against the oracle (oracle/pyref/bn254_g1.py, bn254_g2.py).  The host build counts
every violated range precondition of the lazy form (limb widths, column sums, dominance of the subtraction constants): must stay 0."""
import ctypes as C
import random

import pytest

import hosttest
from oracle.pyref import bn254_g1 as G
from oracle.pyref import bn254_g2 as H

Q, R = G.Q, G.R


def _lib():
    lib = hosttest.load()
    lib.ht_fq29_violations.restype = C.c_ulonglong
    lib.ht_fq29_op.argtypes = [C.c_int] + [C.c_char_p] * 5
    lib.ht_ec29_g1_op.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p]
    lib.ht_ec29_g2_op.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p]
    lib.ht_msm_device_mirror_batch.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_char_p]
    return lib


def _b(x):
    return int(x).to_bytes(32, "little")


def _pt(p):
    return bytes(64) if p is None else _b(p[0]) + _b(p[1])


def _unpt(b):
    x, y = int.from_bytes(b[:32], "little"), int.from_bytes(b[32:64], "little")
    return None if x == 0 and y == 0 else (x, y)


def _p2(p):
    return bytes(128) if p is None else _b(p[0][0]) + _b(p[0][1]) + _b(p[1][0]) + _b(p[1][1])


def _unp2(b):
    v = [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def test_field_layer():
    lib = _lib()
    rng = random.Random(29)
    vals = [0, 1, 2, Q - 1, Q - 2, (1 << 253) % Q, (1 << 232) - 1, (1 << 29) - 1, ((1 << 261) - 1) % Q] + [rng.randrange(Q) for _ in range(400)]
    out = C.create_string_buffer(32)
    for i, a in enumerate(vals):
        b, c, d = vals[(7 * i + 3) % len(vals)], vals[(11 * i + 5) % len(vals)], vals[(13 * i + 1) % len(vals)]
        for op, want in ((0, a * b % Q), (1, (a * b + c * d) % Q), (2, (a - b) % Q), (3, (a + b + c + d) % Q)):
            lib.ht_fq29_op(op, _b(a), _b(b), _b(c), _b(d), out)
            assert int.from_bytes(out.raw, "little") == want, (op, a, b, c, d)
    assert lib.ht_fq29_violations() == 0


def test_g1_formulas_with_every_special_case_and_accumulated_bounds():
    lib = _lib()
    rng = random.Random(12)
    pts = G.random_points(10, 3)
    out = C.create_string_buffer(64)
    for i, p in enumerate(pts):
        q = pts[(i + 5) % len(pts)]
        for a, b in ((p, q), (p, p), (p, G.neg(p)), (p, None), (None, q), (None, None)):
            s = rng.randrange(1, Q)
            for op in (0, 1):
                lib.ht_ec29_g1_op(op, _pt(a), _pt(b), _b(s), 1, out)
                assert _unpt(out.raw) == G.add(a, b), (op, i)
            for op in (2, 3):
                lib.ht_ec29_g1_op(op, _pt(a), _pt(b), _b(s), 1, out)
                assert _unpt(out.raw) == G.add(a, a), (op, i)
        # chains: the accumulator of one operation is the operand of the next (the stored bounds X [1, 11], Y [3, 7] must hold up)
        s = rng.randrange(1, Q)
        lib.ht_ec29_g1_op(0, _pt(p), _pt(q), _b(s), 9, out)
        assert _unpt(out.raw) == G.add(p, G.mul(9, q))
        lib.ht_ec29_g1_op(1, _pt(p), _pt(q), _b(s), 5, out)
        assert _unpt(out.raw) == G.add(p, G.mul(5, q))
        lib.ht_ec29_g1_op(3, _pt(p), _pt(q), _b(s), 7, out)
        assert _unpt(out.raw) == G.mul(128, p)
    assert lib.ht_fq29_violations() == 0


def test_g2_formulas_on_both_halves_of_the_lane_pair():
    lib = _lib()
    rng = random.Random(31)
    pts = H.random_points(5, 2)
    out = C.create_string_buffer(128)
    for i, p in enumerate(pts):
        q = pts[(i + 1) % len(pts)]
        for a, b in ((p, q), (p, p), (p, H.neg(p)), (p, None), (None, q), (None, None)):
            s = _b(rng.randrange(1, Q)) + _b(rng.randrange(Q))
            for op in (0, 1):
                lib.ht_ec29_g2_op(op, _p2(a), _p2(b), s, 1, out)
                assert _unp2(out.raw) == H.add(a, b), (op, i)
            for op in (2, 3):
                lib.ht_ec29_g2_op(op, _p2(a), _p2(b), s, 1, out)
                assert _unp2(out.raw) == H.add(a, a), (op, i)
        s = _b(rng.randrange(1, Q)) + _b(rng.randrange(Q))
        lib.ht_ec29_g2_op(0, _p2(p), _p2(q), s, 6, out)
        assert _unp2(out.raw) == H.add(p, H.mul(6, q))
        lib.ht_ec29_g2_op(1, _p2(p), _p2(q), s, 4, out)
        assert _unp2(out.raw) == H.add(p, H.mul(4, q))
        lib.ht_ec29_g2_op(3, _p2(p), _p2(q), s, 5, out)
        assert _unp2(out.raw) == H.mul(32, p)
    assert lib.ht_fq29_violations() == 0


def _scalars(ks):
    return (C.c_uint64 * (4 * len(ks)))(*[(k >> (64 * i)) & ((1 << 64) - 1) for k in ks for i in range(4)])


@pytest.mark.parametrize("n,c,E,s0", [(1, 3, 1, 16), (60, 4, 3, 4), (900, 7, 2, 16), (2500, 11, 2, 64)])
def test_batched_kernel_bodies_g1(n, c, E, s0):
    """E emails in one series: per-email index lists (witness-like scalars: zeros, ones, bytes, field elements), shared table, every
    layout (classic / precomputed windows, global / workgroup-local sort), shuffled atomic order"""
    lib = _lib()
    rng = random.Random(5000 + n)
    base = G.random_points(min(n, 24), n + 1)
    pts = [base[rng.randrange(len(base))] if rng.random() < 0.93 else None for _ in range(n)]
    if n > 4:
        pts[1] = G.neg(pts[0]) if pts[0] else None
    ks = []
    for e in range(E):
        for i in range(n):
            u = rng.random()
            ks.append(0 if u < 0.3 else 1 if u < 0.6 else rng.randrange(256) if u < 0.75 else R - 1 - rng.randrange(3) if u < 0.8 else rng.randrange(R))
    if n > 8:
        ks[:7] = [R - 1, 0, 1, 1 << (c - 1), (1 << c) - 1, R - 2, 1 << 253]
    buf = b"".join(_pt(p) for p in pts)
    want = []
    for e in range(E):
        folded = {}
        for p, k in zip(pts, ks[e * n:(e + 1) * n]):
            if p is not None:
                folded[p] = (folded.get(p, 0) + k) % R
        want.append(G.msm_naive(list(folded), list(folded.values())) if folded else None)
    out = C.create_string_buffer(64 * E)
    for mont, shuffle, layout in ((0, 0, 11), (1, 5, 11), (0, 3, 1), (1, 0, 2), (0, 5, 10), (1, 2, 9), (0, 0, 0)):
        lib.ht_msm_device_mirror_batch(1, buf, _scalars(ks), n, E, c, mont, shuffle, layout, s0, out)
        assert [_unpt(out.raw[64 * e:64 * e + 64]) for e in range(E)] == want, (mont, shuffle, layout)
    assert lib.ht_fq29_violations() == 0


@pytest.mark.parametrize("n,c,E", [(40, 4, 2), (700, 7, 2)])
def test_batched_kernel_bodies_g2(n, c, E):
    lib = _lib()
    rng = random.Random(7000 + n)
    base = H.random_points(6, n)
    logs = None
    pts = [base[rng.randrange(len(base))] if rng.random() < 0.9 else None for _ in range(n)]
    pts[1] = H.neg(pts[0]) if pts[0] else None
    ks = [rng.choice([0, 1, 1, rng.randrange(256), R - 1, rng.randrange(R)]) for _ in range(n * E)]
    buf = b"".join(_p2(p) for p in pts)
    want = []
    for e in range(E):
        folded = {}
        for p, k in zip(pts, ks[e * n:(e + 1) * n]):
            if p is not None:
                folded[p] = (folded.get(p, 0) + k) % R
        acc = None
        for p, k in folded.items():
            acc = H.add(acc, H.mul(k, p))
        want.append(acc)
    out = C.create_string_buffer(128 * E)
    for mont, shuffle, layout in ((0, 0, 11), (1, 4, 3), (0, 2, 8), (1, 0, 1)):
        lib.ht_msm_device_mirror_batch(2, buf, _scalars(ks), n, E, c, mont, shuffle, layout, 16, out)
        assert [_unp2(out.raw[128 * e:128 * e + 128]) for e in range(E)] == want, (mont, shuffle, layout)
    assert lib.ht_fq29_violations() == 0
