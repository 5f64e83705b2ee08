"""The building blocks of the next row -- the multi-exponentiations of groth16.prove -- against the oracle
(oracle/pyref/bn254_g1.py: pinned by the EIP-196 alt_bn128 vectors): zkwg_fq.h (both Montgomery-product paths, the 8 x 32-bit one
is the device's), zkwg_g1.h (XYZZ additions with every special case, signed-window digits, the bucket method).  Host code only: no
kernel uses these headers yet."""
import ctypes as C
import random

import pytest

import hosttest
from oracle.pyref import bn254_g1 as G

Q, R = G.Q, G.R


def _lib():
    lib = hosttest.load()
    lib.ht_fq_mont_mul.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    lib.ht_fq_op.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.ht_g1_op.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.ht_g1_on_curve.argtypes = [C.c_char_p]
    lib.ht_msm_digits.restype = C.c_uint32
    lib.ht_msm_windows.restype = C.c_uint32
    return lib


def _b(x):
    return int(x).to_bytes(32, "little")


def _pt(p):
    return bytes(64) if p is None else _b(p[0]) + _b(p[1])


def _unpt(b):
    x, y = int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little")
    return None if x == 0 and y == 0 else (x, y)


def test_oracle_is_pinned_by_the_alt_bn128_vectors():
    assert G.on_curve(G.G) and G.add(G.G, G.G) == (G.G2X, G.G2Y) and G.add((G.G2X, G.G2Y), G.G) == (G.G3X, G.G3Y)
    assert G.mul(R, G.G) is None and G.mul(R - 1, G.G) == G.neg(G.G)
    pts = G.random_points(24, 5)
    rng = random.Random(6)
    ks = [rng.randrange(R) for _ in pts]
    assert G.msm_naive(pts, ks) == G.msm_buckets(pts, ks, 5) == G.msm_buckets(pts, ks, 11)


def test_fq_arithmetic_both_montgomery_paths():
    lib = _lib()
    rng = random.Random(11)
    rinv = pow(1 << 256, -1, Q)
    vals = [0, 1, 2, Q - 1, Q - 2, (1 << 253) % Q, (1 << 256) % Q] + [rng.randrange(Q) for _ in range(300)]
    out = C.create_string_buffer(32)
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)]
        for path in (0, 1):
            lib.ht_fq_mont_mul(_b(a), _b(b), out, path)
            assert int.from_bytes(out.raw, "little") == a * b * rinv % Q, (a, b, path)
        for op, want in ((0, (a + b) % Q), (1, (a - b) % Q), (2, a * b % Q), (4, (-a) % Q)):
            lib.ht_fq_op(op, _b(a), _b(b), out)
            assert int.from_bytes(out.raw, "little") == want, (op, a, b)
        if a and i < 40:
            lib.ht_fq_op(3, _b(a), _b(b), out)
            assert int.from_bytes(out.raw, "little") == pow(a, -1, Q)


def test_the_devices_9x29_product_equals_the_64_bit_path_on_edge_and_random_operands():
    """zkwg_comba29.h is every device fr_mont_mul / fq_mont_mul since round 5; here it runs on the host: canonical operands
    incl. 0, 1, p - 1, the limb boundaries of both splittings (29-bit and the << 5 one), R, R^2, and random ones, both fields"""
    lib = _lib()
    lib.ht_comba_mont_mul.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
    out = C.create_string_buffer(32)
    for field, p in ((0, Q), (1, R)):
        rng = random.Random(29 + field)
        rinv = pow(1 << 256, -1, p)
        edge = [0, 1, 2, p - 1, p - 2, (1 << 256) % p, pow(1 << 256, 2, p), (1 << 253) % p, (1 << 232) - 1, 1 << 232, (1 << 224) - 1, 1 << 224]
        edge += [(1 << (29 * i)) - 1 for i in range(1, 9)] + [1 << (29 * i) for i in range(1, 9)] + [((1 << (29 * i)) - 1) >> 5 for i in range(2, 9)]
        edge += [p - (1 << (29 * i)) for i in range(1, 8)]
        vals = [v % p for v in edge] + [rng.randrange(p) for _ in range(400)]
        for i, a in enumerate(vals):
            for b in (vals[(7 * i + 3) % len(vals)], vals[i], p - 1):
                lib.ht_comba_mont_mul(field, _b(a), _b(b), out)
                assert int.from_bytes(out.raw, "little") == a * b * rinv % p, (field, a, b)


def test_g1_additions_with_every_special_case():
    lib = _lib()
    rng = random.Random(12)
    pts = G.random_points(12, 3)
    out = C.create_string_buffer(64)
    cases = []
    for i, p in enumerate(pts):
        q = pts[(i + 5) % len(pts)]
        cases += [(p, q), (p, p), (p, G.neg(p)), (p, None), (None, q), (None, None)]
    for p, q in cases:
        s = rng.randrange(1, Q)
        for op in (0, 1, 4):
            lib.ht_g1_op(op, _pt(p), _pt(q), _b(s), out)
            assert _unpt(out.raw) == G.add(p, q), (op, p, q)
        for op in (2, 3):
            lib.ht_g1_op(op, _pt(p), _pt(q), _b(s), out)
            assert _unpt(out.raw) == G.add(p, p), (op, p)
        assert lib.ht_g1_on_curve(_pt(p)) == 1
    assert lib.ht_g1_on_curve(_pt((1, 3))) == 0


@pytest.mark.parametrize("c", [2, 5, 8, 13, 16])
def test_signed_window_digits_recompose_the_scalar(c):
    lib = _lib()
    K = lib.ht_msm_windows(c)
    rng = random.Random(c)
    half = 1 << (c - 1)
    for k in [0, 1, R - 1, half, half + 1, (1 << 254) - 1, (1 << c) - 1] + [rng.randrange(R) for _ in range(200)]:
        limbs = (C.c_uint64 * 4)(*[(k >> (64 * i)) & ((1 << 64) - 1) for i in range(4)])
        d = (C.c_int32 * K)()
        assert lib.ht_msm_digits(limbs, c, d) == 0
        assert all(-half <= x <= half for x in d)
        assert sum(int(x) << (c * w) for w, x in enumerate(d)) == k


@pytest.mark.parametrize("n,c", [(1, 4), (37, 3), (300, 7), (1024, 10)])
def test_bucket_method_equals_the_oracle(n, c):
    lib = _lib()
    rng = random.Random(100 + n)
    base = G.random_points(min(n, 64), n)
    pts = [base[rng.randrange(len(base))] if rng.random() < 0.9 else None for _ in range(n)]      # repeated bases, some at infinity
    if n > 4:
        pts[1] = G.neg(pts[0]) if pts[0] else None                                             # a base and its negative
    special = [0, 1, R - 1, R - 2, 1 << (c - 1), (1 << c) - 1, (1 << 253)]
    ks = [special[i] if i < len(special) and i < n else rng.randrange(R) for i in range(n)]
    if n > 8:
        ks[7] = ks[6]                                                                          # equal scalars on maybe-equal bases
    buf = b"".join(_pt(p) for p in pts)
    sc = (C.c_uint64 * (4 * n))(*[(k >> (64 * i)) & ((1 << 64) - 1) for k in ks for i in range(4)])
    out = C.create_string_buffer(64)
    lib.ht_msm(buf, sc, C.c_uint64(n), c, out)
    want = G.msm_buckets(pts, ks, 6) if n > 64 else G.msm_naive(pts, ks)
    assert _unpt(out.raw) == want


def test_device_path_compiles_for_gfx950_without_scratch(tmp_path):
    """hipcc cross-compiles the probe kernels of tests/native/g1_device_probe.hip (a bucket's accumulate loop of mixed additions, a
    full addition): the 8 x 32-bit Montgomery path builds for gfx950, and the XYZZ accumulator stays in registers"""
    import os
    import re
    import shutil
    import subprocess
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "zk-email-verify_amd", "csrc"), "-c",
                        os.path.join(ROOT, "tests", "native", "g1_device_probe.hip"), "-o", str(tmp_path / "p.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    vgprs = [int(x) for x in re.findall(r" VGPRs: (\d+)", r.stderr)]
    assert len(scratch) == 2 and scratch == [0, 0], r.stderr[-1500:]
    assert max(vgprs) <= 128, vgprs          # 4 wavefronts per SIMD


@pytest.mark.parametrize("n,c,mont", [(1, 3, 0), (50, 4, 1), (700, 7, 0), (2000, 11, 1), (3000, 13, 0)])
def test_device_msm_bodies_thread_by_thread_equal_the_oracle(n, c, mont):
    """zkwg_msm_core.h -- count / scan / scatter (one atomic per digit, or workgroup-local histograms) / sliced bucket sums / the
    weighted bucket sum (the 8-way (S, A) tree, or bit planes) / window combination, the bodies of the kernels of zkwg_kernels_msm.hip
    -- executed thread by thread on the host in launch order, with the atomic passes in a shuffled thread and workgroup order: equals
    the oracle's bucket method (c = 11, 13: several levels of every tree)."""
    lib = _lib()
    lib.ht_msm_device_mirror.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_char_p]
    rng = random.Random(900 + n)
    base = G.random_points(min(n, 48), n + 1)
    pts = [base[rng.randrange(len(base))] if rng.random() < 0.93 else None for _ in range(n)]
    if n > 4:
        pts[1] = G.neg(pts[0]) if pts[0] else None
    special = [R - 1, 0, 1, 1 << (c - 1), (1 << c) - 1, R - 2, 1 << 253]
    ks = [special[i] if i < len(special) and i < n else rng.randrange(R) for i in range(n)]
    buf = b"".join(_pt(p) for p in pts)
    sc = (C.c_uint64 * (4 * n))(*[(k >> (64 * i)) & ((1 << 64) - 1) for k in ks for i in range(4)])
    want = G.msm_buckets(pts, ks, 7) if n > 64 else G.msm_naive(pts, ks)
    out = C.create_string_buffer(64)
    for shuffle in (0, 5):
        # bit 1: precomputed windows -- K shifted copies of the bases, one bucket set, no Horner pass; bit 2: the weighted bucket sum by
        # bit planes (zk_msm_plane*) instead of the (S, A) tree (bits 1 and 2)
        # bit 3: count / scatter with workgroup-local histograms (14 = the default configuration)
        for layout in ((0, 2, 4, 6, 14, 8) if shuffle else (0, 14)):
            lib.ht_msm_device_mirror(buf, sc, C.c_uint64(n), c, mont, shuffle, layout, out)
            assert _unpt(out.raw) == want, (n, c, mont, shuffle, layout)


@pytest.mark.parametrize("n,c", [(40, 4), (5000, 8), (9000, 10)])
def test_device_msm_bodies_on_witness_like_scalars(n, c):
    """scalars as a witness has them -- mostly 0 and 1, some bytes, a few field elements -- with ones_apart: the bases with scalar
    1 go through zk_msm_ones (8 per lane) + the 8-way joins (several levels here) instead of one bucket"""
    lib = _lib()
    lib.ht_msm_device_mirror.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_char_p]
    rng = random.Random(1700 + n)
    base = G.random_points(32, n + 3)
    pts = [base[rng.randrange(len(base))] if rng.random() < 0.97 else None for _ in range(n)]
    ks = [rng.choice([0, 1]) if rng.random() < 0.9 else (rng.randrange(256) if rng.random() < 0.7 else rng.randrange(R)) for _ in range(n)]
    buf = b"".join(_pt(p) for p in pts)
    sc = (C.c_uint64 * (4 * n))(*[(k >> (64 * i)) & ((1 << 64) - 1) for k in ks for i in range(4)])
    # the oracle, folded over the 32 distinct bases
    folded = {}
    for p, k in zip(pts, ks):
        if p is not None:
            folded[p] = (folded.get(p, 0) + k) % R
    want = G.msm_naive(list(folded), list(folded.values()))
    out = C.create_string_buffer(64)
    for apart in (1, 0, 3, 7, 15, 9):       # bit 1: the precomputed-windows layout, bit 2: bucket sums by bit planes, bit 3: workgroup-local sort
        lib.ht_msm_device_mirror(buf, sc, C.c_uint64(n), c, 1, 3, apart, out)
        assert _unpt(out.raw) == want, (n, c, apart)

def test_g2_arithmetic_against_the_oracle():
    """zkwg_g2.h (Fq2 by Karatsuba, the XYZZ additions over the twist) against oracle/pyref/bn254_g2.py, which is pinned by the
    EIP-197 generator: on the twist, r * G2 = O"""
    from oracle.pyref import bn254_g2 as H
    assert H.on_curve(H.G2) and H.mul(R, H.G2) is None and H.add(H.mul(R - 1, H.G2), H.G2) is None
    lib = _lib()
    for f in (lib.ht_fq2_op, lib.ht_g2_op):
        f.restype = None
    rng = random.Random(31)

    def f2b(a):
        return _b(a[0]) + _b(a[1])

    def p2b(p):
        return bytes(128) if p is None else f2b(p[0]) + f2b(p[1])

    def unf2(b):
        return (int.from_bytes(b[:32], "little"), int.from_bytes(b[32:64], "little"))

    out = C.create_string_buffer(64)
    for _ in range(100):
        a, b = (rng.randrange(Q), rng.randrange(Q)), (rng.randrange(Q), rng.randrange(Q))
        for op, want in ((0, H.f2_add(a, b)), (1, H.f2_sub(a, b)), (2, H.f2_mul(a, b)), (3, H.f2_mul(a, a)), (4, H.f2_inv(a))):
            lib.ht_fq2_op(op, f2b(a), f2b(b), out)
            assert unf2(out.raw) == want, op
    pts = H.random_points(6, 2)
    out = C.create_string_buffer(128)
    for i, p in enumerate(pts):
        q = pts[(i + 1) % len(pts)]
        for a, b in ((p, q), (p, p), (p, H.neg(p)), (p, None), (None, q), (None, None)):
            s = (rng.randrange(1, Q), rng.randrange(Q))
            for op in (0, 1):
                lib.ht_g2_op(op, p2b(a), p2b(b), f2b(s), out)
                got = None if out.raw == bytes(128) else (unf2(out.raw[:64]), unf2(out.raw[64:]))
                assert got == H.add(a, b), (op, i)
            for op in (2, 3):
                lib.ht_g2_op(op, p2b(a), p2b(b), f2b(s), out)
                got = None if out.raw == bytes(128) else (unf2(out.raw[:64]), unf2(out.raw[64:]))
                assert got == H.add(a, a), (op, i)
        assert lib.ht_g2_on_curve(p2b(p)) == 1
    assert lib.ht_g2_on_curve(p2b(((1, 0), (2, 0)))) == 0
