"""G1 multi-exponentiation on the device (include/zkwg.h "prover stage 3") against the oracle (oracle/pyref/bn254_g1.py).  The kernel
bodies are the ones tests/test_g1_cpu.py / test_ec29_cpu.py run thread by thread on the host."""
import random

import pytest

from oracle.pyref import bn254_g1 as G

R = G.R


def _case(n, seed, c=0, mont=False):
    import torch
    import zkwg
    rng = random.Random(seed)
    base = G.random_points(min(n, 64), seed)
    pts = [base[rng.randrange(len(base))] if rng.random() < 0.95 else None for _ in range(n)]
    if n > 4:
        pts[1] = G.neg(pts[0]) if pts[0] else None
    special = [R - 1, 0, 1, 2, R - 2, 1 << 253]
    ks = [special[i] if i < len(special) and i < n else rng.randrange(R) for i in range(n)]
    m = zkwg.Msm(zkwg.Msm.pack_bases(pts), device=0, window_bits=c)
    enc = [(k << 256) % R if mont else k for k in ks]
    buf = b"".join(int(k).to_bytes(32, "little") for k in enc)
    dev = torch.device("cuda", 0)
    d_s = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev)
    d_w = torch.empty(m.work_bytes() + 256, dtype=torch.uint8, device=dev)
    off = (-d_w.data_ptr()) % 256
    got = m.g1_device(d_s, mont, d_w[off:], ones_apart=(n % 2 == 0))
    want = G.msm_buckets(pts, ks, 8)
    assert got == want, (n, c, mont)


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,mont", [(1, 4, False), (300, 7, True), (5000, 0, False), (70000, 13, True)])
def test_gpu_msm_equals_the_oracle(n, c, mont):
    _case(n, 4000 + n, c, mont)


@pytest.mark.gpu
def test_gpu_msm_of_h_sized_input_is_linear():
    """2^18 bases, random scalars: MSM(k) + MSM(k') = MSM(k + k') and MSM(2 k) = 2 MSM(k) (the oracle cannot finish this size)"""
    import torch
    import zkwg
    n = 1 << 18
    rng = random.Random(77)
    base = G.random_points(256, 9)
    pts = [base[i % 256] for i in range(n)]
    m = zkwg.Msm(zkwg.Msm.pack_bases(pts), device=0)
    dev = torch.device("cuda", 0)
    d_w = torch.empty(m.work_bytes() + 256, dtype=torch.uint8, device=dev)
    off = (-d_w.data_ptr()) % 256

    def run(ks):
        buf = b"".join(int(k).to_bytes(32, "little") for k in ks)
        return m.g1_device(torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev), False, d_w[off:])
    k1 = [rng.randrange(R) for _ in range(n)]
    k2 = [rng.randrange(R) for _ in range(n)]
    a, b = run(k1), run(k2)
    assert run([(x + y) % R for x, y in zip(k1, k2)]) == G.add(a, b)
    assert run([2 * x % R for x in k1]) == G.add(a, a)
    # against the oracle through the structure of the bases: sum_i k_i base[i % 256] = sum_j (sum_{i = j mod 256} k_i) base[j]
    folded = [sum(k1[j::256]) % R for j in range(256)]
    assert a == G.msm_naive(base, folded)


@pytest.mark.gpu
@pytest.mark.parametrize("n,E,c,mont,apart", [(3000, 5, 10, False, True), (20000, 3, 13, True, False), (700, 9, 7, False, True)])
def test_gpu_batched_msm_equals_one_at_a_time_and_the_oracle(n, E, c, mont, apart):
    """E scalar vectors in ONE launch series (emails = the second grid dimension of every kernel): every sum equals the oracle's and the
    sum the same plan gives alone; witness-like scalars (zeros, ones, bytes, negatives, field elements) so that the index lists differ
    per email"""
    import torch
    import zkwg
    rng = random.Random(6000 + n)
    base = G.random_points(32, n)
    pts = [base[rng.randrange(32)] if rng.random() < 0.95 else None for _ in range(n)]
    pts[1] = G.neg(pts[0]) if pts[0] else None
    ks = []
    for e in range(E):
        for i in range(n):
            u = rng.random()
            ks.append(0 if u < 0.4 else 1 if u < 0.7 else rng.randrange(256) if u < 0.8 else R - 1 - rng.randrange(2) if u < 0.85 else rng.randrange(R))
    ks[n:2 * n] = [0] * n                                      # an email whose lists are empty
    m = zkwg.Msm(zkwg.Msm.pack_bases(pts), device=0, window_bits=c)
    enc = [(k << 256) % R if mont else k for k in ks]
    dev = torch.device("cuda", 0)
    d_s = torch.frombuffer(bytearray(b"".join(int(k).to_bytes(32, "little") for k in enc)), dtype=torch.uint8).to(dev)
    d_w = torch.empty(m.work_bytes(E) + 256, dtype=torch.uint8, device=dev)
    d_w = d_w[(-d_w.data_ptr()) % 256:]
    got = m.g1_batch_device(d_s, E, mont, d_w, ones_apart=apart)
    for e in range(E):
        folded = {}
        for p, k in zip(pts, ks[e * n:(e + 1) * n]):
            if p is not None:
                folded[p] = (folded.get(p, 0) + k) % R
        assert got[e] == G.msm_naive(list(folded), list(folded.values())), e
        assert got[e] == m.g1_device(d_s[32 * n * e:32 * n * (e + 1)], mont, d_w, ones_apart=apart), e
    assert got[1] is None


@pytest.mark.gpu
@pytest.mark.parametrize("group,n,c", [(1, 900, 7), (2, 900, 7), (1, 6000, 13)])
def test_gpu_classic_layout_when_the_tables_do_not_fit(group, n, c):
    """zkwg_msm_create_ex with a table budget too small for the K shifted copies: the plan keeps the classic layout (K bucket sets, the
    windows combined by a Horner pass in limb form) -- what a prover falls back to beside full HBM -- and gives the same sums as the
    precomputed-windows plan and the oracle, E emails per series, both groups.  At window 13 the classic layout has 20 x 4,096 buckets: more
    than a workgroup's histogram holds, so the counting sort runs its global-counter kernels (zk_msm_count / zk_msm_scan / zk_msm_scatter)"""
    import ctypes as C
    import torch
    from oracle.pyref import bn254_g2 as H
    from zkwg import _lib, prover
    lib = _lib.load()
    rng = random.Random(90 + group)
    E = 3
    ks = [rng.randrange(1, R) for _ in range(40)]
    d_pts = prover.fixed_base(0, group, ks)
    size = 64 if group == 1 else 128
    raw = bytes(d_pts.cpu().numpy())
    idx = [rng.randrange(40) for _ in range(n)]
    d_b = torch.frombuffer(bytearray(b"".join(raw[size * j:size * j + size] for j in idx)), dtype=torch.uint8).to("cuda:0")
    sc = [rng.choice([0, 1, 1, R - 1, rng.randrange(R), rng.randrange(256)]) for _ in range(n * E)]
    d_s = torch.frombuffer(bytearray(b"".join(int(s).to_bytes(32, "little") for s in sc)), dtype=torch.uint8).to("cuda:0")
    got = {}
    for budget in (1, 0):
        h = C.c_void_p()
        assert lib.zkwg_msm_create_ex(0, group, d_b.data_ptr(), 1, n, c, 16, budget, C.byref(h)) == 0
        assert lib.zkwg_msm_precomputed(h) == (0 if budget else 1)
        d_w = torch.empty(lib.zkwg_msm_work_bytes_batch(h, E) + 256, dtype=torch.uint8, device="cuda:0")
        d_w = d_w[(-d_w.data_ptr()) % 256:]
        d_o = torch.empty((128 if group == 1 else 256) * E, dtype=torch.uint8, device="cuda:0")
        for apart in (1, 0):
            assert lib.zkwg_msm_enqueue_batch_device(h, d_s.data_ptr(), 32 * n, E, 0, apart, d_w.data_ptr(), d_o.data_ptr(), None) == 0
            torch.cuda.synchronize()
            pts = (C.c_uint8 * (size * E))()
            assert lib.zkwg_msm_finish_host(group, bytes(d_o.cpu().numpy()), E, pts) == 0
            got[(budget, apart)] = [prover.point_from_montgomery(bytes(pts)[size * e:size * e + size]) for e in range(E)]
        lib.zkwg_msm_destroy(h)
    gen = G.G if group == 1 else H.G2
    mul = G.mul if group == 1 else H.mul
    want = [mul(sum(s * ks[j] for s, j in zip(sc[e * n:(e + 1) * n], idx)) % R, gen) for e in range(E)]
    assert all(v == want for v in got.values()), {k: v == want for k, v in got.items()}


def test_library_exports_the_msm_entry_points_and_refuses_without_a_device():
    import ctypes as C
    import zkwg
    from zkwg import _lib
    lib = _lib.load()
    for name in ("zkwg_msm_create", "zkwg_msm_destroy", "zkwg_msm_work_bytes", "zkwg_msm_window_bits", "zkwg_msm_g1_device", "zkwg_msm_create_ex",
                 "zkwg_msm_enqueue_batch_device", "zkwg_msm_classify_device", "zkwg_msm_enqueue_lists_device", "zkwg_msm_lists_bytes"):
        assert hasattr(lib, name)
    h = C.c_void_p()
    assert lib.zkwg_msm_create(-1, bytes(64), 1, 0, C.byref(h)) != 0          # no CPU fallback
    assert zkwg.Msm.pack_bases([None, (1, 2)])[:64] == bytes(64)
