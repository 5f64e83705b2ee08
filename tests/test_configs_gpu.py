"""GPU parity at BASELINE.json's configurations, against the fast C oracle (itself pinned to the
literal oracle by tests/test_oracle_c.py):
  configs[1] batch=256 EmailVerifier(1024,1536): every witness bit-exact
  configs[2] batch=4096 (sampled rows bit-exact + all statuses), through the two-phase device API
  configs[4] maxBody=65536 long-body stress: bit-exact on a small batch
plus size-independent properties (w[0]=1, public signals, digest of digests)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def _oracle(N, M, fields, n, threads=8):
    from oracle import coracle
    lib = coracle.load()
    W, _, _ = coracle.run_fields(N, M, 0, {k: (v[:1] if isinstance(v, list) else v[:len(v) // n]) for k, v in fields.items()}, 1)
    buf = (C.c_uint8 * (n * 32 * W))()
    W, st, _ = coracle.run_fields(N, M, 0, fields, n, threads=threads, out=buf)
    return W, st, buf


def test_config1_batch256_bit_exact():
    import zkwg
    from zkwg import synth
    N, M, n = 1024, 1536, 256
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    recs, fields = synth.packed_batch(c, seed=21, n=n, body_len=1024)
    # ragged bodies: every 4th email gets a different body length
    recs2, fields2 = synth.packed_batch(c, seed=22, n=n // 4, body_len=300)
    wit, status = c.calculate_batch_host(recs + recs2)
    assert status == [0] * (n + n // 4)
    W, st, buf = _oracle(N, M, fields, n)
    assert W == c.W and st == [0] * n
    assert wit[:n * c.witness_bytes] == bytes(buf)
    W, st, buf2 = _oracle(N, M, fields2, n // 4)
    assert wit[n * c.witness_bytes:] == bytes(buf2)


def test_config2_batch4096_two_phase_device_api():
    import torch
    import zkwg
    from zkwg import synth
    N, M, n, distinct, tile = 1024, 1536, 4096, 64, 512
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    recs, fields = synth.packed_batch(c, seed=31, n=distinct, body_len=1024)
    dev = torch.device("cuda:0")
    h = torch.frombuffer(bytearray(recs), dtype=torch.uint8).view(distinct, c.in_stride)
    d_in = h.repeat(n // distinct, 1).contiguous().to(dev)
    # tamper one email of the big batch: its status must be 4, everyone else's 0
    off = c.lib.zkwg_input_offset(c.h, 1)
    d_in[1234, off] ^= 1
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device=dev)
    d_out = torch.empty(tile * c.witness_bytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream()
    c.prepare_device(d_in, n, d_st, d_scr, st)
    W, ost, buf = _oracle(N, M, fields, distinct)
    ref = bytes(buf)
    wb = c.witness_bytes
    first = None
    for t in range(n // tile):
        c.expand_device(d_in, n, d_scr, t * tile, tile, d_out, st)
        torch.cuda.synchronize()
        rows = d_out.view(tile, wb)
        if t == 0:
            # the first `distinct` rows: full bit-exact compare against the oracle
            host = rows[:distinct].cpu().numpy().tobytes()
            assert host == ref
            first = rows.clone()
        else:
            # replicated inputs => replicated witnesses: every tile equals tile 0 (except the tampered row)
            same = (rows == first).all(dim=1)
            bad = (~same).nonzero().flatten().tolist()
            assert bad == ([1234 - t * tile] if t * tile <= 1234 < (t + 1) * tile else [])
        assert int(rows[:, 0].sum().item()) == tile  # w[0] = 1 in every witness (low byte)
    status = d_st.cpu().tolist()
    assert status[1234] == 4 and sum(status) == 4


def test_config4_long_body_65536():
    import zkwg
    from zkwg import synth
    N, M, n = 1024, 65536, 3
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    assert c.W > 30_000_000
    recs, fields = synth.packed_batch(c, seed=41, n=n, body_len=40000)
    wit, status = c.calculate_batch_host(recs)
    assert status == [0] * n
    W, st, buf = _oracle(N, M, fields, n, threads=3)
    assert W == c.W and st == [0] * n
    assert wit == bytes(buf)


def test_prover_handoff_montgomery_round_trip():
    # zkwg_convert_montgomery_device: x -> x * 2^256 mod r in place on a device witness, and back
    import torch
    import zkwg
    P = zkwg.FIELD_MODULUS if hasattr(zkwg, "FIELD_MODULUS") else 21888242871839275222246405745257275088548364400416034343698204186575808495617
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0)   # the Fr-richest small witness
    from test_rsa_cpu import KAT_MSG, KAT_SIG, KAT_PUB, limbs
    rec = c.pack({"message": KAT_MSG, "signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB)})
    wit, status = c.calculate_batch_host(rec)
    assert status == [0]
    d = torch.frombuffer(bytearray(wit), dtype=torch.uint8).to("cuda:0")
    zkwg.convert_montgomery_device(d, c.W, True)
    torch.cuda.synchronize()
    m = d.cpu().numpy().tobytes()
    R = (1 << 256) % P
    for i in list(range(0, 64)) + list(range(c.W - 3000, c.W)):
        x = int.from_bytes(wit[32 * i:32 * i + 32], "little")
        assert int.from_bytes(m[32 * i:32 * i + 32], "little") == x * R % P, i
    zkwg.convert_montgomery_device(d, c.W, False)
    torch.cuda.synchronize()
    assert d.cpu().numpy().tobytes() == wit
