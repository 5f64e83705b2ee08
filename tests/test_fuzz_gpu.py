"""Robustness: garbage / adversarial input records must never crash or hang the kernels; every such
email is reported with status 4 ("Assert Failed") exactly when the C oracle rejects it too."""
import ctypes as C
import random
import struct

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_garbage_records_do_not_crash_and_status_matches_oracle():
    import zkwg
    from zkwg import synth
    from oracle import coracle
    N, M = 576, 192
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    off = [c.lib.zkwg_input_offset(c.h, f) for f in range(9)]
    rng = random.Random(77)
    good, fields = synth.packed_batch(c, seed=5, n=8, body_len=70)
    recs = []
    # fully random records
    for _ in range(24):
        recs.append(bytes(rng.randrange(256) for _ in range(c.in_stride)))
    # valid emails with one field corrupted
    for k in range(40):
        r = bytearray(good[(k % 8) * c.in_stride:(k % 8 + 1) * c.in_stride])
        mode = k % 10
        if mode == 0:
            r[off[6]:off[6] + 4] = struct.pack("<I", rng.choice([0, 1, 63, 65, N, N + 64, 2 ** 31, 2 ** 32 - 1]))
        elif mode == 1:
            r[off[7]:off[7] + 4] = struct.pack("<I", rng.choice([0, 64, M, M + 64, 2 ** 32 - 64]))
        elif mode == 2:
            r[off[8]:off[8] + 4] = struct.pack("<I", rng.choice([0, N - 1, N, N + 43, 2 ** 20, 2 ** 32 - 1]))
        elif mode == 3:
            r[off[3]:off[3] + 272] = bytes(272)                      # modulus = 0
        elif mode == 4:
            r[off[3]:off[3] + 272] = b"\x01" + bytes(271)            # modulus = 1
        elif mode == 5:
            r[off[3] + 15] = 0xFF                                    # limb >= 2^121
        elif mode == 6:
            r[off[4]:off[4] + 272] = bytes(rng.randrange(256) for _ in range(272))  # random signature limbs
        elif mode == 7:
            r[off[3]:off[3] + 272] = bytes(rng.randrange(256) for _ in range(16)) + bytes(256)  # tiny modulus
        elif mode == 8:
            for _ in range(30):
                r[off[0] + rng.randrange(N)] = rng.randrange(256)
        elif mode == 9:
            r[off[2]:off[2] + 32] = bytes(rng.randrange(256) for _ in range(32))  # wrong precomputed SHA
        recs.append(bytes(r))
    blob = b"".join(recs) + good
    wit, status = c.calculate_batch_host(blob)
    n = len(recs)
    assert status[n:] == [0] * 8
    assert set(status) <= {0, 4}

    # the C oracle must agree on accept / reject for every record
    def col(field, width):
        return b"".join(r[off[field]:off[field] + width] for r in recs)
    u32s = lambda f: (C.c_uint32 * n)(*[struct.unpack("<I", r[off[f]:off[f] + 4])[0] for r in recs])
    lib = coracle.load()
    ost = (C.c_int * n)()
    lib.zkwg_oracle_calculate(0, N, M, 0, n, col(0, N), u32s(6), col(1, M), u32s(7), col(2, 32), col(3, 272), col(4, 272), None,
                              u32s(8), None, 0, ost, 4)
    assert list(ost) == status[:n]


@pytest.mark.timeout(600)
def test_garbage_records_with_all_flags_status_matches_oracle():
    """Same robustness bar for the flag variants (masks + removeSoftLineBreaks): random records, valid emails
    with a corrupted decodedEmailBodyIn / mask / body byte -- accept / reject must equal the C oracle's."""
    import zkwg
    from zkwg import synth
    from oracle import coracle
    N, M = 576, 384
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, enable_header_masking=1,
                     enable_body_masking=1, remove_soft_line_breaks=1)
    off = [c.lib.zkwg_input_offset(c.h, f) for f in range(12)]
    rng = random.Random(99)
    good, fields = synth.packed_batch(c, seed=6, n=6, body_len=250)
    good = bytearray(good)
    for i in range(6):     # binary masks on the valid emails
        base = i * c.in_stride
        good[base + off[9]:base + off[9] + N] = bytes(rng.randrange(2) for _ in range(N))
        good[base + off[10]:base + off[10] + M] = bytes(rng.randrange(2) for _ in range(M))
    good = bytes(good)
    recs = [bytes(rng.randrange(256) for _ in range(c.in_stride)) for _ in range(10)]
    for k in range(30):
        r = bytearray(good[(k % 6) * c.in_stride:(k % 6 + 1) * c.in_stride])
        mode = k % 6
        if mode == 0:
            r[off[11] + rng.randrange(200)] ^= 1 + rng.randrange(255)          # decoded body differs
        elif mode == 1:
            r[off[11]:off[11] + M] = r[off[1]:off[1] + M]                      # soft breaks left in place
        elif mode == 2:
            r[off[10] + rng.randrange(M)] = 2 + rng.randrange(254)              # non-binary body mask
        elif mode == 3:
            r[off[9] + rng.randrange(N)] = 2 + rng.randrange(254)               # non-binary header mask
        elif mode == 4:
            p = rng.randrange(200)
            r[off[1] + p:off[1] + p + 3] = b"=\r\n"                             # body changed (hash + decoded mismatch)
        elif mode == 5:
            r[off[11]:off[11] + M] = bytes(M)                                   # all-zero decoded body
        recs.append(bytes(r))
    n = len(recs)
    wit, status = c.calculate_batch_host(b"".join(recs) + good, want_witness=False)
    assert status[n:] == [0] * 6 and set(status) <= {0, 4}

    def col(field, width):
        return b"".join(r[off[field]:off[field] + width] for r in recs)
    u32s = lambda f: (C.c_uint32 * n)(*[struct.unpack("<I", r[off[f]:off[f] + 4])[0] for r in recs])
    lib = coracle.load()
    ost = (C.c_int * n)()
    hm, bm, dec = col(9, N), col(10, M), col(11, M)
    lib.zkwg_oracle_set_masks(hm, bm)
    lib.zkwg_oracle_set_decoded(dec)
    lib.zkwg_oracle_calculate(0, N, M, 0, n, col(0, N), u32s(6), col(1, M), u32s(7), col(2, 32), col(3, 272), col(4, 272), None,
                              u32s(8), None, 0, ost, 8)
    lib.zkwg_oracle_set_masks(None, None)
    lib.zkwg_oracle_set_decoded(None)
    assert list(ost) == status[:n]
    assert 4 in status[10:n]
