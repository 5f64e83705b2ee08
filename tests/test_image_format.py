"""The compact image as an interchange format (include/zkwg.h "compact image"): the scratch buffer of
zkwg_prepare_device + the input record + the static segment table determine every witness slot.  A decoder written
from the documented format alone (tests/hosttest.py `expand`, pure Python) must reproduce the witness the device
writes -- for the SHA-256 and the RSA mains (all segment types but the EmailVerifier-only DFA / mask / RSLB ones)."""
import ctypes as C

import pytest

import hosttest


class _Fmt:
    """adapter: the documented accessors in the shape hosttest.expand expects"""

    def __init__(self, c):
        self.c = c
        self.segs = [hosttest.Seg(*t, 0) for t in c.segment_table()]

    def ht_W(self, h): return self.c.W
    def ht_segs(self, h): return self.segs
    def ht_nsegs(self, h): return len(self.segs)
    def ht_inv_half(self, h): return self.c.lib.zkwg_inverse_table_half(self.c.h)


def test_layout_accessors_on_a_layout_only_handle():
    import zkwg
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, device=-1)
    lay = c.image_layout(512)
    assert lay["total_bytes"] == c.scratch_bytes(512)
    assert lay["off_hstates"] == 0 < lay["off_bits"] < lay["off_small"] < lay["off_fr"] < lay["total_bytes"]
    per_email = 8 * lay["bits_words"] + 4 * lay["small_words"] + 32 * lay["fr_elems"]
    assert per_email < c.witness_bytes // 100            # the image is < 1 % of the witness
    segs = c.segment_table()
    pos = 0
    for slot, n, typ, *_ in segs:                          # the table tiles [0, W)
        assert slot == pos and n > 0 and typ < 19
        pos += n
    assert pos == c.W


@pytest.mark.gpu
@pytest.mark.parametrize("main", ["rsa", "sha"])
def test_decoding_the_device_image_reproduces_the_device_witness(main):
    import numpy as np
    import torch
    import zkwg
    from conftest import sha_pad
    if main == "rsa":
        from test_rsa_cpu import KAT_MSG, KAT_SIG, KAT_PUB, limbs
        c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0)
        recs = c.pack({"message": KAT_MSG, "signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB)})
    else:
        c = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=128, max_body=0, device=0)
        p, n = sha_pad(b"hello world", 128)
        recs = c.pack({"paddedIn": list(p), "paddedInLength": n})
    n = 2
    recs = recs * n
    dev = torch.device("cuda:0")
    d_in = torch.frombuffer(bytearray(recs), dtype=torch.uint8).view(n, c.in_stride).to(dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device=dev)
    d_out = torch.empty(n * c.witness_bytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream()
    c.prepare_device(d_in, n, d_st, d_scr, st)
    c.expand_device(d_in, n, d_scr, 0, n, d_out, st)
    torch.cuda.synchronize()
    assert d_st.cpu().tolist() == [0] * n
    scr = d_scr.cpu().numpy().tobytes()
    wit = d_out.cpu().numpy().tobytes()
    lay = c.image_layout(n)
    e = 1                                                     # decode the second email of the batch
    bits = np.frombuffer(scr, dtype=np.uint64, count=lay["bits_words"], offset=lay["off_bits"] + 8 * lay["bits_words"] * e)
    small = np.frombuffer(scr, dtype=np.uint32, count=lay["small_words"], offset=lay["off_small"] + 4 * lay["small_words"] * e)
    fr = scr[lay["off_fr"] + 32 * lay["fr_elems"] * e:lay["off_fr"] + 32 * lay["fr_elems"] * (e + 1)]
    rec = recs[e * c.in_stride:(e + 1) * c.in_stride]
    got = hosttest.expand(_Fmt(c), None, rec, [int(x) for x in bits], [int(x) for x in small], fr)
    wb = c.witness_bytes
    exp = [int.from_bytes(wit[e * wb + 32 * i:e * wb + 32 * i + 32], "little") for i in range(c.W)]
    assert got == exp
