"""zkwg_expand_host / zkwg_set_host_expand (include/zkwg.h): the witness expanded in HOST memory from the downloaded
0.45 MB image by the same segment decoders zk_expand runs on the device -- the delivered-to-host route that does not
push 56.9 MB per email through PCIe (consumer: snarkjs on the host, packages/helpers/src/chunked-zkey.ts:80-84).
Bytes must equal the device expansion's (and therefore the oracle's, tests/test_ev_gpu.py)."""
import hashlib
import json
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _ev_records(c, n):
    from zkwg import synth
    recs, _ = synth.packed_batch(c, seed=77, n=n, body_len=100)
    return recs


@pytest.mark.parametrize("flags", [{}, {"enable_header_masking": 1, "enable_body_masking": 1}, {"remove_soft_line_breaks": 1}])
def test_host_expansion_equals_device_expansion_email_verifier(flags):
    import zkwg
    N, M, n = 576, 192, 7
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, **flags)
    case = json.load(open(os.path.join(ROOT, "tests", "golden", "ev_576_192_case.json")))
    if flags:
        recs = _ev_records(c, n)            # all-zero masks / decoded body: the segment types of the variants are what matters here
    else:
        good = c.pack(case["input"])
        bad_inp = dict(case["input"], emailHeader=list(case["input"]["emailHeader"]))
        bad_inp["emailHeader"][10] = str(int(bad_inp["emailHeader"][10]) ^ 1)
        recs = good * 3 + c.pack(bad_inp) + good * 3
    wit_dev, st_dev = c.calculate_batch_host(recs, max_tile=3)
    c.set_host_expand(4)
    wit_host, st_host = c.calculate_batch_host(recs, max_tile=3)       # 3 tiles: 3 + 3 + 1 emails
    c.set_host_expand(0)
    assert st_host == st_dev
    wb = c.witness_bytes
    for i in range(n):
        if st_dev[i] == 0:
            assert wit_host[i * wb:(i + 1) * wb] == wit_dev[i * wb:(i + 1) * wb], i
    if not flags:
        assert st_dev == [0, 0, 0, 4, 0, 0, 0]
        assert hashlib.sha256(wit_host[:wb]).hexdigest() == case["witnessSha256"]


def test_host_expansion_rsa_and_sha_mains_and_sym_layout():
    import random
    import zkwg
    from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG, limbs
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0)
    rec = c.pack({"message": KAT_MSG, "signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB)})
    wit_dev, st = c.calculate_batch_host(rec * 3)
    c.set_host_expand(2)
    wit_host, st2 = c.calculate_batch_host(rec * 3, max_tile=2)
    assert st == st2 == [0, 0, 0] and wit_host == wit_dev
    # a shuffled `.sym` order (segments split into runs, ZkSeg::r0): the host walks the same remapped table
    c0 = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=128, max_body=0, device=-1)
    names = [nm for _, nm in c0.symbols()]
    order = list(range(1, len(names)))
    rng = random.Random(5)
    blocks = [order[i:i + 700] for i in range(0, len(order), 700)]
    rng.shuffle(blocks)
    sym = "".join(f"{k},{k},0,{names[i]}\n" for k, i in enumerate((i for b in blocks for i in b), start=1))
    cs = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=128, max_body=0, device=0, sym=sym)
    from conftest import sha_pad
    padded, ln = sha_pad(b"hello world", 128)
    rec = cs.pack({"paddedIn": list(padded), "paddedInLength": ln})
    wd, s1 = cs.calculate_batch_host(rec * 2)
    cs.set_host_expand(3)
    wh, s2 = cs.calculate_batch_host(rec * 2)
    assert s1 == s2 == [0, 0] and wd == wh
