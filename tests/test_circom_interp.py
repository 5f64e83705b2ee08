"""The circom interpreter (oracle/circom, TEST INFRASTRUCTURE) executes the reference's UNMODIFIED
circuits -- packages/circuits/{email-verifier.circom,lib,utils,helpers,tests/test-circuits} -- with
circomlib restated in circom syntax.  These tests

  * replay the reference's own circuit tests (packages/circuits/tests/*.test.ts known answers) on
    the reference's own test mains through the interpreter;
  * pin the literal Python oracle (oracle/pyref) against the interpreter on EVERY declared signal:
    names, values (and order where the two agree on component creation order);
  * check that the committed GPU fixture (tests/golden/circom_ev_576_192.npz) is what the
    interpreter produces.

They need /root/reference (present in the build container, absent on the GPU box -> skipped)."""
import hashlib
import json
import os
import random
import zlib

import numpy as np
import pytest

from conftest import ROOT, sha_pad
from oracle.circom import ev, AssertFailed, CircomError
from oracle.circom.compare import diff, flat_walk_kept

pytestmark = pytest.mark.skipif(not ev.reference_available(), reason="/root/reference is not present")

TC = "tests/test-circuits/"


def limbs(x, n=121, k=17):
    return [(x >> (n * i)) & ((1 << n) - 1) for i in range(k)]


def by_name(root, prog):
    return {nm: v for nm, v, _, _ in flat_walk_kept(root, prog.templates_src)}


def test_stand_in_regex_file_is_current():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_body_hash_regex", os.path.join(ROOT, "tools", "gen_body_hash_regex.py"))
    gen_body_hash_regex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen_body_hash_regex)
    assert gen_body_hash_regex.main(["--check"]) == 0


def test_sha_test_circom_every_o0_signal_equals_pyref_in_order():
    # sha.test.ts:26-41 on tests/test-circuits/sha-test.circom (Sha256Bytes(640), public inputs)
    from oracle.pyref import zkemail as zk
    prog = ev.program(TC + "sha-test.circom")
    for m in (b"hello world", b""):   # b"0" of sha.test.ts is covered at pyref level
        p, n = sha_pad(m, 640)
        root = prog.run({"paddedIn": list(p), "paddedInLength": n})
        out = root.sigs["out"].vals
        assert int("".join(map(str, out)), 2).to_bytes(32, "big") == hashlib.sha256(m).digest()
        if m == b"hello world":
            main = zk.Sha256Bytes(640, list(p), n, is_main=True)
            main.public = {"paddedIn", "paddedInLength"}
            n_cmp, n_bad, first = diff(root, main)
            assert n_cmp == 2073896 and n_bad == 0, first


def test_rsa_test_circom_kat_and_pyref_names_values():
    # rsa.test.ts:64-144 on tests/test-circuits/rsa-test.circom
    from oracle.pyref import zkemail as zk
    from test_rsa_cpu import KAT_MSG, KAT_PUB, KAT_SIG
    prog = ev.program(TC + "rsa-test.circom")
    inp = {"message": KAT_MSG, "signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB)}
    root = prog.run(inp)
    main = zk.RSAVerifier65537(121, 17, KAT_MSG, limbs(KAT_SIG), limbs(KAT_PUB), is_main=True)
    got = by_name(root, prog)
    exp = {nm: v for nm, v, _ in main.walk()}
    assert got == exp and len(exp) == 205712
    with pytest.raises(AssertFailed):
        prog.run(dict(inp, message=[KAT_MSG[0] + 1] + KAT_MSG[1:]))
    # creation order differs from pyref's array-grouped order where `= T()` statements interleave
    # (lib/rsa.circom:116-125, lib/fp.circom:41-55); documented in DESIGN.md, not an error
    n_cmp, n_bad, _ = diff(root, main)
    assert n_bad > 0


def test_fp_mul_test_circom():
    # fp-mul.test.ts:34-46 / :48-64
    from oracle.pyref import zkemail as zk
    prog = ev.program(TC + "fp-mul-test.circom")
    root = prog.run({"a": [1, 0, 1, 0], "b": [0, 1, 1, 0], "p": [1, 1, 1, 1]})
    assert root.sigs["out"].vals == [0, 0, 0, 0]
    exp = {nm: v for nm, v, _ in zk.FpMul(2, 4, [1, 0, 1, 0], [0, 1, 1, 0], [1, 1, 1, 1]).walk()}
    assert by_name(root, prog) == exp
    rc = ev.program(TC + "fp-mul-test-range-check.circom")
    with pytest.raises(AssertFailed):
        rc.run({"a": [4, 3], "b": [3, 2], "p": [5, 6], "q": [2, 0], "r": [8, 4]})


def test_small_reference_test_circuits():
    rng = random.Random(3)
    # base64.test.ts:21-31,43
    b64 = ev.program(TC + "base64-test.circom")
    for ch, val in [(65, 0), (90, 25), (97, 26), (122, 51), (48, 52), (57, 61), (43, 62), (47, 63), (61, 0)]:
        assert b64.run({"in": ch}).sigs["out"].vals == [val]
    for ch in (34, 64, 91, 44):
        with pytest.raises(AssertFailed):
            b64.run({"in": ch})
    # pack-bits.test.ts
    pb = ev.program(TC + "pack-bits-test.circom")
    h = hashlib.sha256(b"test data").digest()
    bits = [(b >> (7 - i)) & 1 for b in h for i in range(8)]
    assert pb.run({"in": bits}).sigs["out"].vals == [int.from_bytes(h[:16], "big"), int.from_bytes(h[16:], "big")]
    pb10 = ev.program(TC + "pack-bits-10.circom")
    assert pb10.run({"in": [1, 0, 1, 1, 0, 1, 0, 0, 1, 1]}).sigs["out"].vals == [0b101, 0b101, 0b001, 0b100]
    # select-regex-reveal.test.ts:22-120, SelectRegexReveal(34, 8)
    srr = ev.program(TC + "select-regex-reveal-test.circom")
    rev = [ord(c) for c in "zk email"]
    start = rng.randrange(24)
    inp = [0] * 34
    inp[start:start + 8] = rev
    assert srr.run({"in": inp, "startIndex": start}).sigs["out"].vals == rev
    with pytest.raises(AssertFailed):
        srr.run({"in": [0] * 34, "startIndex": 5})
    with pytest.raises(AssertFailed):
        srr.run({"in": inp, "startIndex": start + 1})
    # byte-mask.test.ts:19-47
    bm = ev.program(TC + "byte-mask-test.circom")
    assert bm.run({"in": list(range(1, 11)), "mask": [1, 0] * 5}).sigs["out"].vals == [1, 0, 3, 0, 5, 0, 7, 0, 9, 0]
    with pytest.raises(AssertFailed):
        bm.run({"in": list(range(1, 11)), "mask": [1, 2, 1, 0, 1, 0, 1, 0, 1, 0]})
    # split-bytes-to-words.test.ts:22-31 (vs bigIntToChunkedBytes: 256 bytes BE -> 17 x 121-bit LE limbs)
    sw = ev.program(TC + "split-bytes-to-words-test.circom")
    data = bytes(rng.randrange(256) for _ in range(256))
    assert sw.run({"in": list(data)}).sigs["out"].vals == limbs(int.from_bytes(data, "big"))


def test_remove_soft_line_breaks_and_poseidon_modular_test_circuits():
    from oracle.pyref import poseidon
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "rslb_kats.json")))
    prog = ev.program(TC + "remove-soft-line-breaks-test.circom")
    for case in kat["cases"]:     # remove-soft-line-breaks.test.ts, all 7 cases
        root = prog.run({"encoded": case["encoded"], "decoded": case["decoded"]})
        assert root.sigs["isValid"].vals == [case["isValid"]], case["name"]
    pm = ev.program(TC + "poseidon-modular-test.circom")   # poseidon-modular.test.ts:26-28 shape
    rng = random.Random(37)
    xs = [rng.randrange(1 << 53) for _ in range(37)]
    out = None
    for i in range(0, 37, 16):
        h = poseidon.poseidon_hash(xs[i:i + 16])
        out = h if out is None else poseidon.poseidon_hash([out, h])
    assert pm.run({"in": xs}).sigs["out"].vals == [out]


def test_email_verifier_equals_pyref_and_the_gpu_fixture():
    """EmailVerifier(576,192,121,17,0,0,0,0) from email-verifier.circom: every pyref O0 signal has the
    interpreter's value; the only extra interpreter signals are the linear Ark/Mix/MixS/MixLast wires
    of circomlib's optimised Poseidon; the committed fixture is reproduced; tampering -> Assert Failed."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_circom_fixture as mk
    from oracle.pyref import zkemail as zk
    fx = np.load(os.path.join(ROOT, "tests", "golden", "circom_ev_576_192.npz"))
    inp = json.loads(bytes(fx["inputs"]).decode())
    prog = ev.email_verifier(576, 192, track_how=True)
    root = prog.run(inp)
    got, how, o0 = {}, {}, {}
    for i, (k, v, h, _) in enumerate(flat_walk_kept(root, prog.templates_src)):
        got[k], how[k], o0[k] = v, h, i + 1
    iinp = {k: [int(x) for x in v] if isinstance(v, list) else int(v) for k, v in inp.items()}
    main = zk.EmailVerifier(576, 192, 121, 17, 0, iinp, body_hash_regex=lambda m: zk.BodyHashRegexV1(576, m))
    n_pyref = 0
    for nm, v, flag in main.walk():
        n_pyref += 1
        assert got[nm] == v, nm
        if flag == "H":
            assert how[nm] == "<--", nm
        elif flag == "U":
            assert how[nm] is None, nm
    assert n_pyref == 3111726 and len(got) == 3113237
    extra = {nm for nm in got} - {nm for nm, _, _ in main.walk()}
    assert all(".pEx.ark[" in nm or ".pEx.mix" in nm for nm in extra) and len(extra) == 1511
    # fixture: o0 index + value of every kept-v1 slot
    import zkwg
    sym = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1).symbols()
    assert int(fx["n_o0"][0]) == len(got) + 1
    o0_fx = fx["o0_index"]
    wit = zlib.decompress(bytes(fx["witness"]))
    order = np.argsort(o0_fx, kind="stable")
    for rank in range(0, len(sym), 97):
        slot = int(order[rank])
        name = sym[slot][1]
        if slot:
            assert o0[name] == int(o0_fx[slot])
            assert int.from_bytes(wit[32 * rank:32 * rank + 32], "little") == got[name]
    # tamper (email-verifier.test.ts:61-79 style): a flipped header byte must fail
    bad = dict(inp, emailHeader=list(inp["emailHeader"]))
    bad["emailHeader"][10] = str(int(bad["emailHeader"][10]) ^ 1)
    with pytest.raises(AssertFailed):
        ev.email_verifier(576, 192).run(bad)
