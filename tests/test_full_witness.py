"""Complete (`--O0`-numbered) witnesses: zkwg_circuit_create_full (include/zkwg.h, csrc/zkwg_full.h).

The artefacts -- `.r1cs`, `.sym`, rename rules, input, digest of the complete witness -- come from the circom
interpreter executing the reference's UNMODIFIED sources with constraint generation (oracle/circom/symbolic.py,
oracle/circom/o0_artifacts.py): the RSAVerifier65537(121,17) main (`tests/test-circuits/rsa-test.circom`,
205,713 signals / 208,463 constraints) is committed under tests/golden/, EmailVerifier(576,192)
(3,113,238 / 3,131,414) is generated into artifacts/ by __graft_entry__.build() where /root/reference exists.
The product is handed only the `.sym` + `.r1cs` pair: it emits the signals its schedule produces at their `.sym`
index and derives every other one from the linear constraints of the `.r1cs`.  Also: the device witness satisfies
that `.r1cs` -- a constraint system that does NOT come from zkwg's own derivation (zkwg.r1cs)."""
import gzip
import hashlib
import json
import os

import pytest

from conftest import ROOT


def _load(base):
    meta = json.load(open(base + ".json"))
    return meta, gzip.open(base + ".sym.gz", "rb").read(), gzip.open(base + ".r1cs.gz", "rb").read()


RSA = os.path.join(ROOT, "tests", "golden", "o0_rsa")
EV = os.path.join(ROOT, "artifacts", "o0_ev_576_192")


def _host_complete(c, kept_bytes):
    buf = bytearray(32 * c.W)
    for slot, dst in enumerate(c.layout_map()):
        assert dst is not None
        buf[32 * dst:32 * dst + 32] = kept_bytes[32 * slot:32 * slot + 32]
    c.linear_complete_host(buf)
    return bytes(buf)


def test_rsa_main_complete_witness_from_sym_and_r1cs_on_the_host():
    import zkwg
    from oracle import coracle
    meta, sym, r1cs = _load(RSA)
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1, sym=sym, sym_alias=meta["alias"], r1cs=r1cs)
    assert c.W == meta["n_wires"] == 205713 and c.lib.zkwg_linear_rows(c.h) == 16809
    assert c.symbols()[1][1] == "main.modulus[0]"          # public input first, compiler-style names
    wit, st, W = coracle.calculate(2, 0, 0, 0, [meta["inputs"]], threads=1)
    assert st == [0]
    full = _host_complete(c, wit[0])
    assert hashlib.sha256(full).hexdigest() == meta["witness_sha256"]
    assert c.o0_gather_host(wit[0]) == full      # the linear plan over the kept-v1 witness (round 2's gather tables)
    # the tables the device kernels read since round 3 -- per-wire descriptors, integer / field rows, chains, pre-decoded
    # slots (csrc/zkwg_o0.h) -- evaluated on the host by the kernels' own decode functions over the image the host build of
    # the RSA core produces (zkwg_expand_full_host)
    import ctypes as C
    import hosttest
    from zkwg._lib import Config, MAIN_RSA_VERIFIER
    lib = hosttest.load()
    h = lib.ht_create(C.byref(Config(MAIN_RSA_VERIFIER, 0, 0, 121, 17, 0, 0, 0, 0, 0)))
    rec = c.pack({k: v for k, v in meta["inputs"].items()})
    lay = c.image_layout(1)
    raw = (C.c_uint8 * (lay["total_bytes"] + 256))()
    base = (-C.addressof(raw)) % 256
    at = lambda off: C.c_void_p(C.addressof(raw) + base + off)
    (C.c_uint32 * lay["small_words"]).from_address(at(lay["off_small"]).value)[lib.ht_m_one(h)] = 1
    assert lib.ht_run_rsa(h, rec, None, at(lay["off_bits"]), at(lay["off_small"]), at(lay["off_fr"])) == 1
    lib.ht_destroy(h)
    assert c.expand_full_host(rec, 1, at(0), 0, 1) == full
    for k, v in meta["sample"].items():
        assert int.from_bytes(full[32 * int(k):32 * int(k) + 32], "little") == int(v)
    # without the .r1cs the same file is refused: it numbers signals the schedule does not produce
    with pytest.raises(zkwg.ZkwgError, match="not produced by this schedule|exceeds the schedule"):
        zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1, sym=sym, sym_alias=meta["alias"])
    # a constraint system that does not define them is refused too (here: the constraint section emptied)
    import struct
    hdr_only = bytearray(r1cs)
    # keep the file but claim zero constraints: nothing defines the derived signals
    pos = r1cs.index(struct.pack("<I", 32) + (21888242871839275222246405745257275088548364400416034343698204186575808495617).to_bytes(32, "little"))
    hdr_only[pos + 36 + 24:pos + 36 + 28] = struct.pack("<I", 0)
    with pytest.raises(zkwg.ZkwgError, match="neither produced by this schedule nor defined"):
        zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1, sym=sym, sym_alias=meta["alias"],
                     r1cs=bytes(hdr_only))


@pytest.mark.skipif(not os.path.isdir("/root/reference/packages/circuits"), reason="/root/reference is not present")
def test_rsa_fixture_is_what_the_interpreter_generates(tmp_path):
    from oracle.circom import o0_artifacts
    meta0 = json.load(open(RSA + ".json"))
    meta = o0_artifacts.build("rsa", str(tmp_path), inputs=o0_artifacts.default_inputs("rsa"))
    for k in ("n_wires", "n_constraints", "witness_sha256", "alias", "sample"):
        assert meta[k] == meta0[k], k
    assert gzip.open(str(tmp_path / "o0_rsa.sym.gz")).read() == gzip.open(RSA + ".sym.gz").read()
    assert gzip.open(str(tmp_path / "o0_rsa.r1cs.gz")).read() == gzip.open(RSA + ".r1cs.gz").read()


@pytest.mark.skipif(not os.path.exists(EV + ".json"), reason="artifacts/o0_ev_576_192.* not built (needs /root/reference)")
def test_email_verifier_complete_witness_on_the_host():
    import zkwg
    from oracle import coracle
    meta, sym, r1cs = _load(EV)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1, sym=sym, sym_alias=meta["alias"], r1cs=r1cs)
    assert c.W == meta["n_wires"] == 3113238 and c.lib.zkwg_linear_rows(c.h) == 3113238 - 735631
    wit, st, W = coracle.calculate(0, 576, 192, 0, [meta["inputs"]], threads=1)
    assert st == [0] and W == 735631
    assert hashlib.sha256(_host_complete(c, wit[0])).hexdigest() == meta["witness_sha256"]
    assert hashlib.sha256(c.o0_gather_host(wit[0])).hexdigest() == meta["witness_sha256"]


@pytest.mark.gpu
def test_rsa_main_complete_witness_on_the_gpu_and_it_satisfies_the_r1cs():
    import zkwg
    meta, sym, r1cs = _load(RSA)
    c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0, sym=sym, sym_alias=meta["alias"], r1cs=r1cs)
    rec = c.pack(meta["inputs"])
    from test_rsa_cpu import KAT_MSG
    bad = c.pack(dict(meta["inputs"], message=[str(KAT_MSG[0] + 1)] + meta["inputs"]["message"][1:]))
    wit, status = c.calculate_batch_host(rec + bad + rec)
    assert status == [0, 4, 0]
    wb = c.witness_bytes
    assert hashlib.sha256(wit[:wb]).hexdigest() == meta["witness_sha256"]
    assert wit[2 * wb:] == wit[:wb]
    # checkConstraints against the interpreter-generated constraint system, on the device
    R = zkwg.R1cs(r1cs, device=0)
    assert R.n_constraints == meta["n_constraints"]
    fv = R.first_violations(wit, 3)
    assert fv[0] is None and fv[2] is None and fv[1] is not None     # the failing email violates a constraint


STAND_IN = os.path.join(ROOT, "zk-email-verify_amd", "data", "templates", "zk-regex-circom", "circuits", "common", "body_hash_regex.circom")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(EV + ".json"), reason="artifacts/o0_ev_576_192.* not built (needs /root/reference)")
@pytest.mark.parametrize("regex", [None, STAND_IN], ids=["built-in regex", "regex from the template file"])
def test_email_verifier_complete_witness_on_the_gpu_and_it_satisfies_the_r1cs(regex):
    """(second case: the route INTEGRATION.md section 3 describes for the real artefacts -- the regex template, the
    compiler's `.sym` and `.r1cs` together, zkwg_circuit_create_regex)"""
    import zkwg
    meta, sym, r1cs = _load(EV)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=0, sym=sym, sym_alias=meta["alias"], r1cs=r1cs,
                     regex=regex)
    assert c.W == meta["n_wires"]
    wit, status = c.calculate_batch_host(c.pack(meta["inputs"]) * 2)
    assert status == [0, 0]
    wb = c.witness_bytes
    assert hashlib.sha256(wit[:wb]).hexdigest() == meta["witness_sha256"]
    assert wit[wb:] == wit[:wb]
    R = zkwg.R1cs(r1cs, device=0)
    assert R.n_constraints == meta["n_constraints"] == 3131414
    assert R.first_violations(wit, 2) == [None, None]
