"""Generic field-element input path (include/zkwg.h zkwg_pack_field).

`CircuitInput` values are arbitrary decimal strings (packages/helpers/src/input-generators.ts:6-18).
A byte slot holding 256, or a length equal to r-1, is a legal *input* that the circuit itself rejects:
`Num2Bits(8)` of lib/sha.circom:27 / `Num2Bits(log2Ceil(maxHeadersLength))` of email-verifier.circom:58
fail and circom_runtime throws "Assert Failed".  The packers therefore keep such values (low bits +
a range flag in the record) instead of refusing them; the kernels fail the email with status 4."""
import json
import os

import pytest

from conftest import ROOT

CASE = json.load(open(os.path.join(ROOT, "tests", "golden", "ev_576_192_case.json")))
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _variants():
    inp = CASE["input"]
    v = {}
    v["header byte 256"] = dict(inp, emailHeader=["256"] + inp["emailHeader"][1:])
    v["header length r-1"] = dict(inp, emailHeaderLength=str(R - 1))
    v["body byte 2^40"] = dict(inp, emailBody=inp["emailBody"][:5] + [str(1 << 40)] + inp["emailBody"][6:])
    v["precomputedSHA 300"] = dict(inp, precomputedSHA=inp["precomputedSHA"][:31] + ["300"])
    v["pubkey limb 2^200"] = dict(inp, pubkey=inp["pubkey"][:3] + [str(1 << 200)] + inp["pubkey"][4:])
    v["signature limb r-5"] = dict(inp, signature=[str(R - 5)] + inp["signature"][1:])
    v["bodyHashIndex 2^33"] = dict(inp, bodyHashIndex=str(1 << 33))
    v["body length 2^32"] = dict(inp, emailBodyLength=str(1 << 32))
    return v


def test_oracle_rejects_and_packer_flags_out_of_range_inputs():
    import zkwg
    from oracle.pyref import zkemail as zk, comp
    N, M = CASE["maxHeader"], CASE["maxBody"]
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=-1)
    off = c.lib.zkwg_input_offset(c.h, 12)
    good = c.pack(CASE["input"])
    assert good[off:off + 4] == b"\0\0\0\0"
    for name, inp in _variants().items():
        rec = c.pack(inp)
        assert int.from_bytes(rec[off:off + 4], "little") != 0, name
        ii = {k: [int(x) for x in v] if isinstance(v, list) else int(v) for k, v in inp.items()}
        with pytest.raises(comp.AssertFailed):
            zk.EmailVerifier(N, M, 121, 17, 0, ii, body_hash_regex=lambda m: zk.BodyHashRegexV1(N, m))
    # a value >= r is first reduced mod r (circom_runtime normalize): 255 + r is the byte 255
    wrapped = dict(CASE["input"], emailHeader=[str(int(CASE["input"]["emailHeader"][0]) + R)] + CASE["input"]["emailHeader"][1:])
    assert c.pack(wrapped) == good


@pytest.mark.skipif(not os.path.isdir("/root/reference/packages/circuits"), reason="/root/reference is not present")
def test_reference_circuit_rejects_the_same_inputs():
    # the reference's own email-verifier.circom through the interpreter: "Assert Failed" for two of them
    from oracle.circom import ev, AssertFailed
    prog = ev.email_verifier(CASE["maxHeader"], CASE["maxBody"])
    v = _variants()
    for name in ("header byte 256", "header length r-1"):
        with pytest.raises(AssertFailed):
            prog.run(v[name])


@pytest.mark.gpu
def test_out_of_range_inputs_fail_with_status_4_on_the_gpu():
    import hashlib
    import zkwg
    N, M = CASE["maxHeader"], CASE["maxBody"]
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    vs = _variants()
    recs = c.pack(CASE["input"]) + b"".join(c.pack(i) for i in vs.values()) + c.pack(CASE["input"])
    wit, status = c.calculate_batch_host(recs)
    assert status == [0] + [4] * len(vs) + [0]
    wb = c.witness_bytes
    assert hashlib.sha256(wit[:wb]).hexdigest() == CASE["witnessSha256"]
    assert wit[-wb:] == wit[:wb]
    wc = zkwg.WitnessCalculator(c)
    with pytest.raises(Exception, match="Assert Failed"):
        wc.calculateWitness(vs["header byte 256"])
    # RSA main (no SHA kernels): a 129-bit message limb fails messageN2B (lib/rsa.circom:118)
    from test_rsa_cpu import KAT_MSG, KAT_SIG, KAT_PUB, limbs
    r = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=0)
    ok = r.pack({"message": KAT_MSG, "signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB)})
    bad = r.pack({"message": [KAT_MSG[0] + (1 << 128)] + KAT_MSG[1:], "signature": limbs(KAT_SIG), "modulus": limbs(KAT_PUB)})
    _, st = r.calculate_batch_host(ok + bad, want_witness=False)
    assert st == [0, 4]
