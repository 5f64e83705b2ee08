"""Committed golden digests (tests/golden/witness_digests.json, produced by the literal Python oracle):
the fast C oracle on CPU and the HIP path on the GPU must reproduce them bit for bit."""
import hashlib
import json
import os

import pytest

from conftest import ROOT

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "witness_digests.json")))


def _inputs(case):
    from zkwg import synth, inputs
    d = synth.synthetic_dkim_result(case["seed"], case["index"], body_len=case["body_len"])
    return inputs.generate_email_verifier_inputs_from_dkim_result(
        d, case["max_header"], case["max_body"], ignore_body_hash_check=bool(case["ignore_body_hash_check"]))


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f'{c["max_header"]}-{c["max_body"]}-{c["ignore_body_hash_check"]}')
def test_c_oracle_reproduces_golden_digest(case):
    from oracle import coracle
    wits, status, W = coracle.calculate(0, case["max_header"], case["max_body"], case["ignore_body_hash_check"], [_inputs(case)])
    assert status == [0] and W == case["W"]
    assert hashlib.sha256(wits[0]).hexdigest() == case["sha256"]
    assert int.from_bytes(wits[0][32:64], "little") == int(case["pubkeyHash"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f'{c["max_header"]}-{c["max_body"]}-{c["ignore_body_hash_check"]}')
def test_hip_path_reproduces_golden_digest(case):
    import zkwg
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=case["max_header"], max_body=case["max_body"],
                     ignore_body_hash_check=case["ignore_body_hash_check"], device=0)
    assert c.W == case["W"]
    b = zkwg.WitnessCalculator(c).calculateBinWitness(_inputs(case))
    assert hashlib.sha256(b).hexdigest() == case["sha256"]
    assert int.from_bytes(b[64:96], "little") == int(case["shaHi"]) and int.from_bytes(b[96:128], "little") == int(case["shaLo"])
