"""tools/intake.py (the one-command intake for real circom / snarkjs artefacts) on the artefacts this repo can make
offline: the interpreter-generated `--O0` files of the RSA main (tests/golden/o0_rsa.*) and of EmailVerifier(576,192)
(artifacts/, built where /root/reference exists).  A `.wtns` written from the interpreter's values stands in for the
snarkjs file; a corrupted copy must be reported at the right signal."""
import gzip
import json
import os
import struct

import pytest

import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))      # tests/intake.py (tools/intake.py is its launcher), imported here as a module

REF = os.path.isdir("/root/reference/packages/circuits")
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _wtns(values):
    body = b"".join(int(v).to_bytes(32, "little") for v in values)
    return (b"wtns" + struct.pack("<II", 2, 2) + struct.pack("<IQ", 1, 40) + struct.pack("<I", 32) + P.to_bytes(32, "little") +
            struct.pack("<I", len(values)) + struct.pack("<IQ", 2, len(body)) + body)


def _interpreter_values(kind):
    from oracle.circom import ev
    from oracle.circom.runtime import iter_signals
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "o0_rsa.json")))
    prog = ev.program("tests/test-circuits/rsa-test.circom")
    root = prog.run(meta["inputs"])
    return [1] + [v for _, v, _, _ in iter_signals(root, with_names=False)]


@pytest.mark.skipif(not REF, reason="the interpreter executes /root/reference's circuits")
def test_intake_on_the_rsa_artefacts_without_a_gpu(tmp_path):
    import intake
    vals = _interpreter_values("rsa")
    good = tmp_path / "good.wtns"
    good.write_bytes(_wtns(vals))
    args = ["--build-dir", os.path.join(ROOT, "tests", "golden"), "--name", "o0_rsa", "--input", os.path.join(ROOT, "tests", "golden", "o0_rsa.json"),
            "--main-kind", "rsa", "--device", "-1"]
    rep_path = tmp_path / "rep.json"
    assert intake.main(args + ["--wtns", str(good), "--json", str(rep_path)]) == 0
    rep = json.load(open(rep_path))
    assert rep["ok"] and rep["steps"]["interpreter"]["signals"] == 205713
    assert rep["steps"][".sym names"]["sym_signals"] == 205712 and rep["steps"]["product handle"]["witness_len"] == 205713
    assert rep["steps"][".wtns vs interpreter"]["ok"]
    # one wrong value in the file: reported by name at that index
    vals[12345] = (vals[12345] + 1) % P
    bad = tmp_path / "bad.wtns"
    bad.write_bytes(_wtns(vals))
    assert intake.main(args + ["--wtns", str(bad), "--json", str(rep_path)]) == 1
    rep = json.load(open(rep_path))
    assert "(index 12345)" in rep["steps"][".wtns vs interpreter"]["first_difference"]
    # a .sym that keeps a signal the schedule cannot derive: the product refuses, the tool says why
    sym = gzip.open(os.path.join(ROOT, "tests", "golden", "o0_rsa.sym.gz"), "rb").read().decode()
    bd = tmp_path / "build"
    bd.mkdir()
    (bd / "x.sym").write_text(sym + "205713,205713,0,main.not_a_signal\n")
    (bd / "x.r1cs").write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", "o0_rsa.r1cs.gz"), "rb").read())
    assert intake.main(["--build-dir", str(bd), "--input", os.path.join(ROOT, "tests", "golden", "o0_rsa.json"), "--main-kind", "rsa",
                        "--device", "-1", "--json", str(rep_path)]) == 1
    rep = json.load(open(rep_path))
    assert rep["steps"][".sym names"]["first_only_in_sym"] == "main.not_a_signal"
    assert not rep["steps"]["product handle"]["ok"] and "main.not_a_signal" in rep["steps"]["product handle"]["error"] or "wires" in rep["steps"]["product handle"]["error"]


@pytest.mark.gpu
def test_intake_full_flow_on_the_gpu_from_a_saved_interpreter_run(tmp_path):
    """Steps 2-5 on a box without /root/reference: the interpreter's run of the RSA main comes from
    tests/golden/intake_rsa_interpreter.npz (made here by `tools/intake.py ... --dump-interpreter`); the product handle is
    built from the `.sym` + `.r1cs`, its DEVICE witness is compared with the interpreter's and with a `.wtns`, and
    checkConstraints runs on the device."""
    import intake
    import numpy as np
    dump = os.path.join(ROOT, "tests", "golden", "intake_rsa_interpreter.npz")
    raw = bytes(np.load(dump)["values"])
    vals = [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(len(raw) // 32)]
    wt = tmp_path / "w.wtns"
    wt.write_bytes(_wtns(vals))
    rep_path = tmp_path / "rep.json"
    args = ["--build-dir", os.path.join(ROOT, "tests", "golden"), "--name", "o0_rsa", "--input", os.path.join(ROOT, "tests", "golden", "o0_rsa.json"),
            "--main-kind", "rsa", "--device", "0", "--interpreter-dump", dump, "--json", str(rep_path)]
    assert intake.main(args + ["--wtns", str(wt)]) == 0
    rep = json.load(open(rep_path))
    for k in ("interpreter (saved run)", ".sym names", ".wtns vs interpreter", "product handle", "product vs interpreter", "product vs .wtns",
              "checkConstraints (product witness, device)"):
        assert rep["steps"][k]["ok"], k
    assert rep["steps"]["checkConstraints (product witness, device)"]["constraints"] == 208463
    # a .wtns with one wrong value: the product disagrees with it exactly there
    vals[777] = (vals[777] + 1) % P
    wt.write_bytes(_wtns(vals))
    assert intake.main(args + ["--wtns", str(wt)]) == 1
    rep = json.load(open(rep_path))
    assert "(index 777)" in rep["steps"]["product vs .wtns"]["first_difference"] and rep["steps"]["product vs interpreter"]["ok"]


@pytest.mark.gpu
@pytest.mark.skipif(not (REF and os.path.exists(os.path.join(ROOT, "artifacts", "o0_ev_576_192.json"))),
                    reason="needs /root/reference (interpreter) and artifacts/o0_ev_576_192.*")
def test_intake_full_flow_on_the_email_verifier_artefacts(tmp_path):
    import intake
    tmpl_root = os.path.join(ROOT, "oracle", "circom", "lib")     # node_modules-shaped: @zk-email/zk-regex-circom + circomlib
    rep_path = tmp_path / "rep.json"
    rc = intake.main(["--node-modules", tmpl_root, "--build-dir", os.path.join(ROOT, "artifacts"), "--name", "o0_ev_576_192",
                      "--input", os.path.join(ROOT, "artifacts", "o0_ev_576_192.json"), "--max-header", "576", "--max-body", "192",
                      "--device", "0", "--json", str(rep_path)])
    rep = json.load(open(rep_path))
    assert rc == 0, rep
    for k in ("interpreter", ".sym names", "product handle", "product vs interpreter", "checkConstraints (product witness, device)"):
        assert rep["steps"][k]["ok"], k
    assert rep["steps"]["product handle"]["regex_template"].endswith("body_hash_regex.circom")
