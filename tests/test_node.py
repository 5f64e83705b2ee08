"""The Node/N-API host side (zk-email-verify_amd/js): CPU smoke here, GPU run on the box."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

JS = os.path.join(ROOT, "zk-email-verify_amd", "js")
needs_node = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists(os.path.join(JS, "zkwg_addon.node")),
                                reason="node or the built addon is missing")


@needs_node
def test_node_addon_cpu():
    out = subprocess.run(["node", os.path.join(JS, "test_cpu.js")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "js cpu ok" in out.stdout


@needs_node
@pytest.mark.gpu
def test_node_addon_gpu():
    out = subprocess.run(["node", os.path.join(JS, "test_gpu.js")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "js gpu ok" in out.stdout


@pytest.mark.gpu
def test_python_generate_witness_cli(tmp_path):
    """`python -m zkwg.generate_witness <circuit> input.json witness.wtns`: the documented CLI of the reference
    (docs/zk-email-docs/UsageGuide/README.md:132-140) with the circuit named by its template parameters."""
    import hashlib
    import json
    import sys
    kase = json.load(open(os.path.join(ROOT, "tests", "golden", "ev_576_192_case.json")))
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "zk-email-verify_amd", "py"))
    spec = f"EmailVerifier({kase['maxHeader']},{kase['maxBody']},121,17,0,0,0,0)"
    (tmp_path / "input.json").write_text(json.dumps(kase["input"]))
    r = subprocess.run([sys.executable, "-m", "zkwg.generate_witness", spec, str(tmp_path / "input.json"), str(tmp_path / "w.wtns")],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    wt = (tmp_path / "w.wtns").read_bytes()
    assert wt[:4] == b"wtns" and hashlib.sha256(wt[-32 * kase["witnessLen"]:]).hexdigest() == kase["witnessSha256"]
    bad = dict(kase["input"], emailHeader=list(kase["input"]["emailHeader"]))
    bad["emailHeader"][10] = str(int(bad["emailHeader"][10]) ^ 1)
    (tmp_path / "batch.json").write_text(json.dumps([kase["input"], bad]))
    r = subprocess.run([sys.executable, "-m", "zkwg.generate_witness", spec, str(tmp_path / "batch.json"), str(tmp_path / "b.wtns")],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 1 and "email 1: Error: Assert Failed" in r.stderr
    assert (tmp_path / "b_0.wtns").read_bytes() == wt and not (tmp_path / "b_1.wtns").exists()
