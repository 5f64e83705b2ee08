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
