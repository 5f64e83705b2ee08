import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def sha_pad(msg: bytes, maxlen: int):
    """SHA-256 padding to `maxlen` (restates packages/helpers/src/sha-utils.ts:88-111)."""
    ln = len(msg) * 8
    r = msg + b"\x80"
    while (len(r) * 8 + 64) % 512:
        r += b"\0"
    r += ln.to_bytes(8, "big")
    n = len(r)
    assert n <= maxlen
    return r + b"\0" * (maxlen - n), n
