"""CPU tests: the C-ABI library loads, exports every declared symbol, and its layout
(witness length + symbol table) agrees with the Python oracle's kept-signal walk."""
import re
import os

import pytest

from conftest import ROOT, sha_pad


def test_library_exports_every_declared_symbol():
    from zkwg import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "zkwg.h")).read()
    declared = set(re.findall(r"\b(zkwg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"libzkwg.so does not export {name}"
    assert set(_lib.EXPORTS) == declared
    assert lib.zkwg_abi_version() == 1


def test_layout_only_handle_needs_no_gpu():
    import zkwg
    c = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=128, max_body=0, device=-1)
    assert c.W > 0
    with pytest.raises(zkwg.ZkwgError):
        c.calculate_batch_host(bytes(c.in_stride))


def test_sha_main_layout_matches_oracle():
    import zkwg
    from oracle.pyref import zkemail as zk, comp
    N = 128
    p, n = sha_pad(b"hello world", N)
    main = zk.Sha256Bytes(N, list(p), n, is_main=True)
    sym_oracle = comp.symbols_kept(main)
    c = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=N, max_body=0, device=-1)
    assert c.W == len(sym_oracle)
    assert c.symbols() == sym_oracle
    assert c.n_public == 256 + N + 1
