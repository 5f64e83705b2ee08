"""HIP witness in the circom interpreter's signal order vs the interpreter's witness.

tests/golden/circom_ev_576_192.npz (made by tests/golden/make_circom_fixture.py from the reference's
unmodified `.circom` sources, checked on CPU by tests/test_circom_interp.py) holds, for
EmailVerifier(576,192,121,17,0,0,0,0) on one synthetic email: the O0 signal index the interpreter
assigns to every signal the kept-v1 layout keeps, and the interpreter's values of those signals in
O0 order.  Here the product is handed a `.sym` with exactly that order (labelIdx = O0 index,
witnessIdx = rank among the kept signals -- what the compiler writes when it eliminates the other
signals) and its witness must equal the interpreter's byte for byte."""
import json
import os
import zlib

import numpy as np
import pytest

from conftest import ROOT

FX = os.path.join(ROOT, "tests", "golden", "circom_ev_576_192.npz")


def _sym_from_fixture(sym0, o0):
    order = np.argsort(o0, kind="stable")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    lines = [f"{int(o0[s])},{int(rank[s])},0,{name}\n" for s, name in sym0[1:]]
    return "".join(lines), order


def test_fixture_sym_is_a_valid_layout_for_the_product():
    import zkwg
    fx = np.load(FX)
    c0 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1)
    sym0 = c0.symbols()
    text, order = _sym_from_fixture(sym0, fx["o0_index"])
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1, sym=text)
    assert c.W == c0.W == len(fx["o0_index"])
    got = c.symbols()
    assert [n for _, n in got] == [sym0[int(s)][1] for s in order]
    # the interpreter's creation order differs from kept-v1's array-grouped order (RSAPad, FpMul ...):
    assert [n for _, n in got] != [n for _, n in sym0]
    # documented list of O0 signals the kept-v1 layout drops
    dropped = bytes(fx["dropped"]).decode().splitlines()
    n_dropped = sum(int(l.split()[0]) for l in dropped)
    assert n_dropped == int(fx["n_o0"][0]) - c0.W


@pytest.mark.gpu
def test_hip_witness_in_interpreter_order_equals_the_interpreter():
    import zkwg
    fx = np.load(FX)
    inp = json.loads(bytes(fx["inputs"]).decode())
    c0 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=-1)
    text, _ = _sym_from_fixture(c0.symbols(), fx["o0_index"])
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=0, sym=text)
    wit, status = c.calculate_batch_host(c.pack(inp))
    assert status == [0]
    exp = zlib.decompress(bytes(fx["witness"]))
    assert len(exp) == c.witness_bytes
    if wit != exp:
        bad = [i for i in range(c.W) if wit[32 * i:32 * i + 32] != exp[32 * i:32 * i + 32]]
        raise AssertionError(f"{len(bad)} slots differ, first {bad[:5]}: {c.symbols()[bad[0]]}")
