"""Build tests/golden/circom_ev_576_192.npz from the circom interpreter (oracle/circom) executing the
reference's UNMODIFIED sources (needs /root/reference; run from the repo root):

    python tests/golden/make_circom_fixture.py

Content (self-contained, so the `-m gpu` test needs neither the reference nor the interpreter):
  inputs     JSON of the CircuitInput fed to EmailVerifier(576,192,121,17,0,0,0,0) (synthetic email
             of tests/test_ev_cpu._inputs(576,192,0,index=0,body_len=60))
  o0_index   for every kept-v1 slot s >= 1 (product order, zkwg.Circuit.symbols()) the O0 signal
             index the interpreter assigns to that signal (index 0 = the constant 1)
  witness    zlib( W x 32-byte LE values ) of the interpreter's witness restricted to the signals the
             kept-v1 layout keeps, in INTERPRETER (O0) order -- i.e. what a compiler `.sym` that
             eliminates exactly the dropped signals would index
  n_o0       number of O0 signals (incl. the constant)
  dropped    text: one line `count pattern` per dropped-signal name pattern (indices collapsed)
"""
import collections
import json
import os
import re
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zk-email-verify_amd", "py")]

N, M = 576, 192


def interpreter_table(inp, n=N, m=M, **flags):
    """-> (root, {kept-v1-style name: (o0 index, value)}, n_o0)"""
    from oracle.circom import ev
    from oracle.circom.compare import flat_walk_kept
    p = ev.email_verifier(n, m, **flags)
    root = p.run(inp)
    table = {}
    i = 1
    # kept-v1 (product / pyref) names flatten multi-dimensional signals: a[t][k] -> a[32 t + k]
    for name, v, _, _ in flat_walk_kept(root, p.templates_src):
        table[name] = (i, v)
        i += 1
    return root, table, i


def main():
    import zkwg
    from test_ev_cpu import _inputs
    inp = _inputs(N, M, 0, index=0, body_len=60)
    root, table, n_o0 = interpreter_table(inp)
    c0 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=-1)
    sym = c0.symbols()
    assert sym[0] == (0, "one")
    o0 = np.zeros(len(sym), dtype=np.uint32)
    vals = [None] * len(sym)
    vals[0] = 1
    for slot, name in sym[1:]:
        idx, v = table[name]
        o0[slot] = idx
        vals[slot] = v
    assert len(set(o0.tolist())) == len(sym), "two kept slots map to one O0 signal"
    order = np.argsort(o0, kind="stable")
    wit = b"".join(int(vals[s]).to_bytes(32, "little") for s in order)
    kept_names = {name for _, name in sym}
    dropped = collections.Counter(re.sub(r"\[\d+\]", "[]", nm) for nm in table if nm not in kept_names)
    text = "".join(f"{c} {p}\n" for p, c in sorted(dropped.items()))
    dst = os.path.join(ROOT, "tests", "golden", "circom_ev_576_192.npz")
    np.savez_compressed(dst, inputs=np.frombuffer(json.dumps({k: [str(x) for x in v] if isinstance(v, list) else str(v)
                                                              for k, v in inp.items()}).encode(), dtype=np.uint8),
                        o0_index=o0, witness=np.frombuffer(zlib.compress(wit, 9), dtype=np.uint8),
                        n_o0=np.array([n_o0], dtype=np.uint64),
                        dropped=np.frombuffer(text.encode(), dtype=np.uint8))
    print(f"wrote {dst}: W_kept={len(sym)}, n_o0={n_o0}, dropped patterns={len(dropped)}, "
          f"{os.path.getsize(dst)} bytes")


if __name__ == "__main__":
    main()
