#!/usr/bin/env python3
"""Real-artefact intake, one command (TEST INFRASTRUCTURE -- hence under tests/: it runs the circom interpreter of oracle/circom as the
checker; `tools/intake.py` is a launcher for this file).

The rows of SURVEY.md section 8 that stay "parity unpinned" are blocked on files, not on code: the circom compiler's
`.sym` / `.r1cs` of EmailVerifier, zk-regex's generated `body_hash_regex.circom` (+ circomlib) under `node_modules`,
and a snarkjs `.wtns` for one `input.json` (the reference's compile line: docs/zk-email-docs/UsageGuide/README.md:59;
`snarkjs wtns calculate`: :132-140; `loadSymbols`: packages/circuits/tests/email-verifier.test.ts:204-206).  The day they
exist:

    python tools/intake.py --node-modules NM --build-dir BUILD --input input.json [--wtns witness.wtns]
                           [--max-header 1024 --max-body 1536] [--main-kind ev|rsa] [--device 0|-1]

does, in order, and reports the FIRST difference of every comparison by signal name:
  1. interpreter: executes the reference's unmodified circuit with NM first on the include path (real circomlib, real
     zk-regex) -> every signal by compiler-style name;
  2. `.sym`: the interpreter's names and numbering against the file's (names only in one of them, first index mismatch);
  3. `.wtns` (if given): the file's value of every `.sym` signal against the interpreter's;
  4. product: handle from the regex template + `.sym` + `.r1cs` (zkwg_circuit_create_regex / _create_full); a circom
     construct outside the loader's subset is reported with its file:line (zkwg_last_error);
  5. product witness on the device against the interpreter and the `.wtns`; `checkConstraints` of the product witness
     against the `.r1cs` on the device.
--device -1 stops after step 4 with a layout-only handle (no GPU).  Exit code 0 = everything that could be compared agrees.

`--build-dir` may also hold the artefacts this repo generates offline (`o0_<tag>.sym.gz`, `.r1cs.gz`, `.json`:
oracle/circom/o0_artifacts.py); tests/test_intake.py runs the tool on those."""
import argparse
import glob
import gzip
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "zk-email-verify_amd", "py"), os.path.join(ROOT, "tests")]

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _read(path):
    return gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()


def find_artefacts(build_dir, name="*"):
    """-> (sym path, r1cs path): the first <name>.sym[.gz] / <name>.r1cs[.gz] of the directory (or of its first level)."""
    out = []
    for ext in ("sym", "r1cs"):
        hits = sorted(glob.glob(os.path.join(build_dir, name + "." + ext)) + glob.glob(os.path.join(build_dir, name + "." + ext + ".gz")) +
                      glob.glob(os.path.join(build_dir, "*", name + "." + ext)))
        if not hits:
            raise SystemExit(f"intake: no .{ext} file under {build_dir}")
        out.append(hits[0])
    return out


def parse_sym(text):
    """`labelIdx,witnessIdx,componentIdx,name` -> {name: witnessIdx} for the signals the compiler kept, + the count of eliminated ones."""
    kept, dropped = {}, 0
    for line in text.splitlines():
        if not line:
            continue
        a, b, c, name = line.split(",", 3)
        if int(b) < 0:
            dropped += 1
        else:
            kept[name] = int(b)
    return kept, dropped


def parse_wtns(data):
    """snarkjs `.wtns` -> list of ints."""
    if data[:4] != b"wtns":
        raise SystemExit("intake: not a .wtns file")
    nsec = struct.unpack_from("<I", data, 8)[0]
    pos, n8, nw, body = 12, None, None, None
    for _ in range(nsec):
        sid, size = struct.unpack_from("<IQ", data, pos)
        pos += 12
        if sid == 1:
            n8 = struct.unpack_from("<I", data, pos)[0]
            nw = struct.unpack_from("<I", data, pos + 4 + n8)[0]
        elif sid == 2:
            body = data[pos:pos + size]
        pos += size
    if n8 != 32 or body is None or len(body) != 32 * nw:
        raise SystemExit("intake: malformed .wtns")
    return [int.from_bytes(body[32 * i:32 * i + 32], "little") for i in range(nw)]


def run(args, log=print):
    rep = {"ok": True, "steps": {}}

    def step(name, ok, **kw):
        rep["steps"][name] = dict(ok=bool(ok), **kw)
        if not ok:
            rep["ok"] = False
        log(f"[intake] {name}: {'ok' if ok else 'DIFFERENT'} " + " ".join(f"{k}={v}" for k, v in kw.items()))

    if args.node_modules:
        os.environ["ZKWG_ZK_REGEX_DIR"] = args.node_modules       # first on the interpreter's include path (oracle/circom/ev.py)
    from oracle.circom import ev
    from oracle.circom.o0_artifacts import alias_rules
    from oracle.circom.runtime import iter_signals
    inputs = json.load(open(args.input))
    inputs = inputs.get("inputs", inputs) if isinstance(inputs, dict) else inputs
    sym_path, r1cs_path = find_artefacts(args.build_dir, args.name)
    sym_text = _read(sym_path).decode()
    r1cs = _read(r1cs_path)

    # 1. the interpreter on the reference's own sources (or its saved result: --interpreter-dump, for boxes without /root/reference)
    if args.interpreter_dump:
        import numpy as np
        z = np.load(args.interpreter_dump)
        names = bytes(z["names"]).decode().split("\n")
        raw = bytes(z["values"])
        values = [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(len(raw) // 32)]
        alias = bytes(z["alias"]).decode()
        step("interpreter (saved run)", len(names) == len(values), signals=len(names))
    else:
        if args.main_kind == "rsa":
            prog = ev.program("tests/test-circuits/rsa-test.circom")
        else:
            prog = ev.email_verifier(args.max_header, args.max_body)
        try:
            root = prog.run(inputs)
        except Exception as e:     # AssertFailed, or a construct of the real files the interpreter does not cover
            step("interpreter", False, error=repr(e)[:300])
            return rep
        names, values = ["one"], [1]
        for nm, v, _, _ in iter_signals(root):
            names.append(nm)
            values.append(v)
        alias = "".join(f"{a}={b}\n" for a, b in alias_rules(root, prog.templates_src))
        step("interpreter", True, signals=len(names))
        if args.dump_interpreter:
            import numpy as np
            np.savez_compressed(args.dump_interpreter, names=np.frombuffer("\n".join(names).encode(), dtype=np.uint8),
                                values=np.frombuffer(b"".join(v.to_bytes(32, "little") for v in values), dtype=np.uint8),
                                alias=np.frombuffer(alias.encode(), dtype=np.uint8))
    by_name = dict(zip(names, values))

    # 2. names and numbering against the compiler's .sym
    kept, dropped = parse_sym(sym_text)
    only_sym = [n for n in kept if n not in by_name]
    only_int = [n for n in names[1:] if n not in kept] if not dropped else []
    order_bad = next((n for i, n in enumerate(names) if i and kept.get(n, i) != i), None) if not dropped else None
    step(".sym names", not only_sym and not only_int, sym_signals=len(kept), eliminated=dropped,
         first_only_in_sym=only_sym[0] if only_sym else None, first_only_in_interpreter=only_int[0] if only_int else None)
    if not dropped:
        step(".sym numbering (interpreter's restated compiler order)", order_bad is None,
             first_mismatch=None if order_bad is None else f"{order_bad}: interpreter {names.index(order_bad)} file {kept[order_bad]}")
    by_idx = sorted((i, n) for n, i in kept.items())

    # 3. the snarkjs witness against the interpreter
    wt = None
    if args.wtns:
        wt = parse_wtns(_read(args.wtns))
        bad = next(((i, n) for i, n in by_idx if n in by_name and i < len(wt) and wt[i] != by_name[n]), None)
        step(".wtns vs interpreter", bad is None and len(wt) == max(kept.values()) + 1, wtns_len=len(wt),
             first_difference=None if bad is None else f"{bad[1]} (index {bad[0]}): file {wt[bad[0]]} interpreter {by_name[bad[1]]}")

    # 4. the product handle from the same artefacts
    import zkwg
    kw = dict(sym=sym_text, sym_alias=alias, r1cs=r1cs, device=args.device)
    if args.main_kind == "rsa":
        mk, N, M = zkwg.MAIN_RSA_VERIFIER, 0, 0
    else:
        mk, N, M = zkwg.MAIN_EMAIL_VERIFIER, args.max_header, args.max_body
        tmpl = args.regex_template
        if tmpl is None and args.node_modules:
            cand = os.path.join(args.node_modules, "@zk-email", "zk-regex-circom", "circuits", "common", "body_hash_regex.circom")
            tmpl = cand if os.path.exists(cand) else None
        if tmpl:
            kw.update(regex=tmpl, regex_include_dirs=[d for d in (args.node_modules,) if d])
    try:
        c = zkwg.Circuit(mk, max_header=N, max_body=M, **kw)
    except zkwg.ZkwgError as e:
        step("product handle", False, error=str(e)[:400])
        return rep
    step("product handle", c.W == max(kept.values()) + 1, witness_len=c.W, regex_template=kw.get("regex"), linear_rows=int(c.lib.zkwg_linear_rows(c.h)))
    if args.device < 0:
        log("[intake] --device -1: no device witness")
        return rep

    # 5. the product witness on the device
    wit, status = c.calculate_batch_host(c.pack(inputs))
    w = zkwg.witness_ints(wit)
    bad = next(((i, n) for i, n in by_idx if n in by_name and w[i] != by_name[n]), None)
    step("product vs interpreter", status == [0] and bad is None, status=status[0],
         first_difference=None if bad is None else f"{bad[1]} (index {bad[0]}): product {w[bad[0]]} interpreter {by_name[bad[1]]}")
    if wt is not None:
        bad = next((i for i in range(min(len(w), len(wt))) if w[i] != wt[i]), None)
        nm = None if bad is None else next((n for i, n in by_idx if i == bad), "?")
        step("product vs .wtns", bad is None and len(w) == len(wt),
             first_difference=None if bad is None else f"{nm} (index {bad}): product {w[bad]} file {wt[bad]}")
    R = zkwg.R1cs(r1cs, device=args.device)
    fv = R.first_violations(wit, 1)[0]
    step("checkConstraints (product witness, device)", fv is None, constraints=R.n_constraints, first_violated=fv)
    return rep


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--node-modules", default=None, help="node_modules root holding circomlib/ and @zk-email/zk-regex-circom/ (default: the restated copies)")
    ap.add_argument("--build-dir", required=True, help="directory with the compiler's .sym and .r1cs (plain or .gz)")
    ap.add_argument("--name", default="*", help="base name of the artefacts inside --build-dir when it holds several circuits (glob)")
    ap.add_argument("--input", required=True, help="input.json (the CircuitInput of generateEmailVerifierInputs), or an o0_*.json of this repo")
    ap.add_argument("--wtns", default=None, help="snarkjs witness of the same input")
    ap.add_argument("--main-kind", default="ev", choices=["ev", "rsa"])
    ap.add_argument("--max-header", type=int, default=1024)
    ap.add_argument("--max-body", type=int, default=1536)
    ap.add_argument("--regex-template", default=None, help="body_hash_regex.circom (default: the one under --node-modules)")
    ap.add_argument("--device", type=int, default=0, help="GPU index, -1 = stop after building a layout-only handle")
    ap.add_argument("--dump-interpreter", default=None, help="save the interpreter's run (names, values, rename rules) as .npz")
    ap.add_argument("--interpreter-dump", default=None, help="use a saved interpreter run instead of executing the circuit (a box without /root/reference)")
    ap.add_argument("--json", default=None, help="write the report here")
    args = ap.parse_args(argv)
    rep = run(args)
    if args.json:
        json.dump(rep, open(args.json, "w"), indent=1)
    print("[intake] RESULT:", "all comparisons agree" if rep["ok"] else "DIFFERENCES (see above)")
    return 0 if rep["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
