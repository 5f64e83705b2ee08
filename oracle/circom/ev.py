"""Convenience layer over the interpreter for the reference's circuits (test infrastructure).

`program(main_file)` loads a reference test circuit with the include path the reference's own test
harness passes to circom (`include: node_modules`, packages/circuits/tests/*.test.ts `wasm_tester`
options): circomlib -> oracle/circom/lib/circomlib (restated), zk-regex -> $ZKWG_ZK_REGEX_DIR if set
(a real `@zk-email/zk-regex-circom` checkout) else the stand-in under oracle/circom/lib, generated
Poseidon constants -> oracle/_ref/.
"""
import os

from .runtime import Program

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_CIRCUITS = "/root/reference/packages/circuits"
LIB = os.path.join(HERE, "lib")
GEN = os.path.join(ROOT, "oracle", "_ref")


def reference_available():
    return os.path.isfile(os.path.join(REF_CIRCUITS, "email-verifier.circom"))


def ensure_generated():
    dst = os.path.join(GEN, "poseidon_constants.circom")
    if not os.path.isfile(dst):
        from . import gen_poseidon_constants
        gen_poseidon_constants.main([])
    return dst


def include_paths():
    ensure_generated()
    paths = []
    real = os.environ.get("ZKWG_ZK_REGEX_DIR")
    if real:
        # a node_modules-style root that contains @zk-email/zk-regex-circom/circuits/...
        paths.append(real)
    paths += [LIB, GEN]
    return paths


def program(main_file, **kw):
    if not os.path.isabs(main_file):
        main_file = os.path.join(REF_CIRCUITS, main_file)
    return Program(main_file, include_paths(), **kw)


def email_verifier(max_header, max_body, n=121, k=17, ignore_body_hash_check=0, header_mask=0,
                   body_mask=0, remove_soft_line_breaks=0, **kw):
    """Program whose main is EmailVerifier(...) with `public [pubkey]`, exactly like
    tests/test-circuits/email-verifier-test.circom but with free parameters."""
    p = Program(None, include_paths(), **kw)
    p.load(os.path.join(REF_CIRCUITS, "email-verifier.circom"))
    p.main = (["pubkey"], "EmailVerifier",
              [("num", v) for v in (max_header, max_body, n, k, ignore_body_hash_check, header_mask,
                                    body_mask, remove_soft_line_breaks)])
    return p
