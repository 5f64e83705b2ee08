"""circom-2 subset interpreter (TEST INFRASTRUCTURE -- part of the oracle, never of the product path).

Purpose: execute the reference's own `.circom` sources *unmodified* from
/root/reference/packages/circuits (email-verifier.circom, lib/*.circom, utils/*.circom,
helpers/*.circom, tests/test-circuits/*.circom), so that witness values, signal names and the O0
signal order come from the reference text itself and not from a hand restatement.

What is NOT the reference's: the circom compiler (absent offline -- this package restates the
language semantics of circom 2.1.x from its documentation), circomlib
2.0.5 (absent; its templates are restated *in circom syntax* under `oracle/circom/lib/circomlib`,
SURVEY.md Appendix A.1/A.2/A.4) and zk-regex's generated `body_hash_regex.circom` (absent; a circom
rendering of zkwg's own DFA circuit stands in until the real file is supplied, see
`oracle/circom/lib/@zk-email/zk-regex-circom`).

Only tests/, tools that build fixtures, and `__graft_entry__.build()` (which writes `oracle/_ref/`)
may import this package.
"""
from .runtime import Program, AssertFailed, CircomError, P  # noqa: F401
