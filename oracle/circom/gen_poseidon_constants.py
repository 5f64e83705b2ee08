"""Generate circomlib's `poseidon_constants.circom` (absent offline) -- TEST INFRASTRUCTURE.

circomlib's optimised Poseidon (`poseidon.circom`) reads four tables per
width t: POSEIDON_C (round constants, shifted), POSEIDON_S (sparse partial-round matrices),
POSEIDON_M (MDS) and POSEIDON_P (the dense matrix in front of the partial rounds).  They are derived
here from the textbook constants of oracle/pyref/poseidon.py (Grain LFSR + Cauchy MDS, pinned by
circomlibjs' published hash vectors) by the optimisation of the Poseidon reference implementation
(hadeshash `calc_equivalent_constants` / `calc_equivalent_matrices`), in the variant circomlib's
table sizes imply (t*R_F + R_P constants: every partial round and the first full round after them get
a scalar only):

  textbook round r:   x <- M * Sbox_r(x + c_r)
  1. k_0 = c_0, k_r = M^-1 c_r (r >= 1)  -- added after the S-box of round r-1, before the matrix;
  2. for r = R_f+R_P .. R_f+1 (descending): keep k_r[0]; push (0, k_r[1:]) through the partial S-box
     and the matrix of round r-1:  k_{r-1} += M^-1 (0, k_r[1:]);
  3. matrices, from the last partial round back to the first: A = M''(sparse) * M'(=diag(1, A^))
     with M'' = [[a00, v^T A^^-1], [w, I]];  M' commutes with the partial S-box and merges into the
     previous round's matrix;  the leftover of the first partial round gives P = M'_0 * M.

The S-box inputs (hence Sigma.in2 / in4 / out, the only non-linear signals) equal the textbook's;
tests compare the interpreter's hash and S-box values with oracle/pyref/poseidon.py.  The circom
tables store matrices transposed (Mix does `lc += M[j][i]*in[j]`).  [EXT, unverifiable offline]:
whether circomlib's published tables use exactly these (equivalent) representatives.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.pyref import poseidon as _pos  # noqa: E402

P = _pos.P
R_F = 8


def _inv(x):
    return pow(x % P, P - 2, P)


def mat_inv(A):
    n = len(A)
    M = [list(r) + [1 if i == j else 0 for j in range(n)] for i, r in enumerate(A)]
    for c in range(n):
        p = next(r for r in range(c, n) if M[r][c] % P)
        M[c], M[p] = M[p], M[c]
        iv = _inv(M[c][c])
        M[c] = [x * iv % P for x in M[c]]
        for r in range(n):
            if r != c and M[r][c]:
                f = M[r][c]
                M[r] = [(x - f * y) % P for x, y in zip(M[r], M[c])]
    return [r[n:] for r in M]


def mat_mul(A, B):
    n, m, k = len(A), len(B[0]), len(B)
    return [[sum(A[i][l] * B[l][j] for l in range(k)) % P for j in range(m)] for i in range(n)]


def mat_vec(A, v):
    return [sum(a * x for a, x in zip(row, v)) % P for row in A]


def optimized(t):
    """-> (C, S, Mt, Pt) as poseidon.circom indexes them."""
    r_p = _pos.N_ROUNDS_P[t - 2]
    R = R_F + r_p
    r_f = R_F // 2
    c, M = _pos.constants(t)
    Minv = mat_inv(M)
    k = [c[0:t]] + [mat_vec(Minv, c[r * t:(r + 1) * t]) for r in range(1, R)]
    for r in range(r_f + r_p, r_f, -1):
        v = k[r]
        push = mat_vec(Minv, [0] + v[1:])
        k[r - 1] = [(a + b) % P for a, b in zip(k[r - 1], push)]
        k[r] = [v[0]] + [0] * (t - 1)
    C = []
    for r in range(0, r_f + 1):
        C += k[r]
    for r in range(r_f + 1, r_f + r_p + 1):
        C.append(k[r][0])
    for r in range(r_f + r_p + 1, R):
        C += k[r]
    assert len(C) == t * R_F + r_p
    # sparse factorisation, last partial round first
    S = [None] * r_p
    A = M
    for p in range(r_p - 1, -1, -1):
        a00 = A[0][0]
        v = A[0][1:]
        w = [A[i][0] for i in range(1, t)]
        Ah = [row[1:] for row in A[1:]]
        Ahinv = mat_inv(Ah)
        vhat = [sum(v[i] * Ahinv[i][j] for i in range(t - 1)) % P for j in range(t - 1)]
        S[p] = [a00] + vhat + w
        Mp = [[1] + [0] * (t - 1)] + [[0] + row for row in Ah]
        A = mat_mul(Mp, M)
    Pm = A
    Sflat = [x for s in S for x in s]
    Mt = [[M[j][i] for j in range(t)] for i in range(t)]
    Pt = [[Pm[j][i] for j in range(t)] for i in range(t)]
    return C, Sflat, Mt, Pt


def permute_optimized(t, inputs):
    """Python mirror of poseidon.circom's PoseidonEx data flow; returns (hash, sbox_inputs)."""
    C, S, Mt, Pt = optimized(t)
    r_p = _pos.N_ROUNDS_P[t - 2]
    sbox_in = []

    def mix(Mx, s):
        return [sum(Mx[j][i] * s[j] for j in range(t)) % P for i in range(t)]

    def sig(x):
        sbox_in.append(x)
        return pow(x, 5, P)
    s = [0] + [x % P for x in inputs]
    s = [(s[i] + C[i]) % P for i in range(t)]
    for r in range(3):
        s = [sig(x) for x in s]
        s = [(s[i] + C[(r + 1) * t + i]) % P for i in range(t)]
        s = mix(Mt, s)
    s = [sig(x) for x in s]
    s = [(s[i] + C[4 * t + i]) % P for i in range(t)]
    s = mix(Pt, s)
    for r in range(r_p):
        s[0] = (sig(s[0]) + C[5 * t + r]) % P
        base = (2 * t - 1) * r
        n0 = sum(S[base + i] * s[i] for i in range(t)) % P
        s = [n0] + [(s[i] + s[0] * S[base + t + i - 1]) % P for i in range(1, t)]
    for r in range(3):
        s = [sig(x) for x in s]
        s = [(s[i] + C[5 * t + r_p + r * t + i]) % P for i in range(t)]
        s = mix(Mt, s)
    s = [sig(x) for x in s]
    return sum(Mt[j][0] * s[j] for j in range(t)) % P, sbox_in


def render(ts):
    out = ["// GENERATED by oracle/circom/gen_poseidon_constants.py -- stands in for circomlib's",
           "// poseidon_constants.circom (absent offline).  Widths: t in %s." % list(ts),
           "pragma circom 2.0.0;", ""]
    tabs = {t: optimized(t) for t in ts}

    def fn(name, idx, fmt):
        out.append(f"function {name}(t) {{")
        first = True
        for t in ts:
            kw = "if" if first else "} else if"
            first = False
            out.append(f"    {kw} (t=={t}) {{")
            out.append("        return " + fmt(tabs[t][idx]) + ";")
        out.append("    } else {")
        out.append("        assert(0);")
        out.append("        return [0];")
        out.append("    }")
        out.append("}")
        out.append("")

    def vec(v):
        return "[" + ",".join(str(x) for x in v) + "]"

    def mat(m):
        return "[" + ",".join(vec(r) for r in m) + "]"
    fn("POSEIDON_C", 0, vec)
    fn("POSEIDON_S", 1, vec)
    fn("POSEIDON_M", 2, mat)
    fn("POSEIDON_P", 3, mat)
    return "\n".join(out)


def main(argv):
    ts = [3, 6, 10, 17]
    if "--all" in argv:
        ts = list(range(2, 18))
    dst = os.path.join(ROOT, "oracle", "_ref", "poseidon_constants.circom")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write(render(ts))
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1:])
