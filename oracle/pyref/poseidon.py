"""Poseidon (circomlib 2.0.5 `poseidon.circom`, x^5 S-box over BN254 Fr) restated.

[EXT] circomlib's source and its `poseidon_constants.circom` are absent from the build
container.  The constants are regenerated here from the published
procedure: round constants from the Grain LFSR of the Poseidon reference implementation
(field = prime, sbox = x^alpha, n = 254, t, R_F = 8, R_P) by rejection sampling, and the MDS
matrix as the Cauchy matrix M[i][j] = 1 / (x_i + y_j) with x, y the next 2t values of the
same Grain stream.  They are
pinned by the globally known vector poseidon([1, 2]) (tests/test_oracle_kats.py).

circomlib evaluates an *optimised* form (sparse partial-round matrices, shifted round
constants).  That form is functionally identical to the textbook permutation and applies
the S-box to the same values, so the quadratic (kept) signals -- Sigma.in2, Sigma.in4,
Sigma.out of every S-box -- are computed here from the textbook rounds; the linear Ark/Mix
signals of the optimised form are not part of the kept layout.
"""
from .comp import Comp, P

N_ROUNDS_P = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]
N_ROUNDS_F = 8
_cache = {}


def _grain(t, r_f, r_p, n=254):
    bits = []

    def put(v, w):
        bits.extend(int(b) for b in bin(v)[2:].zfill(w))

    put(1, 2); put(0, 4); put(n, 12); put(t, 12); put(r_f, 10); put(r_p, 10)
    bits.extend([1] * 30)
    state = bits

    def step():
        nb = state[62] ^ state[51] ^ state[38] ^ state[23] ^ state[13] ^ state[0]
        state.pop(0)
        state.append(nb)
        return nb

    for _ in range(160):
        step()
    while True:
        nb = step()
        while nb == 0:
            step()
            nb = step()
        yield step()


def constants(t):
    """(C, M): C = (R_F + R_P) * t round constants, M = t x t MDS matrix."""
    if t in _cache:
        return _cache[t]
    r_p = N_ROUNDS_P[t - 2]
    gen = _grain(t, N_ROUNDS_F, r_p)
    C = []
    while len(C) < (N_ROUNDS_F + r_p) * t:
        v = 0
        for _ in range(254):
            v = (v << 1) | next(gen)
        if v < P:
            C.append(v)
    # MDS: Cauchy matrix 1/(x_i + y_j) with x, y drawn from the same Grain stream (no
    # rejection, reduced mod p), first candidate (create_mds_p of the reference generator).
    rl = []
    for _ in range(2 * t):
        v = 0
        for _ in range(254):
            v = (v << 1) | next(gen)
        rl.append(v % P)
    assert len(set(rl)) == 2 * t
    xs, ys = rl[:t], rl[t:]
    M = [[pow(xs[i] + ys[j], P - 2, P) for j in range(t)] for i in range(t)]
    _cache[t] = (C, M)
    return C, M


def Sigma(x):
    """poseidon.circom Sigma: in2 <== in*in; in4 <== in2*in2; out <== in4*in."""
    c = Comp("Sigma")
    out = c.out("out")
    c.inp("in").set(x, "L")
    in2 = c.mid("in2").set(x * x, "Q")
    in4 = c.mid("in4").set(in2 * in2, "Q")
    c.o = out.set(in4 * x, "Q")
    return c


def Poseidon(nInputs, inputs):
    """poseidon.circom Poseidon(nInputs) -> PoseidonEx(nInputs, 1) with initialState = 0.
    Sub-component order of PoseidonEx: ark[], sigmaF[8][t], sigmaP[R_P], mix[], mixS[], mixLast[]
    (only the Sigma instances carry kept signals)."""
    t = nInputs + 1
    r_p = N_ROUNDS_P[t - 2]
    C, M = constants(t)
    c = Comp(f"Poseidon({nInputs})")
    out = c.out("out")
    c.inp("inputs", nInputs).setall(inputs, "L")
    pEx = Comp(f"PoseidonEx({nInputs},1)")
    pout = pEx.out("out", 1)
    pEx.inp("inputs", nInputs).setall(inputs, "L")
    pEx.inp("initialState").set(0, "L")
    c.sub("pEx", pEx)

    state = [0] + [x % P for x in inputs]
    sigmaF = [[None] * t for _ in range(N_ROUNDS_F)]
    sigmaP = [None] * r_p
    rc = 0

    def mix(s):
        return [sum(M[i][j] * s[j] for j in range(t)) % P for i in range(t)]

    fr = 0
    for r in range(N_ROUNDS_F + r_p):
        state = [(state[i] + C[rc + i]) % P for i in range(t)]
        rc += t
        full = r < N_ROUNDS_F // 2 or r >= N_ROUNDS_F // 2 + r_p
        if full:
            for j in range(t):
                sigmaF[fr][j] = Sigma(state[j])
                state[j] = sigmaF[fr][j].o
            fr += 1
        else:
            k = r - N_ROUNDS_F // 2
            sigmaP[k] = Sigma(state[0])
            state[0] = sigmaP[k].o
        state = mix(state)
    for r in range(N_ROUNDS_F):
        for j in range(t):
            pEx.sub(f"sigmaF[{r}][{j}]", sigmaF[r][j])
    for r in range(r_p):
        pEx.sub(f"sigmaP[{r}]", sigmaP[r])
    pout.set(state[0], "L", 0)
    c.o = out.set(state[0], "L")
    return c


def poseidon_hash(inputs):
    return Poseidon(len(inputs), inputs).o
