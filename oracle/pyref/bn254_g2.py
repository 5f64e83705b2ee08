"""TEST INFRASTRUCTURE (oracle): BN254 G2 -- the group of `proof.pi_b = curve.G2.multiExpAffine(buffBasesB2, buffWitness)` in
snarkjs's groth16_prove.js [EXT] (see oracle/pyref/bn254_g1.py for the call sites and the pinning of the dependency).

G2 = the order-r subgroup of the sextic twist  y^2 = x^3 + 3 / (9 + i)  over Fq2 = Fq[i] / (i^2 + 1); an element of Fq2 is a pair
(c0, c1) = c0 + c1 i, and ffjavascript / the zkey store a G2 point as x.c0 | x.c1 | y.c0 | y.c1 (32-byte little-endian limbs each,
Montgomery form, 128 bytes, all zeros = infinity).  Pinned by: the generator below is the alt_bn128 G2 generator of EIP-197 (the
pairing precompile the reference's verifier contracts call), it satisfies the twist equation and r * G2 = O.  Pure Python integers.
"""
from oracle.pyref.bn254_g1 import Q, R

O = None


def f2(a, b=0):
    return (a % Q, b % Q)


def f2_add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def f2_sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def f2_neg(a):
    return ((-a[0]) % Q, (-a[1]) % Q)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * d % Q, (-a[1]) * d % Q)


B2 = f2_mul(f2(3), f2_inv(f2(9, 1)))
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))


def on_curve(p):
    if p is O:
        return True
    x, y = p
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), B2)


def neg(p):
    return O if p is O else (p[0], f2_neg(p[1]))


def add(p, q):
    if p is O:
        return q
    if q is O:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return O
        lam = f2_mul(f2_mul(f2(3), f2_mul(x1, x1)), f2_inv(f2_add(y1, y1)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def mul(k, p):
    k %= R
    acc = O
    while k:
        if k & 1:
            acc = add(acc, p)
        p = add(p, p)
        k >>= 1
    return acc


def msm_naive(points, scalars):
    acc = O
    for p, k in zip(points, scalars):
        acc = add(acc, mul(k, p))
    return acc


def random_points(n, seed):
    import random
    rng = random.Random(seed)
    return [mul(rng.randrange(1, R), G2) for _ in range(n)]
