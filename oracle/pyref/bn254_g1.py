"""TEST INFRASTRUCTURE (oracle): BN254 (alt_bn128) G1 and the multi-exponentiations of `snarkjs.groth16.prove` -- the step that
follows the H evaluations of oracle/pyref/ntt.py ("MSM is the next row").

Reference call site: packages/helpers/src/chunked-zkey.ts:80-84 (`snarkjs.groth16.fullProve`).  The arithmetic lives in the
third-party packages snarkjs (pinned: sampritipanda/snarkjs#fef81fc5 = 0.5.0 for helpers, 0.7.5 for circuits; yarn.lock:7767-
7800) and ffjavascript (0.2.x / 0.3.1), neither of which is in /root/reference.  Restated from their published algorithm
[EXT]:

  snarkjs src/groth16_prove.js
      proof.pi_a = curve.G1.multiExpAffine(buffBasesA,  buffWitness)           # section 5 of the zkey, all nVars wires
      pib1       = curve.G1.multiExpAffine(buffBasesB1, buffWitness)           # section 6
      proof.pi_b = curve.G2.multiExpAffine(buffBasesB2, buffWitness)           # section 7 (G2: not restated here yet)
      proof.pi_c = curve.G1.multiExpAffine(buffBasesC,  buffWitness[nPublic+1:])   # section 8, private wires only
      resH       = curve.G1.multiExpAffine(buffBasesH,  buffPodd_T)            # section 9, the H evaluations (domainSize points)
      pi_a = pi_a + vk_alpha_1 + r * vk_delta_1;   pi_b = pi_b + vk_beta_2 + s * vk_delta_2;   pib1 = pib1 + vk_beta_1 + s * vk_delta_1
      pi_c = pi_c + resH + s * pi_a + r * pib1 - (r * s) * vk_delta_1
  ffjavascript: scalars enter multiExpAffine in standard form (the witness is converted with Fr.fromMontgomery / batchFromMontgomery
  first); the bases are affine points in Montgomery form, 64 bytes each (x | y little-endian limbs), the point at infinity is all zeros.

Curve: y^2 = x^3 + 3 over Fq, q = 21888242871839275222246405745257275088696311157297823662689037894645226208583, generator (1, 2), group
order r = the scalar field of the circuit.  Pinned by: the generator is on the curve, r * G = O, and 2 G / 3 G equal the alt_bn128
vectors of the Ethereum precompile tests (EIP-196), which the reference's own verifier contracts run against
(packages/contracts -- the pairing precompile).  "parity unpinned" for the MSM stage as a whole: no zkey / proof of the real
snarkjs exists offline; the definitions above are pinned by group laws and by agreement of three independent evaluations (naive
double-and-add, this bucket method, the product's).  Pure Python integers.
"""
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
B = 3
G = (1, 2)
O = None   # the point at infinity

# EIP-196 (alt_bn128 addition / multiplication precompiles) known answers
G2X = 0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3
G2Y = 0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4
G3X = 0x0769bf9ac56bea3ff40232bcb1b6bd159315d84715b8e679f2d355961915abf0
G3Y = 0x2ab799bee0489429554fdb7c8d086475319e63b40b9c5b57cdf1ff3dd9fe2261


def on_curve(p):
    if p is O:
        return True
    x, y = p
    return (y * y - x * x * x - B) % Q == 0


def neg(p):
    return O if p is O else (p[0], (-p[1]) % Q)


def add(p, q):
    if p is O:
        return q
    if q is O:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return O
        lam = 3 * x1 * x1 * pow(2 * y1, Q - 2, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, Q - 2, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


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


def msm_buckets(points, scalars, c=8):
    """the bucket method every fast implementation uses (ffjavascript's multiExpAffine included): windows of c bits, per window one
    bucket per digit value, sum_d d * bucket[d] by a running sum from the top, windows combined by c doublings each"""
    nwin = (254 + c - 1) // c
    total = O
    for w in reversed(range(nwin)):
        for _ in range(c):
            total = add(total, total)
        buckets = [O] * (1 << c)
        for p, k in zip(points, scalars):
            d = ((k % R) >> (w * c)) & ((1 << c) - 1)
            if d:
                buckets[d] = add(buckets[d], p)
        run, acc = O, O
        for d in range((1 << c) - 1, 0, -1):
            run = add(run, buckets[d])
            acc = add(acc, run)
        total = add(total, acc)
    return total


def random_points(n, seed):
    """n points k_i * G with pseudo-random k_i (test bases: the zkey's are absent)"""
    import random
    rng = random.Random(seed)
    return [mul(rng.randrange(1, R), G) for _ in range(n)]
