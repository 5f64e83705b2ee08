"""TEST INFRASTRUCTURE (oracle): the step of `snarkjs.groth16.prove` that follows A.w | B.w | C.w -- the evaluations of
a(x) b(x) - c(x) on the odd coset of the doubled domain, which the prover then feeds to the H multi-exponentiation.

Reference call site: packages/helpers/src/chunked-zkey.ts:80-84 (`snarkjs.groth16.fullProve`).  The arithmetic lives in the
third-party packages snarkjs (pinned: sampritipanda/snarkjs#fef81fc5 = 0.5.0 for helpers, 0.7.5 for circuits; yarn.lock:7767-
7800) and ffjavascript (0.2.x / 0.3.1), neither of which is in /root/reference.  Restated from their published algorithm
[EXT]:

  snarkjs src/groth16_prove.js
      [A_T, B_T, C_T] = buildABC1(...)                      # A.w, B.w, C.w per constraint, length domainSize, zero padded
      inc   = (power == Fr.s) ? Fr.shift : Fr.w[power + 1]  # the primitive 2 m-th root for m = 2^power
      A     = Fr.ifft(A_T);  Aodd = batchApplyKey(A, 1, inc) # coefficient i times inc^i
      Aodd_T = Fr.fft(Aodd)                                 # = a(inc * w^k), k = 0 .. m-1   (same for B, C)
      P_T[k] = Aodd_T[k] * Bodd_T[k] - Codd_T[k]            # joinABC
  ffjavascript F1Field: s = 2-adicity of r - 1, t = (r - 1) / 2^s, nqr = smallest quadratic non-residue (5 for BN254 Fr),
      w[s] = nqr^t, w[i] = w[i+1]^2, shift = nqr^2;  fft: X[k] = sum_j x[j] w^{jk}, ifft its inverse (natural order both).

"parity unpinned" for this stage: no vector of the real snarkjs exists offline; the definitions above are pinned only by
self-consistency (ifft . fft = id, the polynomial identity below).  Pure Python integers; O(n^2) reference for small domains,
Horner spot checks for the full size.
"""
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
S = 28                                   # r - 1 = 2^28 * t
T = (P - 1) >> S
NQR = 5
assert pow(NQR, (P - 1) // 2, P) == P - 1 and all(pow(g, (P - 1) // 2, P) == 1 for g in (2, 3, 4))
W = [0] * (S + 1)
W[S] = pow(NQR, T, P)
for _i in range(S - 1, -1, -1):
    W[_i] = W[_i + 1] * W[_i + 1] % P
assert W[0] == 1 and W[1] == P - 1
SHIFT = NQR * NQR % P


def root(power):
    """primitive 2^power-th root of unity used by Fr.fft for a domain of that size"""
    return W[power]


def coset_inc(power):
    return SHIFT if power == S else W[power + 1]


def dft(xs, w):
    n = len(xs)
    return [sum(x * pow(w, j * k, P) for j, x in enumerate(xs)) % P for k in range(n)]


def fft(xs):
    power = len(xs).bit_length() - 1
    assert 1 << power == len(xs)
    return dft(xs, root(power))


def ifft(xs):
    power = len(xs).bit_length() - 1
    ninv = pow(len(xs), P - 2, P)
    return [v * ninv % P for v in dft(xs, pow(root(power), P - 2, P))]


def h_evaluations(a, b, c, power):
    """O(n^2): P_T of groth16_prove.js for A.w = a, B.w = b, C.w = c (each zero-padded to 2^power)."""
    n = 1 << power
    inc = coset_inc(power)
    out = []
    odd = []
    for v in (a, b, c):
        co = ifft(list(v) + [0] * (n - len(v)))
        odd.append(fft([x * pow(inc, i, P) % P for i, x in enumerate(co)]))
    return [(odd[0][k] * odd[1][k] - odd[2][k]) % P for k in range(n)]


def coset_eval_direct(vals, power, k):
    """value at x_k = inc * w^k of the polynomial of degree < 2^power that interpolates `vals` (zero-padded) on the domain
    {w^j}: barycentric form  p(x) = (x^n - 1) / n * sum_j vals[j] w^j / (x - w^j)  -- O(n) per point, for spot checks at full size"""
    n = 1 << power
    w = root(power)
    x = coset_inc(power) * pow(w, k, P) % P
    zn = (pow(x, n, P) - 1) * pow(n, P - 2, P) % P
    acc = 0
    wj = 1
    # batch the inversions: sum_j v_j w^j / (x - w^j)
    dens, nums = [], []
    for j in range(n):
        if j < len(vals) and vals[j]:
            dens.append((x - wj) % P)
            nums.append(vals[j] * wj % P)
        wj = wj * w % P
    if not dens:
        return 0
    pref = [1]
    for d in dens:
        pref.append(pref[-1] * d % P)
    inv = pow(pref[-1], P - 2, P)
    for i in range(len(dens) - 1, -1, -1):
        acc = (acc + nums[i] * inv % P * pref[i]) % P
        inv = inv * dens[i] % P
    return zn * acc % P
