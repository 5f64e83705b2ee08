"""TEST INFRASTRUCTURE (oracle): a Groth16 set-up with a KNOWN trapdoor and the prover's arithmetic in Python integers -- what the
device-side prover stages (A.w | B.w | C.w -> H evaluations -> multi-exponentiations -> pi_a, pi_b, pi_c) are checked against.

Reference call site: packages/helpers/src/chunked-zkey.ts:80-84 (`snarkjs.groth16.fullProve` = wtns.calculate, then groth16.prove
with a .zkey).  No .zkey exists offline and a real one cannot be made without the ceremony files, so the tests make
their own proving key from a trapdoor (tau, alpha, beta, gamma, delta) -- a "toy" key: insecure by construction, structurally the key
snarkjs writes.  What PINS this module: a proof assembled from these definitions must be ACCEPTED by
oracle/pyref/bn254_pairing.groth16_verify, and that verifier is pinned on the reference's own proof
(packages/rust-verifier/tests/data/proof_of_twitter, verifier semantics packages/rust-verifier/src/verifier_utils.rs:20-130).  The
verification equation therefore fixes every convention below up to mutual consistency of set-up and prover -- which is the freedom
snarkjs itself has.  The conventions are snarkjs' [EXT, restated: src/zkey_new.js, src/groth16_prove.js]:

  domain      n = 2^power >= m + nPublic + 1, w = ffjavascript's root of that order (oracle/pyref/ntt.py)
  matrices    A gets nPublic + 1 extra rows  A[m + s] = w_s  (s = 0 .. nPublic; B = C = 0 there): they make the public wires' polynomials
              linearly independent.  C_T is taken as A_T o B_T by the prover (equal to C.w for a valid witness).
  wire polys  a_i(x) = sum_j A[j][i] L_j(x) over the domain (same for b_i, c_i); L_j the Lagrange basis
  key         A_i = [a_i(tau)]_1, B1_i = [b_i(tau)]_1, B2_i = [b_i(tau)]_2,
              C_i = [(beta a_i + alpha b_i + c_i)(tau) / delta]_1 for private wires, IC_i = [(...) / gamma]_1 for wire 0 and the public ones
              H_j = [L'_j(tau) Z(tau) / (Z(x_j) delta)]_1 with x_j = inc w^j the odd coset of the doubled domain, L'_j its Lagrange basis,
                    Z(x) = x^n - 1 (Z(x_j) = -2): sum_j P_odd[j] H_j = h(tau) Z(tau) / delta for P_odd = the evaluations of a b - c on the coset
  proof       pi_a = alpha + sum w_i A_i + r delta,  pi_b = beta + sum w_i B2_i + s delta  (pib1 the same in G1),
              pi_c = sum_priv w_i C_i + sum_j P_odd[j] H_j + s pi_a + r pib1 - r s delta
Everything here is a SCALAR (the discrete logarithm of the group element, known because the trapdoor is): a multi-exponentiation of
the device must equal (sum_i k_i s_i) G for bases s_i G -- checked with one scalar multiplication.  Pure Python integers.
"""
import random

from oracle.pyref import bn254_g1 as G1
from oracle.pyref import bn254_g2 as G2
from oracle.pyref import ntt

R = G1.R


def _inv(x):
    return pow(x % R, R - 2, R)


def _batch_inv(xs):
    pref = [1]
    for x in xs:
        pref.append(pref[-1] * x % R)
    inv = _inv(pref[-1])
    out = [0] * len(xs)
    for i in range(len(xs) - 1, -1, -1):
        out[i] = inv * pref[i] % R
        inv = inv * xs[i] % R
    return out


def domain_power(m, n_public):
    need = m + n_public + 1
    p = 0
    while (1 << p) < need:
        p += 1
    return max(p, 1)


def lagrange_at(tau, power, shift=1):
    """L_j(tau) for the points x_j = shift * w^j, j < 2^power:  L_j(t) = Zs(t) x_j / (n x_j^n (t - x_j)),  Zs(t) = t^n - shift^n"""
    n = 1 << power
    w = ntt.root(power)
    xs = [0] * n
    x = shift % R
    for j in range(n):
        xs[j] = x
        x = x * w % R
    sn = pow(shift, n, R)
    zs = (pow(tau, n, R) - sn) % R
    den = _batch_inv([(tau - xj) % R for xj in xs])
    k = zs * _inv(n * sn) % R
    return [k * xs[j] % R * den[j] % R for j in range(n)]


class ToyKey:
    pass


def setup(n_wires, n_public, constraints, seed=1):
    """constraints: list of (a, b, c) dicts wire -> coefficient (tests/r1cs_util.py's form); wires 1 .. n_public are the public ones"""
    rng = random.Random(seed)
    k = ToyKey()
    k.n_wires, k.n_public, k.m = n_wires, n_public, len(constraints)
    k.power = domain_power(k.m, n_public)
    k.n = 1 << k.power
    k.tau, k.alpha, k.beta, k.gamma, k.delta = (rng.randrange(2, R) for _ in range(5))
    lag = lagrange_at(k.tau, k.power)
    k.lag = lag
    a, b, c = [0] * n_wires, [0] * n_wires, [0] * n_wires
    for j, (ra, rb, rc) in enumerate(constraints):
        lj = lag[j]
        for i, v in ra.items():
            a[i] = (a[i] + v * lj) % R
        for i, v in rb.items():
            b[i] = (b[i] + v * lj) % R
        for i, v in rc.items():
            c[i] = (c[i] + v * lj) % R
    for s in range(n_public + 1):                       # the extra rows of A
        a[s] = (a[s] + lag[k.m + s]) % R
    k.a_tau, k.b_tau, k.c_tau = a, b, c
    dinv, ginv = _inv(k.delta), _inv(k.gamma)
    mix = [(k.beta * a[i] + k.alpha * b[i] + c[i]) % R for i in range(n_wires)]
    k.ic = [mix[i] * ginv % R for i in range(n_public + 1)]
    k.c_key = [0] * (n_public + 1) + [mix[i] * dinv % R for i in range(n_public + 1, n_wires)]     # (zkey section 8 starts at wire nPublic + 1)
    inc = ntt.coset_inc(k.power)
    lodd = lagrange_at(k.tau, k.power, inc)
    zt = (pow(k.tau, k.n, R) - 1) % R
    hk = zt * _inv(-2 * k.delta) % R
    k.h_key = [x * hk % R for x in lodd]
    k.z_tau = zt
    return k


def abc_rows(key, constraints, w):
    """A_T, B_T, C_T of groth16_prove.js buildABC1, zero-padded to the domain (C_T = A_T o B_T)"""
    ev = lambda d: sum(v * w[i] for i, v in d.items()) % R
    A = [ev(ra) for ra, _, _ in constraints] + [w[s] % R for s in range(key.n_public + 1)]
    B = [ev(rb) for _, rb, _ in constraints] + [0] * (key.n_public + 1)
    A += [0] * (key.n - len(A))
    B += [0] * (key.n - len(B))
    return A, B, [x * y % R for x, y in zip(A, B)]


def prove_scalars(key, constraints, w, r, s):
    """discrete logarithms of (pi_a, pi_b, pi_c) and of the partial sums a device prover forms:
    {'a': sum w_i A_i, 'b': sum w_i B_i, 'c': sum_priv w_i C_i, 'h': sum P_odd[j] H_j, 'pi_a', 'pi_b', 'pi_c'}"""
    w = [x % R for x in w]
    sa = sum(x * y for x, y in zip(w, key.a_tau)) % R
    sb = sum(x * y for x, y in zip(w, key.b_tau)) % R
    sc = sum(w[i] * key.c_key[i] for i in range(key.n_public + 1, key.n_wires)) % R
    # h(tau) Z(tau) / delta without a transform: a(tau) b(tau) - c(tau) with c the interpolation of A_T o B_T
    A, B, C = abc_rows(key, constraints, w)
    ct = sum(x * y for x, y in zip(C, key.lag)) % R
    sh = (sa * sb - ct) % R * _inv(key.delta) % R
    pa = (key.alpha + sa + r * key.delta) % R
    pb = (key.beta + sb + s * key.delta) % R
    pc = (sc + sh + s * pa + r * pb - r * s % R * key.delta) % R
    return {"a": sa, "b": sb, "c": sc, "h": sh, "pi_a": pa, "pi_b": pb, "pi_c": pc}


def h_scalar_from_evaluations(key, p_odd):
    """sum_j P_odd[j] H_j as a scalar: what the H multi-exponentiation of the device must equal, from ITS scalars"""
    return sum(x * y for x, y in zip(p_odd, key.h_key)) % R


# ---- snarkjs JSON -------------------------------------------------------------------------------------------------------
def g1_json(p):
    return ["0", "1", "0"] if p is None else [str(p[0]), str(p[1]), "1"]


def g2_json(p):
    return [["0", "0"], ["1", "0"], ["0", "0"]] if p is None else [[str(p[0][0]), str(p[0][1])], [str(p[1][0]), str(p[1][1])], ["1", "0"]]


def vkey_json(key):
    return {"protocol": "groth16", "curve": "bn128", "nPublic": key.n_public,
            "vk_alpha_1": g1_json(G1.mul(key.alpha, G1.G)), "vk_beta_2": g2_json(G2.mul(key.beta, G2.G2)),
            "vk_gamma_2": g2_json(G2.mul(key.gamma, G2.G2)), "vk_delta_2": g2_json(G2.mul(key.delta, G2.G2)),
            "IC": [g1_json(G1.mul(x, G1.G)) for x in key.ic]}


def proof_json(sc):
    return {"pi_a": g1_json(G1.mul(sc["pi_a"], G1.G)), "pi_b": g2_json(G2.mul(sc["pi_b"], G2.G2)), "pi_c": g1_json(G1.mul(sc["pi_c"], G1.G)),
            "protocol": "groth16", "curve": "bn128"}
