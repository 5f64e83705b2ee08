"""oracle/pyref/groth16.py (toy set-up with a known trapdoor + the prover's arithmetic) against the PINNED verifier
oracle/pyref/bn254_pairing.py: proofs assembled from its definitions verify, tampered ones do not, and the H path through the coset
evaluations (oracle/pyref/ntt.py = what the device computes) gives the same element as the direct h(tau).  CPU only."""
import random

import pytest

import r1cs_util
from oracle.pyref import bn254_pairing as P
from oracle.pyref import groth16 as G
from oracle.pyref import ntt

R = G.R


def _case(n_wires, cons, w, n_public, seed):
    key = G.setup(n_wires, n_public, cons, seed)
    rng = random.Random(seed + 1)
    r, s = rng.randrange(R), rng.randrange(R)
    sc = G.prove_scalars(key, cons, w, r, s)
    return key, sc


@pytest.mark.parametrize("name", ["num2bits", "sigma", "random"])
def test_toy_proofs_verify_under_the_pinned_verifier(name):
    if name == "num2bits":
        n_wires, cons, w = r1cs_util.num2bits_system(6, 45)
        n_public = 1
    elif name == "sigma":
        n_wires, cons, w = r1cs_util.sigma_chain_system(7, 3)
        n_public = 1
    else:
        n_wires = 24
        cons, w = r1cs_util.random_system(11, n_wires, 19)
        n_public = 3
    assert r1cs_util.first_violation(cons, w) is None
    key, sc = _case(n_wires, cons, w, n_public, 5)
    vk, proof = G.vkey_json(key), G.proof_json(sc)
    pub = [str(w[i] % R) for i in range(1, n_public + 1)]
    assert P.groth16_verify(vk, pub, proof)
    bad = list(pub)
    bad[0] = str((int(bad[0]) + 1) % R)
    assert not P.groth16_verify(vk, bad, proof)
    # a wrong witness (one private wire changed) gives a proof that does not verify
    w2 = list(w)
    w2[-1] = (w2[-1] + 1) % R
    sc2 = G.prove_scalars(key, cons, w2, 3, 4)
    assert not P.groth16_verify(vk, pub, G.proof_json(sc2))


def test_h_through_the_coset_evaluations_equals_h_of_tau():
    """sum_j P_odd[j] H_j with P_odd = the odd-coset evaluations of a b - c (ntt.h_evaluations: the device's stage 2) is the
    element h(tau) Z(tau) / delta the proof needs"""
    n_wires = 20
    cons, w = r1cs_util.random_system(3, n_wires, 27)
    key, sc = _case(n_wires, cons, w, 2, 9)
    A, B, C = G.abc_rows(key, cons, w)
    p_odd = ntt.h_evaluations(A, B, C, key.power)
    assert G.h_scalar_from_evaluations(key, p_odd) == sc["h"]
    # without the extra public rows the evaluations differ (the rows belong to the A block)
    A0 = A[:key.m] + [0] * (key.n - key.m)
    C0 = [x * y % R for x, y in zip(A0, B)]
    assert ntt.h_evaluations(A0, B, C0, key.power) != p_odd
