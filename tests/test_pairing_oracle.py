"""oracle/pyref/bn254_pairing.py pinned on the reference's own Groth16 vector (VERDICT r4 missing 5):
packages/rust-verifier/tests/data/proof_of_twitter/{proof,vkey,public}.json, checked by the reference with
packages/rust-verifier/src/verifier_utils.rs:20-130.  CPU only (pure Python integers)."""
import copy
import json
import os

import pytest

from oracle.pyref import bn254_g1 as G1
from oracle.pyref import bn254_g2 as G2
from oracle.pyref import bn254_pairing as P

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_of_twitter")


def _load():
    return tuple(json.load(open(os.path.join(D, n))) for n in ("vkey.json", "public.json", "proof.json"))


def test_golden_copy_equals_the_reference_files():
    src = "/root/reference/packages/rust-verifier/tests/data/proof_of_twitter"
    if not os.path.isdir(src):
        pytest.skip("needs /root/reference")
    for n in ("vkey.json", "public.json", "proof.json"):
        assert json.load(open(os.path.join(src, n))) == json.load(open(os.path.join(D, n)))


def test_the_reference_proof_verifies():
    vk, pub, pr = _load()
    assert vk["protocol"] == "groth16" and vk["curve"] == "bn128" and vk["nPublic"] == len(pub) == 3
    assert P.groth16_verify(vk, pub, pr)


@pytest.mark.parametrize("which", [0, 1, 2])
def test_a_changed_public_input_is_rejected(which):
    vk, pub, pr = _load()
    pub = list(pub)
    pub[which] = str(int(pub[which]) + 1)
    assert not P.groth16_verify(vk, pub, pr)


def test_a_changed_proof_element_is_rejected():
    vk, pub, pr = _load()
    # another point of the group in place of pi_a / pi_c / pi_b (stays on the curve: the check that fails is the pairing equation)
    for key in ("pi_a", "pi_c"):
        bad = copy.deepcopy(pr)
        x, y = G1.add(P.g1_from_json(pr[key]), G1.G)
        bad[key] = [str(x), str(y), "1"]
        assert not P.groth16_verify(vk, pub, bad)
    bad = copy.deepcopy(pr)
    (x0, x1), (y0, y1) = G2.add(P.g2_from_json(pr["pi_b"]), G2.G2)
    bad["pi_b"] = [[str(x0), str(x1)], [str(y0), str(y1)], ["1", "0"]]
    assert not P.groth16_verify(vk, pub, bad)
    # wrong number of public inputs, a public input outside the field
    assert not P.groth16_verify(vk, pub[:2], pr)
    assert not P.groth16_verify(vk, [str(int(pub[0]) + P.R)] + list(pub[1:]), pr)


def test_pairing_value_equals_the_one_snarkjs_stored():
    """vk_alphabeta_12 = e(vk_alpha_1, vk_beta_2) as written by snarkjs: equal coefficient by coefficient"""
    vk, _, _ = _load()
    ab = P.pairing_as_snarkjs(P.g1_from_json(vk["vk_alpha_1"]), P.g2_from_json(vk["vk_beta_2"]))
    assert P.to_snarkjs_f12(ab) == vk["vk_alphabeta_12"]


def test_bilinearity_and_non_degeneracy():
    e = P.pairing(G1.G, G2.G2)
    assert e != P.F12_ONE and P.f12_pow(e, P.R) == P.F12_ONE
    assert P.pairing(G1.mul(5, G1.G), G2.mul(7, G2.G2)) == P.f12_pow(e, 35)
    assert P.pairing(G1.neg(G1.G), G2.G2) == P.pairing(G1.G, G2.neg(G2.G2))
    assert P.pairing(None, G2.G2) == P.F12_ONE
