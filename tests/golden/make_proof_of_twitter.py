"""Copies the reference's one stored Groth16 proof with its verification key into tests/golden/proof_of_twitter/ (needs
/root/reference; run from the repo root):

    python tests/golden/make_proof_of_twitter.py

Source: packages/rust-verifier/tests/data/proof_of_twitter/{proof,vkey,public}.json -- a real snarkjs proof of the reference's
Twitter circuit, the vector its own verifier test checks (packages/rust-verifier/src/verifier_utils.rs:20-130).  These are test
vectors (numbers), not code; they pin oracle/pyref/bn254_pairing.py, which in turn judges what the prover stages produce
(tests/test_pairing_oracle.py, tests/test_prove.py).  The GPU box has no /root/reference, hence the copy.
"""
import json
import os

SRC = "/root/reference/packages/rust-verifier/tests/data/proof_of_twitter"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "proof_of_twitter")

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for name in ("proof.json", "vkey.json", "public.json"):
        with open(os.path.join(SRC, name)) as f:
            obj = json.load(f)
        with open(os.path.join(DST, name), "w") as f:
            json.dump(obj, f, indent=1)
            f.write("\n")
        print("wrote", os.path.join(DST, name))
