"""Test-only: constraints of EmailVerifier(N, M, 121, 17, 0, 0, 0, 0) over the kept wires, assembled from
tests/sha_r1cs.py, tests/rsa_r1cs.py and the Poseidon block of tests/test_r1cs.py following
packages/circuits/email-verifier.circom:42-174: header Sha256Bytes -> shaHi/shaLo (PackBits) and the RSA
message limbs -> RSAVerifier65537; body Sha256BytesPartial; PoseidonLarge -> pubkeyHash.  Not covered (left to
the oracle parity tests): AssertZeroPadding, BodyHashRegex, SelectRegexReveal, Base64Decode, the body-hash
comparison and the length Num2Bits."""
import sha_r1cs
import rsa_r1cs
from sha_r1cs import lc_add, const, wire


def email_verifier_constraints(symbols, N, M, poseidon_block):
    slot_of = {n: s for s, n in symbols}
    b = sha_r1cs.Builder(slot_of)
    _, sha = sha_r1cs.sha256_bytes_constraints(symbols, N, "main.anon_Sha256Bytes", "main.emailHeader", builder=b,
                                               length="main.emailHeaderLength")
    sha_r1cs.sha256_bytes_constraints(symbols, M, "main.anon_Sha256BytesPartial", "main.emailBody",
                                      pre="main.precomputedSHA", builder=b, length="main.emailBodyLength")
    cons = b.cons
    # PackBits(256, 128) (utils/bytes.circom:194-210 via email-verifier.circom:68-71): sha is fed in as is, chunk 0
    # -> shaHi, chunk 1 -> shaLo, each chunk big-endian: out[i] = sum_j in[128 i + j] * 2^(127 - j)
    for i, name in enumerate(("main.shaHi", "main.shaLo")):
        acc = lc_add({}, wire(slot_of[name]), -1)
        for j in range(128):
            acc = lc_add(acc, sha[128 * i + j], 1 << (127 - j))
        cons.append((acc, const(1), {}))
    # rsaMessage[i \\ n].in[i % n] <== sha[255 - i] (email-verifier.circom:74-84)
    n = 121
    message = [{} for _ in range(17)]
    for i in range(256):
        message[i // n] = lc_add(message[i // n], sha[255 - i], 1 << (i % n))
    rb = rsa_r1cs.RsaBuilder(slot_of, "main.rsaVerifier")
    arr = lambda nm: [wire(slot_of[f"main.{nm}[{i}]"]) for i in range(17)]
    cons += rb.rsa_verifier(message, arr("signature"), arr("pubkey"))
    cons += poseidon_block(symbols, len(symbols))
    return cons
