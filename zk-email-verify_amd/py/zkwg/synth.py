"""Deterministic synthetic DKIM-signed emails for tests and bench.py.

One fixed RSA-2048 key (data/synthetic_rsa2048_key.json, generated once with
`openssl genrsa 2048`), e = 65537.  Email i: a printable-ASCII body in <= 76-char CRLF
lines (already relaxed-canonical), `bh = base64(sha256(body))`, a `test.eml`-shaped signed
header that ends with the canonical `dkim-signature:` field (b= emptied), signature =
EMSA-PKCS1-v1_5(sha256(header))^d mod N.
"""
import base64
import hashlib
import json
import os
import random

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIGESTINFO = bytes.fromhex("3031300d060960864801650304020105000420")
_key = None


def test_key():
    global _key
    if _key is None:
        raw = json.load(open(os.path.join(_HERE, "data", "synthetic_rsa2048_key.json")))
        k = {n: int(raw[n], 16) for n in ("n", "d", "p", "q")}
        k["e"] = raw["e"]
        k["dp"], k["dq"], k["qinv"] = k["d"] % (k["p"] - 1), k["d"] % (k["q"] - 1), pow(k["q"], -1, k["p"])
        _key = k
    return _key


def pkcs1_sign_digest(key, digest: bytes) -> int:
    """EMSA-PKCS1-v1_5 (SHA-256) signature as an integer (CRT)."""
    klen = (key["n"].bit_length() + 7) // 8
    em = b"\x00\x01" + b"\xff" * (klen - 3 - len(_DIGESTINFO) - 32) + b"\x00" + _DIGESTINFO + digest
    m = int.from_bytes(em, "big")
    m1, m2 = pow(m % key["p"], key["dp"], key["p"]), pow(m % key["q"], key["dq"], key["q"])
    h = (key["qinv"] * (m1 - m2)) % key["p"]
    return m2 + h * key["q"]


def synthetic_body(rng, length):
    """printable ASCII, <= 76-char lines, CRLF line ends, exactly one trailing CRLF, no trailing spaces."""
    out = bytearray()
    while len(out) < length - 2:
        n = min(76, length - 2 - len(out) - 2)
        if n <= 0:
            break
        line = bytes(rng.randrange(0x21, 0x7F) for _ in range(n))
        out += line + b"\r\n"
    while len(out) < length - 2:
        out[-2:-2] = b"x"
    return bytes(out[:length - 2]) + b"\r\n" if len(out) >= length else bytes(out)


def synthetic_dkim_result(seed, index, body_len=1024):
    """dict(headers, body, bodyHash, publicKey, signature) as `verifyDKIMSignature` would return."""
    rng = random.Random((seed << 20) ^ index)
    key = test_key()
    body = synthetic_body(rng, body_len)
    bh = base64.b64encode(hashlib.sha256(body).digest()).decode()
    mid = "%016x" % rng.getrandbits(64)
    subj = "".join(chr(rng.randrange(0x61, 0x7B)) for _ in range(rng.randrange(8, 24)))
    headers = (
        f"from:Synthetic Sender <sender{index}@example.com>\r\n"
        "content-type:text/plain; charset=us-ascii\r\n"
        "mime-version:1.0 (Mac OS X Mail 16.0)\r\n"
        f"subject:{subj}\r\n"
        f"message-id:<{mid}@example.com>\r\n"
        "date:Sat, 14 Oct 2023 22:09:53 +0300\r\n"
        "to:recipient@example.org\r\n"
        "dkim-signature:v=1; a=rsa-sha256; c=relaxed/relaxed; d=example.com; s=sel; t=1697310606; "
        f"bh={bh}; h=from:Content-Type:Mime-Version:Subject:Message-Id:Date:To; b="
    ).encode()
    sig = pkcs1_sign_digest(key, hashlib.sha256(headers).digest())
    return {"headers": headers, "body": body, "bodyHash": bh, "publicKey": key["n"], "signature": sig}
