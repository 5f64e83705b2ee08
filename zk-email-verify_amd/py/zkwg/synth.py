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


def synthetic_body(rng, length, soft_breaks=False):
    """printable ASCII, <= 76-char lines, CRLF line ends, exactly one trailing CRLF, no trailing spaces.
    soft_breaks: quoted-printable style, about half of the lines end in a soft line break "=\\r\\n"
    (and no other '=' appears, as in QP text without escapes)."""
    out = bytearray()
    while len(out) < length - 2:
        n = min(76, length - 2 - len(out) - 2)
        if n <= 0:
            break
        if soft_breaks:
            line = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz ABCDEFGHIJ0123456789.,;:!?-") for _ in range(n))
            line = line.rstrip(b" ") + b"x" * (n - len(line.rstrip(b" ")))
            if n == 76 and rng.random() < 0.5:
                line = line[:75] + b"="
        else:
            line = bytes(rng.randrange(0x21, 0x7F) for _ in range(n))
        out += line + b"\r\n"
    while len(out) < length - 2:
        out[-2:-2] = b"x"
    return bytes(out[:length - 2]) + b"\r\n" if len(out) >= length else bytes(out)


def synthetic_dkim_result(seed, index, body_len=1024, soft_breaks=False, body=None):
    """dict(headers, body, bodyHash, publicKey, signature) as `verifyDKIMSignature` would return.
    `body`: use these canonical body bytes instead of a generated body."""
    rng = random.Random((seed << 20) ^ index)
    key = test_key()
    if body is None:
        body = synthetic_body(rng, body_len, soft_breaks)
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


def packed_batch(circuit, seed, n, body_len=1024, first_index=0):
    """n packed input records (bytes, circuit.in_stride each) for an EmailVerifier circuit, built
    exactly as generate_email_verifier_inputs_from_dkim_result would (input-generators.ts:190-252)
    but written straight into the packed record layout (no decimal-string round trip).
    Also returns the per-email field arrays for the oracle."""
    from . import inputs as gen
    from ._lib import (IN_HEADER, IN_BODY, IN_PRECOMPUTED_SHA, IN_PUBKEY, IN_SIGNATURE, IN_HEADER_LEN,
                       IN_BODY_LEN, IN_BODY_HASH_INDEX, IN_DECODED_BODY)
    cfg = circuit.cfg
    stride = circuit.in_stride
    off = [circuit.lib.zkwg_input_offset(circuit.h, f) for f in range(12)]
    rslb = bool(cfg.remove_soft_line_breaks)
    buf = bytearray(n * stride)
    fields = {"header": bytearray(), "hlen": [], "body": bytearray(), "blen": [], "pre": bytearray(),
              "pubkey": bytearray(), "sig": bytearray(), "bhi": [], "decoded": bytearray()}

    def limbs16(x):
        return b"".join(((x >> (121 * i)) & ((1 << 121) - 1)).to_bytes(16, "little") for i in range(17))

    for i in range(n):
        d = synthetic_dkim_result(seed, first_index + i, body_len, soft_breaks=rslb)
        hp, hl = gen.sha256_pad(d["headers"], cfg.max_header)
        base = i * stride
        buf[base + off[IN_HEADER]:base + off[IN_HEADER] + cfg.max_header] = hp
        buf[base + off[IN_HEADER_LEN]:base + off[IN_HEADER_LEN] + 4] = hl.to_bytes(4, "little")
        pk, sg = limbs16(d["publicKey"]), limbs16(d["signature"])
        buf[base + off[IN_PUBKEY]:base + off[IN_PUBKEY] + 272] = pk
        buf[base + off[IN_SIGNATURE]:base + off[IN_SIGNATURE] + 272] = sg
        fields["header"] += hp; fields["hlen"].append(hl); fields["pubkey"] += pk; fields["sig"] += sg
        if not cfg.ignore_body_hash_check:
            body = d["body"]
            sha_len = ((len(body) + 63 + 65) // 64) * 64
            bp, bpl = gen.sha256_pad(body, max(cfg.max_body, sha_len))
            pre, rem, rem_len = gen.generate_partial_sha(bp, bpl, None, cfg.max_body)
            bhi = d["headers"].find(d["bodyHash"].encode())
            buf[base + off[IN_BODY]:base + off[IN_BODY] + cfg.max_body] = rem
            buf[base + off[IN_PRECOMPUTED_SHA]:base + off[IN_PRECOMPUTED_SHA] + 32] = pre
            buf[base + off[IN_BODY_LEN]:base + off[IN_BODY_LEN] + 4] = rem_len.to_bytes(4, "little")
            buf[base + off[IN_BODY_HASH_INDEX]:base + off[IN_BODY_HASH_INDEX] + 4] = bhi.to_bytes(4, "little")
            fields["body"] += rem; fields["blen"].append(rem_len); fields["pre"] += pre; fields["bhi"].append(bhi)
            if rslb:
                clean, _ = gen.remove_soft_line_breaks(rem)
                buf[base + off[IN_DECODED_BODY]:base + off[IN_DECODED_BODY] + cfg.max_body] = clean
                fields["decoded"] += clean
    return bytes(buf), fields
