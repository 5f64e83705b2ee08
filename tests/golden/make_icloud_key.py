"""Recovers the RSA-2048 modulus of the DKIM key `d=icloud.com; s=1a1hai` OFFLINE and writes
tests/golden/icloud_1a1hai.json (run in the build container; /root/reference does not exist on the GPU box).

The reference fetches this key from live DNS (packages/helpers/src/dkim/index.ts:105-131), so it is stored
nowhere in the tree.  But the tree holds TWO different emails signed by it:
    packages/circuits/tests/test-emails/test.eml                      (t=1693038337)
    packages/helpers/tests/test-data/email-good-large.eml             (t=1712141644)
For an RSASSA-PKCS1-v1_5 signature s of the encoded message EM:  s^e = EM (mod N), i.e. N | s^e - EM.
With two signatures  N | gcd(s1^e - EM1, s2^e - EM2); the cofactor is (almost surely) a product of tiny
primes, which trial division strips.  e = 65537, so the two powers are 134-Mbit integers: computed with
libgmp through ctypes (__gmpz_pow_ui ~1 s each, __gmpz_gcd ~1 min).

EM = 00 01 ff..ff 00 || DigestInfo(SHA-256) || SHA-256(relaxed-canonical signed header)  (RFC 8017 9.2;
the circuit's RSAPad, packages/circuits/lib/rsa.circom:101-181, builds the same string).
The header canonicalisation restates packages/helpers/src/lib/mailauth/header/relaxed.ts:5-78 and
tools.ts:441-454 (formatRelaxedLine) like make_test_eml_fixture.py does.

Self-checks before anything is written: N has exactly 2048 bits, pow(s, 65537, N) == EM for BOTH emails,
and the body hashes match the bh= tags."""
import base64
import ctypes
import ctypes.util
import hashlib
import json
import os
import re
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [
    ("packages/circuits/tests/test-emails/test.eml", "/root/reference/packages/circuits/tests/test-emails/test.eml"),
    ("packages/helpers/tests/test-data/email-good-large.eml", "/root/reference/packages/helpers/tests/test-data/email-good-large.eml"),
]
DIGEST_INFO = bytes.fromhex("3031300d060960864801650304020105000420")
E = 65537


def canonicalise(path):
    """-> dict(canonical header bytes, canonical body bytes, bh, signature int)."""
    raw = open(path, "rb").read().replace(b"\r\n", b"\n").replace(b"\n", b"\r\n")
    head, _, body = raw.partition(b"\r\n\r\n")
    fields = []
    for line in head.split(b"\r\n"):
        if line[:1] in (b" ", b"\t") and fields:
            fields[-1] += b"\r\n" + line
        else:
            fields.append(line)

    def relaxed(field):
        name, _, value = field.partition(b":")
        value = re.sub(rb"\r\n", b"", value)
        value = re.sub(rb"[ \t]+", b" ", value).strip()
        return name.strip().lower() + b":" + value

    dk = [f for f in fields if f.lower().startswith(b"dkim-signature:")][0]
    dkr = relaxed(dk)
    tags = {}
    for part in dkr.partition(b":")[2].split(b";"):
        k, _, v = part.strip().partition(b"=")
        tags[k.decode()] = v
    assert tags["d"] == b"icloud.com" and tags["s"] == b"1a1hai" and tags["a"] == b"rsa-sha256" and tags["c"] == b"relaxed/relaxed"
    names = tags["h"].decode().replace(" ", "").split(":")
    # signed fields in h= order; a name listed k times takes the k-th instance from the bottom (RFC 6376 5.4.2)
    used = {}
    out = []
    for name in names:
        cands = [f for f in fields if f.lower().startswith(name.lower().encode() + b":")]
        k = used.get(name.lower(), 0)
        used[name.lower()] = k + 1
        if k < len(cands):
            out.append(relaxed(cands[len(cands) - 1 - k]))
    dk_empty = re.sub(rb"b=[^;]*$", b"b=", dkr)
    canon_header = b"\r\n".join(out) + b"\r\n" + dk_empty
    lines = [re.sub(rb"[ \t]+", b" ", ln).rstrip(b" ") for ln in body.split(b"\r\n")]
    while lines and lines[-1] == b"":
        lines.pop()
    canon_body = b"\r\n".join(lines) + b"\r\n"
    bh = re.sub(rb"\s+", b"", tags["bh"]).decode()
    sig = int.from_bytes(base64.b64decode(re.sub(rb"\s+", b"", tags["b"])), "big")
    return {"header": canon_header, "body": canon_body, "bh": bh, "signature": sig, "t": tags["t"].decode()}


def emsa(header, em_len=256):
    t = DIGEST_INFO + hashlib.sha256(header).digest()
    return int.from_bytes(b"\x00\x01" + b"\xff" * (em_len - len(t) - 3) + b"\x00" + t, "big")


class Mpz(ctypes.Structure):
    _fields_ = [("alloc", ctypes.c_int), ("size", ctypes.c_int), ("d", ctypes.c_void_p)]


def gmp():
    lib = ctypes.CDLL(ctypes.util.find_library("gmp") or "libgmp.so.10")
    for f in ("__gmpz_init", "__gmpz_clear"):
        getattr(lib, f).argtypes = [ctypes.POINTER(Mpz)]
    lib.__gmpz_set_str.argtypes = [ctypes.POINTER(Mpz), ctypes.c_char_p, ctypes.c_int]
    lib.__gmpz_pow_ui.argtypes = [ctypes.POINTER(Mpz), ctypes.POINTER(Mpz), ctypes.c_ulong]
    lib.__gmpz_sub.argtypes = [ctypes.POINTER(Mpz)] * 3
    lib.__gmpz_gcd.argtypes = [ctypes.POINTER(Mpz)] * 3
    lib.__gmpz_sizeinbase.argtypes = [ctypes.POINTER(Mpz), ctypes.c_int]
    lib.__gmpz_sizeinbase.restype = ctypes.c_size_t
    lib.__gmpz_get_str.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(Mpz)]
    lib.__gmpz_get_str.restype = ctypes.c_char_p
    return lib


def main():
    mails = [canonicalise(p) for _, p in SOURCES]
    for m in mails:
        assert base64.b64encode(hashlib.sha256(m["body"]).digest()).decode() == m["bh"], "body hash != bh"
    lib = gmp()
    diffs = []
    for m in mails:
        s, em, d = Mpz(), Mpz(), Mpz()
        for x in (s, em, d):
            lib.__gmpz_init(ctypes.byref(x))
        lib.__gmpz_set_str(ctypes.byref(s), hex(m["signature"])[2:].encode(), 16)
        lib.__gmpz_set_str(ctypes.byref(em), hex(emsa(m["header"]))[2:].encode(), 16)
        t0 = time.time()
        lib.__gmpz_pow_ui(ctypes.byref(d), ctypes.byref(s), E)
        lib.__gmpz_sub(ctypes.byref(d), ctypes.byref(d), ctypes.byref(em))
        print(f"s^65537 - EM: {lib.__gmpz_sizeinbase(ctypes.byref(d), 2)} bits, {time.time() - t0:.1f} s", flush=True)
        diffs.append(d)
    g = Mpz()
    lib.__gmpz_init(ctypes.byref(g))
    t0 = time.time()
    lib.__gmpz_gcd(ctypes.byref(g), ctypes.byref(diffs[0]), ctypes.byref(diffs[1]))
    print(f"gcd: {lib.__gmpz_sizeinbase(ctypes.byref(g), 2)} bits, {time.time() - t0:.1f} s", flush=True)
    n = int(lib.__gmpz_get_str(None, 16, ctypes.byref(g)).decode(), 16)
    # strip small cofactors
    p = 2
    while p < 100000:
        while n % p == 0 and n.bit_length() > 2048:
            n //= p
        p += 1 if p == 2 else 2
    assert n.bit_length() == 2048, n.bit_length()
    for m in mails:
        assert pow(m["signature"], E, n) == emsa(m["header"]), "signature does not verify under the recovered modulus"
    # DER SubjectPublicKeyInfo (what the p= tag of the DNS record carries)
    def der_len(k):
        return bytes([k]) if k < 128 else bytes([0x80 | len(k.to_bytes((k.bit_length() + 7) // 8, "big"))]) + k.to_bytes((k.bit_length() + 7) // 8, "big")
    def der_int(v):
        b = v.to_bytes((v.bit_length() + 8) // 8, "big")
        return b"\x02" + der_len(len(b)) + b
    rsa_pub = b"\x30" + der_len(len(der_int(n) + der_int(E))) + der_int(n) + der_int(E)
    bitstr = b"\x03" + der_len(len(rsa_pub) + 1) + b"\x00" + rsa_pub
    alg = bytes.fromhex("300d06092a864886f70d0101010500")
    spki = b"\x30" + der_len(len(alg + bitstr)) + alg + bitstr
    fixture = {
        "domain": "icloud.com", "selector": "1a1hai", "e": E,
        "modulus_hex": hex(n)[2:],
        "modulus_base64": base64.b64encode(n.to_bytes(256, "big")).decode(),
        "dns_p_tag_spki_base64": base64.b64encode(spki).decode(),
        "recovered_by": "gcd(s1^65537 - EM1, s2^65537 - EM2) over the two emails below (tests/golden/make_icloud_key.py); "
                        "both signatures verify under it",
        "emails": [
            {"source": src, "t": m["t"], "bh": m["bh"], "signature_hex": hex(m["signature"])[2:],
             "canonical_header_hex": m["header"].hex(), "canonical_body_sha256": hashlib.sha256(m["body"]).hexdigest(),
             "canonical_body_len": len(m["body"]), "header_sha256": hashlib.sha256(m["header"]).hexdigest()}
            for (src, _), m in zip(SOURCES, mails)],
    }
    json.dump(fixture, open(os.path.join(HERE, "icloud_1a1hai.json"), "w"), indent=1)
    print("modulus", hex(n)[:20], "...", hex(n)[-10:], "bits", n.bit_length())


if __name__ == "__main__":
    sys.exit(main())
