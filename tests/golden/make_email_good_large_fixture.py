"""Generates tests/golden/email_good_large.json from the reference's helper fixture
/root/reference/packages/helpers/tests/test-data/email-good-large.eml (build container only).
Relaxed body canonicalisation as in make_test_eml_fixture.py; self-check: SHA-256 of the canonical body
equals the bh= tag of the email's DKIM-Signature."""
import base64
import hashlib
import json
import os
import re

SRC = "/root/reference/packages/helpers/tests/test-data/email-good-large.eml"
raw = open(SRC, "rb").read().replace(b"\r\n", b"\n").replace(b"\n", b"\r\n")
head, _, body = raw.partition(b"\r\n\r\n")
lines = [re.sub(rb"[ \t]+", b" ", ln).rstrip(b" ") for ln in body.split(b"\r\n")]
while lines and lines[-1] == b"":
    lines.pop()
canon_body = b"\r\n".join(lines) + b"\r\n"
bh = re.search(rb"bh=([A-Za-z0-9+/=\s]+);", head, re.S).group(1)
bh = re.sub(rb"\s+", b"", bh).decode()
ok = base64.b64encode(hashlib.sha256(canon_body).digest()).decode() == bh
fixture = {
    "source": "packages/helpers/tests/test-data/email-good-large.eml",
    "canonical_body_hex": canon_body.hex(),
    "bh": bh,
    "body_hash_matches_bh": ok,
    # packages/helpers/tests/input-generators.test.ts:39-53
    "selector": "thousands",
    "expected_body_prefix": "h hundreds of thousands of blocks.",
}
json.dump(fixture, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "email_good_large.json"), "w"), indent=1)
print(len(canon_body), ok)
