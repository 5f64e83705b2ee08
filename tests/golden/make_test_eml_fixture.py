"""Generates tests/golden/test_eml.json from the reference's own fixture
/root/reference/packages/circuits/tests/test-emails/test.eml (run in the build container; the
reference checkout does not exist on the GPU box).

Restates the relaxed canonicalisation the reference applies before hashing
(packages/helpers/src/lib/mailauth/header/relaxed.ts:5-78, tools.ts:441-454 formatRelaxedLine,
body/relaxed.ts) for this one email: signed headers in `h=` order, lower-cased names, unfolded,
single spaces, `dkim-signature` last with `b=` emptied and no trailing CRLF."""
import base64
import hashlib
import json
import os
import re

SRC = "/root/reference/packages/circuits/tests/test-emails/test.eml"
raw = open(SRC, "rb").read().replace(b"\r\n", b"\n").replace(b"\n", b"\r\n")
head, _, body = raw.partition(b"\r\n\r\n")
# unfold header fields
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
h = re.search(rb"[; ]h=([^;]+);", dk).group(1).decode().replace(" ", "").split(":")
out = []
for name in h:
    cands = [f for f in fields if f.lower().startswith(name.lower().encode() + b":")]
    out.append(relaxed(cands[-1]))
dkr = relaxed(dk)
dkr = re.sub(rb"b=[^;]*$", b"b=", dkr)
canon_header = b"\r\n".join(out) + b"\r\n" + dkr
# relaxed body: strip trailing whitespace per line, collapse WSP, remove trailing empty lines, end with CRLF
lines = [re.sub(rb"[ \t]+", b" ", ln).rstrip(b" ") for ln in body.split(b"\r\n")]
while lines and lines[-1] == b"":
    lines.pop()
canon_body = b"\r\n".join(lines) + b"\r\n"
bh = re.search(rb"bh=([^;]+);", dk).group(1).decode()
fixture = {
    "source": "packages/circuits/tests/test-emails/test.eml",
    "canonical_header_hex": canon_header.hex(),
    "canonical_body_hex": canon_body.hex(),
    "bh": bh,
    "header_sha256": hashlib.sha256(canon_header).hexdigest(),
    "body_sha256_b64": base64.b64encode(hashlib.sha256(canon_body).digest()).decode(),
    "body_hash_index": canon_header.decode("latin-1").find(bh),
    "rsa_test_ts_message_limbs": ["1156466847851242602709362303526378170", "191372789510123109308037416804949834", "7204"],
}
json.dump(fixture, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_eml.json"), "w"), indent=1)
print(len(canon_header), fixture["body_hash_index"], fixture["body_sha256_b64"] == bh)
