"""Host-side input generation: mirrors packages/helpers/src/{input-generators,sha-utils,
binary-format}.ts of the reference (the `CircuitInput` object is the boundary format).

  sha256_pad                      <- sha-utils.ts:88-111      sha256Pad
  generate_partial_sha            <- sha-utils.ts:30-80       generatePartialSHA
  partial_sha                     <- sha-utils.ts:82-85 + lib/fast-sha256.ts:240-251 cacheState
  to_circom_bigint_bytes          <- binary-format.ts:71-83   toCircomBigIntBytes
  generate_email_verifier_inputs_from_dkim_result
                                  <- input-generators.ts:190-252
"""
import struct

MAX_HEADER_PADDED_BYTES = 1024  # constants.ts:2
MAX_BODY_PADDED_BYTES = 1536    # constants.ts:3
CIRCOM_BIGINT_N = 121           # constants.ts:5
CIRCOM_BIGINT_K = 17            # constants.ts:6

_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
_IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def _rotr(x, r):
    return ((x >> r) | (x << (32 - r))) & 0xFFFFFFFF


def _compress(st, block):
    w = list(struct.unpack(">16I", block))
    for t in range(16, 64):
        s0 = _rotr(w[t - 15], 7) ^ _rotr(w[t - 15], 18) ^ (w[t - 15] >> 3)
        s1 = _rotr(w[t - 2], 17) ^ _rotr(w[t - 2], 19) ^ (w[t - 2] >> 10)
        w.append((w[t - 16] + s0 + w[t - 7] + s1) & 0xFFFFFFFF)
    a, b, c, d, e, f, g, h = st
    for t in range(64):
        t1 = (h + (_rotr(e, 6) ^ _rotr(e, 11) ^ _rotr(e, 25)) + ((e & f) ^ (~e & g)) + _K[t] + w[t]) & 0xFFFFFFFF
        t2 = ((_rotr(a, 2) ^ _rotr(a, 13) ^ _rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & 0xFFFFFFFF
        h, g, f, e, d, c, b, a = g, f, e, (d + t1) & 0xFFFFFFFF, c, b, a, (t1 + t2) & 0xFFFFFFFF
    return [(x + y) & 0xFFFFFFFF for x, y in zip(st, (a, b, c, d, e, f, g, h))]


def partial_sha(msg: bytes) -> bytes:
    """SHA-256 midstate after absorbing `msg` (a multiple of 64 bytes), as 32 big-endian bytes
    (sha-utils.ts:82-85 -> lib/fast-sha256.ts:240-251 cacheState)."""
    assert len(msg) % 64 == 0
    st = list(_IV)
    for i in range(0, len(msg), 64):
        st = _compress(st, msg[i:i + 64])
    return struct.pack(">8I", *st)


def sha256_pad(message: bytes, max_sha_bytes: int):
    """sha-utils.ts:88-111 -> (padded bytes of length max_sha_bytes, padded message length)."""
    msg_len_bits = len(message) * 8
    res = message + b"\x80"
    while (len(res) * 8 + 64) % 512 != 0:
        res += b"\x00"
    res += msg_len_bits.to_bytes(8, "big")
    assert (len(res) * 8) % 512 == 0, "Padding did not complete properly!"
    message_len = len(res)
    if len(res) > max_sha_bytes:
        raise ValueError(
            f"Padding to max length did not complete properly! Your padded message is {len(res)} long but max is {max_sha_bytes}!")
    res += b"\x00" * (max_sha_bytes - len(res))
    return res, message_len


def find_index_in_uint8array(array: bytes, selector: bytes) -> int:
    """sha-utils.ts:9-24, literally (on a mismatch j restarts at 0 and i still advances, so this is NOT a
    general substring search: "aab" is not found in "aaab")."""
    i = j = 0
    while i < len(array):
        if array[i] == selector[j]:
            j += 1
            if j == len(selector):
                return i - j + 1
        else:
            j = 0
        i += 1
    return -1


def generate_partial_sha(body: bytes, body_length: int, selector_string, max_remaining_body_length: int):
    """sha-utils.ts:30-80."""
    selector_index = 0
    if selector_string:
        sel = selector_string.encode() if isinstance(selector_string, str) else selector_string
        selector_index = find_index_in_uint8array(body, sel)
        if selector_index == -1:
            raise ValueError(f'SHA precompute selector "{selector_string}" not found in the body')
    sha_cutoff_index = (selector_index // 64) * 64
    precompute_text = body[:sha_cutoff_index]
    body_remaining = body[sha_cutoff_index:]
    body_remaining_length = body_length - len(precompute_text)
    if body_remaining_length > max_remaining_body_length:
        raise ValueError(
            f"Remaining body {body_remaining_length} after the selector is longer than max ({max_remaining_body_length})")
    if len(body_remaining) % 64 != 0:
        raise ValueError("Remaining body was not padded correctly with int64s")
    body_remaining = body_remaining + b"\x00" * max(0, max_remaining_body_length - len(body_remaining))
    return partial_sha(precompute_text), body_remaining, body_remaining_length


def to_circom_bigint_bytes(num: int):
    """binary-format.ts:71-83: 17 x 121-bit limbs as decimal strings."""
    msk = (1 << CIRCOM_BIGINT_N) - 1
    return [str((num >> (i * CIRCOM_BIGINT_N)) & msk) for i in range(CIRCOM_BIGINT_K)]


def remove_soft_line_breaks(body: bytes):
    """input-generators.ts:127-158: drops every "=\r\n", zero-pads back to len(body); also returns
    the clean -> original position map."""
    result = bytearray()
    position_map = {}
    i = 0
    n = len(body)
    while i < n:
        if i + 2 < n and body[i] == 61 and body[i + 1] == 13 and body[i + 2] == 10:
            i += 3
        else:
            position_map[len(result)] = i
            result.append(body[i])
            i += 1
    result.extend(b"\0" * (n - len(result)))
    return bytes(result), position_map


def get_adjusted_selector(original_body: bytes, selector: str, clean_content: bytes, position_map) -> str:
    """input-generators.ts:44-105 (findSelectorInCleanContent + getAdjustedSelector)."""
    body_string = original_body.decode("utf-8", errors="replace")
    if selector in body_string:
        return selector
    clean_string = clean_content.decode("utf-8", errors="replace")
    selector_index = clean_string.find(selector)
    if selector_index == -1:
        raise ValueError(f'SHA precompute selector "{selector}" not found in cleaned body')
    if selector_index not in position_map:
        raise ValueError("Failed to map selector position to original body")
    original_index = position_map[selector_index]
    return body_string[original_index:original_index + len(selector) + 3]


def generate_email_verifier_inputs_from_dkim_result(dkim, max_headers_length=None, max_body_length=None,
                                                    ignore_body_hash_check=False, sha_precompute_selector=None,
                                                    enable_header_masking=False, header_mask=None,
                                                    enable_body_masking=False, body_mask=None,
                                                    remove_soft_line_breaks_flag=False):
    """input-generators.ts:190-252.  `dkim` = dict(headers: bytes, body: bytes, bodyHash: str,
    publicKey: int, signature: int) -- the DKIMVerificationResult fields the function uses."""
    headers = dkim["headers"]
    message_padded, message_padded_len = sha256_pad(headers, max_headers_length or MAX_HEADER_PADDED_BYTES)
    inputs = {
        "emailHeader": [str(b) for b in message_padded],
        "emailHeaderLength": str(message_padded_len),
        "pubkey": to_circom_bigint_bytes(dkim["publicKey"]),
        "signature": to_circom_bigint_bytes(dkim["signature"]),
    }
    if enable_header_masking:
        inputs["headerMask"] = list(header_mask)
    if not ignore_body_hash_check:
        body, body_hash = dkim.get("body"), dkim.get("bodyHash")
        if not body and body != b"" or not body_hash:
            raise ValueError("body and bodyHash are required when ignoreBodyHashCheck is false")
        body_hash_index = headers.decode("latin-1").find(body_hash)
        max_body = max_body_length or MAX_BODY_PADDED_BYTES
        body_sha_length = ((len(body) + 63 + 65) // 64) * 64
        body_padded, body_padded_len = sha256_pad(body, max(max_body, body_sha_length))
        adjusted_selector = sha_precompute_selector
        if sha_precompute_selector:
            clean, pmap = remove_soft_line_breaks(body_padded)
            sel = sha_precompute_selector if isinstance(sha_precompute_selector, str) else sha_precompute_selector.decode()
            adjusted_selector = get_adjusted_selector(body, sel, clean, pmap)
        pre, remaining, remaining_len = generate_partial_sha(body_padded, body_padded_len,
                                                             adjusted_selector, max_body)
        inputs["emailBodyLength"] = str(remaining_len)
        inputs["precomputedSHA"] = [str(b) for b in pre]
        inputs["bodyHashIndex"] = str(body_hash_index)
        inputs["emailBody"] = [str(b) for b in remaining]
        if remove_soft_line_breaks_flag:
            clean, _ = remove_soft_line_breaks(remaining)
            inputs["decodedEmailBodyIn"] = [str(b) for b in clean]
        if enable_body_masking:
            inputs["bodyMask"] = list(body_mask)
    return inputs
