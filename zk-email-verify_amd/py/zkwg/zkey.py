"""snarkjs `.zkey` (groth16) reader / writer: the file `groth16.prove(zkey, wtns)` takes its bases from (reference call site:
packages/helpers/src/chunked-zkey.ts:80-84; the reference's helpers download it in chunks `circuit.zkeyb ... zkeyk`, :9-77).

No .zkey exists offline, so the layout below is restated from snarkjs' published code (src/zkey_utils.js,
src/zkey_new.js) [EXT] and is UNPINNED until a real file is read: the round trip write -> read is what the tests check.

    "zkey" | u32 version = 1 | u32 nSections = 10 | sections: u32 id, u64 size, payload
    1  u32 protocol (1 = groth16)
    2  u32 n8q, q | u32 n8r, r | u32 nVars, u32 nPublic, u32 domainSize | alpha1 (G1) beta1 (G1) beta2 (G2) gamma2 (G2) delta1 (G1) delta2 (G2)
    3  IC: nPublic + 1 G1 points           4  u32 nCoeffs, then (u32 matrix, u32 constraint, u32 signal, n8r value) -- A and B only; the
    5  A: nVars G1                             value is the coefficient times R^2 (so that a Montgomery product with a standard-form witness
    6  B1: nVars G1                            value lands in Montgomery form); rows nConstraints + s = wire s of A are the public rows
    7  B2: nVars G2                        8  C: nVars - nPublic - 1 G1 (private wires)
    9  H: domainSize G1                   10  contributions (64-byte hash, u32 count)
Points: affine, little-endian Montgomery limbs, x | y (G2: x.c0 | x.c1 | y.c0 | y.c1), all zeros = infinity -- exactly the layout
zkwg_msm_create* takes, so sections 5-9 are uploaded as they are.
"""
import struct

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def read_zkey(data):
    """-> dict: n_vars, n_public, domain_size, power, points alpha1 ... delta2 (bytes as stored), ic / a / b1 / b2 / c / h (bytes of the
    whole section), coeffs = [(matrix, constraint, signal, coefficient)] (standard-form integers)"""
    if data[:4] != b"zkey":
        raise ValueError("not a .zkey file")
    version, nsec = struct.unpack_from("<II", data, 4)
    if version != 1:
        raise ValueError(f".zkey version {version} is not supported")
    pos, sec = 12, {}
    for _ in range(nsec):
        if pos + 12 > len(data):
            raise ValueError(".zkey: truncated section table")
        sid, size = struct.unpack_from("<IQ", data, pos)
        if pos + 12 + size > len(data):
            raise ValueError(f".zkey: section {sid} runs past the end of the file")
        sec[sid] = (pos + 12, size)
        pos += 12 + size
    for need in (1, 2, 3, 5, 6, 7, 8, 9):
        if need not in sec:
            raise ValueError(f".zkey: section {need} is missing")
    if struct.unpack_from("<I", data, sec[1][0])[0] != 1:
        raise ValueError(".zkey: not a groth16 key")
    p = sec[2][0]
    n8q = struct.unpack_from("<I", data, p)[0]
    q = int.from_bytes(data[p + 4:p + 4 + n8q], "little")
    p += 4 + n8q
    n8r = struct.unpack_from("<I", data, p)[0]
    r = int.from_bytes(data[p + 4:p + 4 + n8r], "little")
    p += 4 + n8r
    if (n8q, q, n8r, r) != (32, Q, 32, R):
        raise ValueError(".zkey: not a BN254 key")
    n_vars, n_public, domain = struct.unpack_from("<III", data, p)
    p += 12
    out = {"n_vars": n_vars, "n_public": n_public, "domain_size": domain, "power": domain.bit_length() - 1}
    if domain & (domain - 1):
        raise ValueError(".zkey: the domain size is not a power of two")
    for name, size in (("alpha1", 64), ("beta1", 64), ("beta2", 128), ("gamma2", 128), ("delta1", 64), ("delta2", 128)):
        out[name] = bytes(data[p:p + size])
        p += size
    want = {3: 64 * (n_public + 1), 5: 64 * n_vars, 6: 64 * n_vars, 7: 128 * n_vars, 8: 64 * (n_vars - n_public - 1), 9: 64 * domain}
    for sid, name in ((3, "ic"), (5, "a"), (6, "b1"), (7, "b2"), (8, "c"), (9, "h")):
        o, size = sec[sid]
        if size != want[sid]:
            raise ValueError(f".zkey: section {sid} holds {size} bytes, expected {want[sid]}")
        out[name] = bytes(data[o:o + size])
    out["coeffs"] = []
    if 4 in sec:
        o, _ = sec[4]
        n = struct.unpack_from("<I", data, o)[0]
        o += 4
        r2inv = pow(pow(1 << 256, 2, R), -1, R)
        for _ in range(n):
            m, c, s = struct.unpack_from("<III", data, o)
            v = int.from_bytes(data[o + 12:o + 44], "little")
            out["coeffs"].append((m, c, s, v * r2inv % R))
            o += 44
    return out


def write_zkey(n_vars, n_public, domain_size, points, ic, a, b1, b2, c, h, coeffs=()):
    """points: dict alpha1, beta1, beta2, gamma2, delta1, delta2 (bytes as stored); ic ... h: bytes of the sections; coeffs as read_zkey gives"""
    assert len(ic) == 64 * (n_public + 1) and len(a) == 64 * n_vars and len(b1) == 64 * n_vars and len(b2) == 128 * n_vars
    assert len(c) == 64 * (n_vars - n_public - 1) and len(h) == 64 * domain_size
    r2 = pow(1 << 256, 2, R)
    hdr = struct.pack("<I", 32) + Q.to_bytes(32, "little") + struct.pack("<I", 32) + R.to_bytes(32, "little") + struct.pack("<III", n_vars, n_public, domain_size)
    for name, size in (("alpha1", 64), ("beta1", 64), ("beta2", 128), ("gamma2", 128), ("delta1", 64), ("delta2", 128)):
        assert len(points[name]) == size
        hdr += points[name]
    s4 = struct.pack("<I", len(coeffs)) + b"".join(struct.pack("<III", m, cc, s) + (v * r2 % R).to_bytes(32, "little") for m, cc, s, v in coeffs)
    secs = [(1, struct.pack("<I", 1)), (2, hdr), (3, ic), (4, s4), (5, a), (6, b1), (7, b2), (8, c), (9, h), (10, bytes(64) + struct.pack("<I", 0))]
    out = [b"zkey", struct.pack("<II", 1, len(secs))]
    for sid, payload in secs:
        out.append(struct.pack("<IQ", sid, len(payload)))
        out.append(payload)
    return b"".join(out)


def proving_key_from_zkey(device, data):
    """zkwg.prover.ProvingKey with sections 5-9 uploaded as they are"""
    import torch
    from .prover import ProvingKey
    z = read_zkey(data) if not isinstance(data, dict) else data
    dev = torch.device("cuda", device)
    up = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    return ProvingKey(device, z["n_vars"], z["n_public"], z["power"], up(z["a"]), up(z["b1"]), up(z["b2"]), up(z["c"]), up(z["h"]),
                      z["alpha1"], z["beta1"], z["beta2"], z["delta1"], z["delta2"])
