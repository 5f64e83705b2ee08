"""Test-only helpers for the `.r1cs` path: a writer of the iden3 binary format, a pure-Python constraint
evaluator (the checker of the checker) and small constraint systems with their witnesses."""
import random

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


from zkwg.r1cs import write_r1cs  # noqa: E402,F401  (the product's writer; tests only add the evaluator)


def first_violation(constraints, w):
    ev = lambda d: sum(c * w[k] for k, c in d.items()) % P
    for i, (a, b, c) in enumerate(constraints):
        if (ev(a) * ev(b) - ev(c)) % P:
            return i
    return None


def num2bits_system(nbits, value):
    """circomlib Num2Bits(n): wires [1, in, out[0..n)]: out_i*(out_i-1)=0 and sum 2^i out_i = in."""
    cons = [({2 + i: 1}, {2 + i: 1, 0: -1}, {}) for i in range(nbits)]
    cons.append(({**{2 + i: 1 << i for i in range(nbits)}, 1: -1}, {0: 1}, {}))
    w = [1, value] + [(value >> i) & 1 for i in range(nbits)]
    return 2 + nbits, cons, w


def sigma_chain_system(x, rounds):
    """poseidon.circom Sigma repeated: per round in2 = in*in, in4 = in2*in2, out = in4*in (+ round constant)."""
    w = [1, x % P]
    cons = []
    cur = 1
    for r in range(rounds):
        v = w[cur]
        i2, i4, o = len(w), len(w) + 1, len(w) + 2
        w += [v * v % P, pow(v, 4, P), pow(v, 5, P)]
        cons += [({cur: 1}, {cur: 1}, {i2: 1}), ({i2: 1}, {i2: 1}, {i4: 1}), ({i4: 1}, {cur: 1}, {o: 1})]
        # next input = out + (r+1)  (a linear signal of its own, as Ark would be after O0)
        nxt = len(w)
        w.append((w[o] + r + 1) % P)
        cons.append(({o: 1, 0: r + 1, nxt: -1}, {0: 1}, {}))
        cur = nxt
    return len(w), cons, w


def random_system(seed, n_wires, m, max_terms=12):
    """A random satisfiable system: random sparse A, B, C' and one correcting term in C per constraint."""
    rng = random.Random(seed)
    w = [1] + [rng.choice([0, 1, rng.randrange(256), rng.randrange(P)]) for _ in range(n_wires - 1)]
    nz = [i for i, v in enumerate(w) if v]
    coefs = lambda: rng.choice([1, P - 1, 2, rng.randrange(1 << 16), rng.randrange(P), 1 << rng.randrange(253)])
    ev = lambda d: sum(c * w[k] for k, c in d.items()) % P
    cons = []
    for _ in range(m):
        a = {rng.randrange(n_wires): coefs() for _ in range(rng.randrange(0, max_terms))}
        b = {rng.randrange(n_wires): coefs() for _ in range(rng.randrange(0, max_terms))}
        c = {rng.randrange(n_wires): coefs() for _ in range(rng.randrange(0, max_terms))}
        k = rng.choice(nz)
        c.pop(k, None)
        c[k] = (ev(a) * ev(b) - ev(c)) * pow(w[k], P - 2, P) % P
        cons.append((a, b, c))
    return cons, w


def read_r1cs(data):
    """iden3 `.r1cs` reader in plain Python (independent of the product's C++ parser): -> (header dict, constraints) with
    constraints = [(A, B, C)], each a list of (wire, coefficient) pairs (integers mod P, as stored)."""
    import struct
    assert data[:4] == b"r1cs" and struct.unpack_from("<I", data, 4)[0] == 1
    nsec = struct.unpack_from("<I", data, 8)[0]
    pos = 12
    sec = {}
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", data, pos)
        sec[typ] = (pos + 12, size)
        pos += 12 + size
    o, _ = sec[1]
    n8 = struct.unpack_from("<I", data, o)[0]
    assert n8 == 32 and int.from_bytes(data[o + 4:o + 36], "little") == P
    n_wires, n_pub_out, n_pub_in, n_prv_in = struct.unpack_from("<IIII", data, o + 36)
    n_labels, m = struct.unpack_from("<QI", data, o + 52)
    hdr = {"n_wires": n_wires, "n_pub_out": n_pub_out, "n_pub_in": n_pub_in, "n_prv_in": n_prv_in, "n_labels": n_labels, "n_constraints": m}
    o, size = sec[2]
    mv = memoryview(data)
    cons = []
    fb = int.from_bytes
    for _ in range(m):
        lcs = []
        for _ in range(3):
            n = fb(mv[o:o + 4], "little")
            o += 4
            lc = []
            for _ in range(n):
                lc.append((fb(mv[o:o + 4], "little"), fb(mv[o + 4:o + 36], "little")))
                o += 36
            lcs.append(lc)
        cons.append(tuple(lcs))
    assert o == sec[2][0] + size
    return hdr, cons


def abc_digest(cons, w, montgomery=False):
    """SHA-256 of A.w | B.w | C.w (all A values, then B, then C; 32-byte little-endian) in Python integers."""
    import hashlib
    sc = (1 << 256) % P if montgomery else 1
    h = hashlib.sha256()
    for j in range(3):
        for t in cons:
            h.update((sum(cf * w[k] for k, cf in t[j]) * sc % P).to_bytes(32, "little"))
    return h.hexdigest()
