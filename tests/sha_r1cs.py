"""Test-only: the R1CS of the SHA-256 part of `Sha256Bytes(N)` over the wires of the kept layout, derived from
the circomlib templates the reference instantiates (lib/sha.circom:17-38, 89-203; circomlib
sha256/{sha256compression,sigmaplus,t1,t2,sigma,xor3,ch,maj}.circom and binsum.circom):
every kept signal of every Sha256compression block gets its defining constraint, written in terms of other
kept wires (aliases resolved symbolically), plus the byte decompositions.  Independent of the witness kernels
and of the oracle's evaluator -- used to run `checkConstraints` on real device witnesses."""
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617

K256 = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
IV256 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def lc_add(a, b, kb=1):
    out = dict(a)
    for w, c in b.items():
        v = (out.get(w, 0) + kb * c) % P
        if v:
            out[w] = v
        else:
            out.pop(w, None)
    return out


def const(v):
    return {0: v % P} if v % P else {}


def wire(s):
    return {s: 1}


def const_word(x):   # LSB-first bits of a 32-bit constant
    return [const((x >> k) & 1) for k in range(32)]


def rotr(w, r):
    return [w[(i + r) % 32] for i in range(32)]


def shr(w, r):
    return [w[i + r] if i + r < 32 else {} for i in range(32)]


class Builder:
    def __init__(self, slot_of):
        self.slot = slot_of
        self.cons = []

    def arr(self, name, n):
        s = self.slot[name + "[0]"]
        return [s + k for k in range(n)]

    def boolean(self, s):
        self.cons.append((wire(s), lc_add(wire(s), const(-1)), {}))

    def xor3(self, pre, a, b, c):
        out, mid = self.arr(pre + ".out", 32), self.arr(pre + ".mid", 32)
        for k in range(32):
            self.cons.append((b[k], c[k], wire(mid[k])))
            B = lc_add(lc_add(lc_add(const(1), b[k], -2), c[k], -2), wire(mid[k]), 4)
            C = lc_add(lc_add(lc_add(wire(out[k]), b[k], -1), c[k], -1), wire(mid[k]), 2)
            self.cons.append((a[k], B, C))
        return [wire(s) for s in out]

    def ch(self, pre, a, b, c):
        out = self.arr(pre + ".out", 32)
        for k in range(32):
            self.cons.append((a[k], lc_add(b[k], c[k], -1), lc_add(wire(out[k]), c[k], -1)))
        return [wire(s) for s in out]

    def maj(self, pre, a, b, c):
        out, mid = self.arr(pre + ".out", 32), self.arr(pre + ".mid", 32)
        for k in range(32):
            self.cons.append((b[k], c[k], wire(mid[k])))
            self.cons.append((a[k], lc_add(lc_add(b[k], c[k]), wire(mid[k]), -2), lc_add(wire(out[k]), wire(mid[k]), -1)))
        return [wire(s) for s in out]

    def binsum(self, pre, ins, nout):
        out = self.arr(pre + ".out", nout)
        lin = {}
        for op in ins:
            for k in range(32):
                lin = lc_add(lin, op[k], 1 << k)
        for k in range(nout):
            self.boolean(out[k])
            lin = lc_add(lin, wire(out[k]), -(1 << k))
        self.cons.append((lin, const(1), {}))
        return [wire(s) for s in out]

    def small_sigma(self, pre, x, ra, rb, rc):
        return self.xor3(pre + ".xor3", rotr(x, ra), rotr(x, rb), shr(x, rc))

    def big_sigma(self, pre, x, ra, rb, rc):
        return self.xor3(pre + ".xor3", rotr(x, ra), rotr(x, rb), rotr(x, rc))

    def compression(self, pre, hin, inp):
        """hin: 8 words of 32 LSB-first bit LCs; inp: 512 bit LCs (MSB-first per word).  Returns the 8 output
        words, LSB-first (= the next block's hin)."""
        w = [[inp[t * 32 + 31 - k] for k in range(32)] for t in range(16)]
        for t in range(16, 64):
            q = f"{pre}.sigmaPlus[{t - 16}]"
            s1 = self.small_sigma(q + ".sigma1", w[t - 2], 17, 19, 10)
            s0 = self.small_sigma(q + ".sigma0", w[t - 15], 7, 18, 3)
            w.append(self.binsum(q + ".sum", [s1, w[t - 7], s0, w[t - 16]], 34)[:32])
        a, b, c, d, e, f, g, h = hin
        # component order in the layout: all t1, then all t2, then suma, sume -- constraints may come in any order
        for t in range(64):
            q1, q2 = f"{pre}.t1[{t}]", f"{pre}.t2[{t}]"
            chv = self.ch(q1 + ".ch", e, f, g)
            bs1 = self.big_sigma(q1 + ".bigsigma1", e, 6, 11, 25)
            t1 = self.binsum(q1 + ".sum", [h, bs1, chv, const_word(K256[t]), w[t]], 35)[:32]
            bs0 = self.big_sigma(q2 + ".bigsigma0", a, 2, 13, 22)
            mj = self.maj(q2 + ".maj", a, b, c)
            t2 = self.binsum(q2 + ".sum", [bs0, mj], 33)[:32]
            ne = self.binsum(f"{pre}.sume[{t}]", [d, t1], 33)[:32]
            na = self.binsum(f"{pre}.suma[{t}]", [t1, t2], 33)[:32]
            h, g, f, e, d, c, b, a = g, f, e, ne, c, b, a, na
        fin = [a, b, c, d, e, f, g, h]
        return [self.binsum(f"{pre}.fsum[{j}]", [hin[j], fin[j]], 33)[:32] for j in range(8)]


def log2ceil(a):
    n, r = a - 1, 0
    while n > 0:
        r += 1
        n //= 2
    return r


def sha256_bytes_constraints(symbols, n_bytes, comp="main", data="main.paddedIn", pre=None, builder=None,
                             length="main.paddedInLength", tail=True):
    """Constraints of the Sha256Bytes(n_bytes) / Sha256BytesPartial instance `comp` whose paddedIn is the signal
    array `data` and whose paddedInLength is the wire `length`, over the kept wires: byte decompositions, every
    compression block, and (tail) Sha256General's block selection -- inBlockIndex, the length bound and the 256
    ItemAtIndex selectors (lib/sha.circom:105-129, 190-198; utils/array.circom:16-64).  pre: name of the 32-byte
    preHash array for Sha256BytesPartial (lib/sha.circom:47-80, 212-292), None for the IV.
    Returns (constraints, out) with out = the 256 output bits (MSB first) as linear combinations."""
    slot_of = {n: s for s, n in symbols}
    b = builder or Builder(slot_of)

    def byte_bits(comp_arr, src, i):
        o = b.arr(f"{comp}.{comp_arr}[{i}].out", 8)
        lin = lc_add({}, wire(slot_of[f"{src}[{i}]"]), -1)
        for k in range(8):
            b.boolean(o[k])
            lin = lc_add(lin, wire(o[k]), 1 << k)
        b.cons.append((lin, const(1), {}))
        return [wire(o[7 - j]) for j in range(8)]   # MSB first

    bits = []   # sha.paddedIn[8 i + j] = bytes[i].out[7 - j]
    for i in range(n_bytes):
        bits += byte_bits("bytes", data, i)
    if pre is None:
        hin = [const_word(x) for x in IV256]
    else:
        pb = []
        for i in range(32):
            pb += byte_bits("states", pre, i)
        hin = [[pb[32 * j + 31 - k] for k in range(32)] for j in range(8)]
    nblocks = n_bytes // 64
    block_out = []
    for blk in range(nblocks):
        hin = b.compression(f"{comp}.sha.sha256compression[{blk}]", hin, bits[512 * blk:512 * (blk + 1)])
        block_out.append(hin)
    if not tail:
        return b.cons, None
    sha = comp + ".sha"
    len_bits = lc_add({}, wire(slot_of[length]), 8)              # paddedInLength * 8
    ibi = wire(slot_of[sha + ".inBlockIndex"])
    b.cons.append((lc_add(len_bits, ibi, -512), const(1), {}))   # paddedInLength === inBlockIndex * 512
    nb = log2ceil(n_bytes * 8)
    # LessEqThan(nb)(len, maxBits) = LessThan(nb)(len, maxBits + 1): Num2Bits(nb+1)(len + 2^nb - maxBits - 1); === 1
    lt = b.arr(sha + ".bitLengthVerifier.lt.n2b.out", nb + 1)
    acc = lc_add(lc_add({}, len_bits, -1), const(n_bytes * 8 + 1 - (1 << nb)))
    for k, o in enumerate(lt):
        b.boolean(o)
        acc = lc_add(acc, wire(o), 1 << k)
    b.cons.append((acc, const(1), {}))
    b.cons.append((wire(lt[nb]), const(1), {}))                  # out = 1 - n2b.out[nb] === 1
    index = lc_add(ibi, const(-1))
    out = []
    for k in range(256):
        q = f"{sha}.arraySelectors[{k}]"
        tot_v, tot_i = {}, const(-1)
        for i in range(nblocks):
            o, inv = slot_of[f"{q}.eqs[{i}].isz.out"], slot_of[f"{q}.eqs[{i}].isz.inv"]
            x = lc_add(index, const(-i))                         # isz.in = in[1] - in[0] = index - i
            b.cons.append((x, wire(inv), lc_add(const(1), wire(o), -1)))
            b.cons.append((x, wire(o), {}))
            num = slot_of[f"{q}.calcTotalValue.nums[{i}]"]
            blk_bit = block_out[i][k // 32][31 - k % 32]          # compression out[k], MSB-first words
            b.cons.append((wire(o), blk_bit, wire(num)))
            tot_v = lc_add(tot_v, wire(num))
            tot_i = lc_add(tot_i, wire(o))
        b.cons.append((tot_i, const(1), {}))                     # calcTotalIndex.sum === 1
        out.append(tot_v)
    return b.cons, out


def sha256_main_constraints(symbols, n_bytes):
    """`component main { public [paddedIn, paddedInLength] } = Sha256Bytes(n)` (tests/test-circuits/sha-test.circom):
    everything above plus out[k] === the selected bit."""
    slot_of = {n: s for s, n in symbols}
    cons, out = sha256_bytes_constraints(symbols, n_bytes)
    for k in range(256):
        cons.append((lc_add(out[k], wire(slot_of[f"main.out[{k}]"]), -1), const(1), {}))
    return cons
