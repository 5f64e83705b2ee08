"""zkwg.r1cs -- the constraint system (R1CS) that matches the kept-v1 witness layout.

`groth16.prove(zkey, wtns)` (second half of `fullProve`, packages/helpers/src/chunked-zkey.ts:80) multiplies the
witness against a proving key built from an `.r1cs` whose wire numbering is the witness order.  The circom
compiler is what normally produces that file; it is not available offline, and its wire order is
not the kept-v1 order anyway.  This module derives the constraint system directly from the reference templates
-- packages/circuits/email-verifier.circom:42-174, lib/{sha,rsa,fp,bigint,base64}.circom,
utils/{array,regex,hash,bytes}.circom and the circomlib templates they instantiate -- over
the wires of the kept-v1 layout: every alias / linear signal is resolved into a linear combination of kept
wires, every quadratic definition and every `===` becomes one constraint.  BodyHashRegex is zkwg's own DFA
circuit, the only part that is not a restatement of a reference template.

    python -m zkwg.r1cs --max-header 1024 --max-body 1536 -o email-verifier.r1cs --sym email-verifier.sym

writes the iden3 `.r1cs` (+ the `.sym`) of EmailVerifier(maxHeader, maxBody, 121, 17, 0, 0, 0, 0); together with
a `.wtns` from zkwg they form a consistent triple for `snarkjs zkey new` / `groth16 prove`.  The template flags
ignoreBodyHashCheck / enableHeaderMasking / enableBodyMasking / removeSoftLineBreaks are supported.  Validation: tests/test_r1cs.py (every wire constrained, oracle and device
witnesses satisfy every constraint, `zk_r1cs_check` agrees with a pure-Python evaluator on corrupted witnesses).
"""
import json
import os
import struct

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


# ------------------------------------------------------------------ RSAVerifier65537(121, 17)
N_, K_ = 121, 17


def lc_scale(a, k):
    return {w: c * k % P for w, c in a.items() if c * k % P}


class RsaBuilder:
    def __init__(self, slot_of, prefix, n=N_, k=K_):
        self.slot = slot_of
        self.cons = []
        self.pre = prefix
        self._interp = None
        self.n, self.k = n, k          # chunk bits / chunks of the big integers (lib/fp.circom, lib/bigint.circom are generic)

    def s(self, name):
        return self.slot[name]

    def arr(self, name, n):
        s0 = self.slot[name + "[0]"]
        return [s0 + i for i in range(n)]

    def lin(self, lc):            # lc === 0
        self.cons.append((lc, const(1), {}))

    def num2bits(self, pre, x, n):
        out = self.arr(pre + ".out", n)
        acc = lc_scale(x, -1)
        for i, o in enumerate(out):
            self.cons.append((wire(o), lc_add(wire(o), const(-1)), {}))
            acc = lc_add(acc, wire(o), 1 << i)
        self.lin(acc)
        return [wire(o) for o in out]

    def iszero(self, pre, x):
        o, inv = self.s(pre + ".out"), self.s(pre + ".inv")
        self.cons.append((x, wire(inv), lc_add(const(1), wire(o), -1)))   # out = -in*inv + 1
        self.cons.append((x, wire(o), {}))                               # in*out = 0
        return wire(o)

    def less_than(self, pre, n, a, b):
        bits = self.num2bits(pre + ".n2b", lc_add(lc_add(a, const(1 << n)), b, -1), n + 1)
        return lc_add(const(1), bits[n], -1)

    def and_(self, pre, a, b):
        o = self.s(pre + ".out")
        self.cons.append((a, b, wire(o)))
        return wire(o)

    def or_(self, pre, a, b):
        o = self.s(pre + ".out")
        self.cons.append((a, b, lc_add(lc_add(a, b), wire(o), -1)))
        return wire(o)

    def big_less_than(self, pre, a, b):
        k = self.k
        lt = [self.less_than(f"{pre}.lt[{i}]", self.n, a[i], b[i]) for i in range(k)]
        eq = [self.iszero(f"{pre}.eq[{i}].isz", lc_add(b[i], a[i], -1)) for i in range(k)]
        ors, eq_ands = [None] * (k - 1), [None] * (k - 1)
        for i in range(k - 2, -1, -1):
            if i == k - 2:
                an = self.and_(f"{pre}.ands[{i}]", eq[k - 1], lt[k - 2])
                eq_ands[i] = self.and_(f"{pre}.eq_ands[{i}]", eq[k - 1], eq[k - 2])
                ors[i] = self.or_(f"{pre}.ors[{i}]", lt[k - 1], an)
            else:
                an = self.and_(f"{pre}.ands[{i}]", eq_ands[i + 1], lt[i])
                eq_ands[i] = self.and_(f"{pre}.eq_ands[{i}]", eq_ands[i + 1], eq[i])
                ors[i] = self.or_(f"{pre}.ors[{i}]", ors[i + 1], an)
        return ors[0]

    @staticmethod
    def poly(limbs, x):
        acc = {}
        for i, l in enumerate(limbs):
            acc = lc_add(acc, l, pow(x, i, P))
        return acc

    def interp_matrix(self, n):
        # t = poly_interp(v) (lib/bigint-func.circom:65-103) is linear in v: coefficient i of the polynomial through
        # (x, v[x]), x = 0..n-1, is sum_x L_x[i] * v[x] with L_x the Lagrange basis polynomial of node x
        if self._interp is None:
            T = [[0] * n for _ in range(n)]
            for x in range(n):
                num = [1]                      # prod_{y != x} (X - y), coefficients low to high
                den = 1
                for y in range(n):
                    if y == x:
                        continue
                    nxt = [0] * (len(num) + 1)
                    for d, cf in enumerate(num):
                        nxt[d] = (nxt[d] - y * cf) % P
                        nxt[d + 1] = (nxt[d + 1] + cf) % P
                    num = nxt
                    den = den * (x - y) % P
                inv = pow(den, P - 2, P)
                for i in range(n):
                    T[i][x] = num[i] * inv % P
            self._interp = T
        return self._interp

    def fp_mul(self, pre, a, b, p):
        k, m, N = self.k, 2 * self.k - 1, self.n
        cb = N + k.bit_length() + 5    # CheckCarryToZero: m + EPSILON - n bits per carry, m = 2n + log_ceil(k) + 2 (131 for (121, 17))
        v_ab = self.arr(pre + ".v_ab", m)
        q = [wire(s) for s in self.arr(pre + ".q", k)]
        r = [wire(s) for s in self.arr(pre + ".r", k)]
        v_pq_r = self.arr(pre + ".v_pq_r", m)
        for x in range(m):
            self.cons.append((self.poly(a, x), self.poly(b, x), wire(v_ab[x])))
        for i in range(k):
            self.num2bits(f"{pre}.q_range_check[{i}]", q[i], N)
        for i in range(k):
            self.num2bits(f"{pre}.r_range_check[{i}]", r[i], N)
        self.lin(lc_add(self.big_less_than(pre + ".r_p_lt_check", r, p), const(-1)))
        for x in range(m):
            self.cons.append((self.poly(p, x), self.poly(q, x), lc_add(wire(v_pq_r[x]), self.poly(r, x), -1)))
        v_t = [lc_add(wire(v_ab[x]), wire(v_pq_r[x]), -1) for x in range(m)]
        T = self.interp_matrix(m)
        t = []
        for i in range(m):
            acc = {}
            for x in range(m):
                acc = lc_add(acc, v_t[x], T[i][x])
            t.append(acc)
        # CheckCarryToZero(n, 2n + log_ceil(k) + 2, 2k-1)
        carry = self.arr(pre + ".tCheck.carry", m)
        for i in range(m - 1):
            lhs = t[i] if i == 0 else lc_add(t[i], wire(carry[i - 1]))
            self.lin(lc_add(lhs, wire(carry[i]), -(1 << N)))
            self.num2bits(f"{pre}.tCheck.carryRangeChecks[{i}]", lc_add(wire(carry[i]), const(1 << (cb - 1))), cb)
        self.lin(lc_add(t[m - 1], wire(carry[m - 2])))
        return r

    def rsa_pad(self, pre, modulus, message):
        n, k = N_, K_
        base_len, msg_len = 408, 256
        mod_bits, msg_bits = [], []
        for i in range(k):
            mod_bits += self.num2bits(f"{pre}.modulusN2B[{i}]", modulus[i], n)
        for i in range(k):
            msg_bits += self.num2bits(f"{pre}.messageN2B[{i}]", message[i], n)
        for i in range(msg_len, n * k):
            self.lin(msg_bits[i])
        padded = [None] * (n * k)
        for i in range(msg_len):
            padded[i] = msg_bits[i]
        for i in range(base_len, base_len + 8):
            padded[i] = {}
        for i in range(msg_len, base_len):
            padded[i] = const((0x3031300D060960864801650304020105000420 >> (i - msg_len)) & 1)
        prefix = {}
        for i in range(n * k - 1, base_len + 8 - 1, -1):
            if i + 8 < n * k:
                prefix = lc_add(prefix, mod_bits[i + 8])
                if i % 8 == 0:
                    idx = (i - (base_len + 8)) // 8
                    z = self.iszero(f"{pre}.modulusZero[{idx}]", prefix)
                    padded[i] = lc_add(const(1), z, -1)
                else:
                    padded[i] = padded[i + 1]
            else:
                padded[i] = {}
        for i in range(base_len + 8, base_len + 8 + 65):
            self.lin(lc_add(padded[i], const(-1)))
        out = []
        for i in range(k):
            acc = {}
            for j in range(n):
                acc = lc_add(acc, padded[i * n + j], 1 << j)
            out.append(acc)
        return out

    def rsa_verifier(self, message, signature, modulus):
        pre = self.pre
        padded = self.rsa_pad(pre + ".padder", modulus, message)
        for i in range(K_):
            self.num2bits(f"{pre}.signatureRangeCheck[{i}]", signature[i], N_)
        self.lin(lc_add(self.big_less_than(pre + ".bigLessThan", signature, modulus), const(-1)))
        cur = signature
        for i in range(16):
            cur = self.fp_mul(f"{pre}.bigPow.doublers[{i}]", cur, cur, modulus)
        out = self.fp_mul(f"{pre}.bigPow.adder", signature, cur, modulus)
        for i in range(K_):
            self.lin(lc_add(out[i], padded[i], -1))
        return self.cons


def rsa_main_constraints(symbols):
    """`component main { public [modulus] } = RSAVerifier65537(121, 17)` (tests/test-circuits/rsa-test.circom)."""
    slot_of = {n: s for s, n in symbols}
    b = RsaBuilder(slot_of, "main")
    arr = lambda nm: [wire(slot_of[f"main.{nm}[{i}]"]) for i in range(K_)]
    return b.rsa_verifier(arr("message"), arr("signature"), arr("modulus"))


def fp_mul_main_constraints(symbols, n, k):
    """`component main = FpMul(n, k)` (tests/test-circuits/fp-mul-test.circom:5: FpMul(2, 4)); out[i] <== r[i]."""
    slot_of = {nm: s for s, nm in symbols}
    b = RsaBuilder(slot_of, "main", n, k)
    arr = lambda nm: [wire(slot_of[f"main.{nm}[{i}]"]) for i in range(k)]
    r = b.fp_mul("main", arr("a"), arr("b"), arr("p"))
    out = arr("out")
    for i in range(k):
        b.lin(lc_add(out[i], r[i], -1))
    return b.cons


# ------------------------------------------------------------------ EmailVerifier
def one_minus(x):
    return lc_add(const(1), x, -1)


def multi_or(g, pre, ins):
    """zk-regex MultiOR: out = 1 - IsZero(sum).out (names <pre>.is_zero.out / .inv)"""
    acc = {}
    for x in ins:
        acc = lc_add(acc, x)
    return one_minus(g.iszero(pre + ".is_zero", acc))


def body_hash_regex_v1(g, pre, header_wires):
    T = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "data", "bh_dfa.json")))
    S, ACC = T["n_states"], T["accept"]
    prims, classes, trans, public = T["prims"], T["classes"], T["transitions"], T["public"]
    msg_bytes = len(header_wires)
    nb = msg_bytes + 1
    in_ = [const(255)] + list(header_wires)
    eq_idx = [k for k, p in enumerate(prims) if p[0] == "eq"]
    rg_idx = [k for k, p in enumerate(prims) if p[0] == "range"]
    multi_cls = [k for k, c in enumerate(classes) if len(c["members"]) > 1]
    incoming = {d: [] for d in range(S)}
    for t_i, (f, to, cid) in enumerate(trans):
        incoming[to].append(t_i)
    tmp_multi = [d for d in range(1, S) if len([t for t in incoming[d] if trans[t][0] != 0]) > 1]
    st_multi = [d for d in range(1, S) if [t for t in incoming[d] if trans[t][0] == 0] and [t for t in incoming[d] if trans[t][0] != 0]]
    states = [[const(1)] + [{} for _ in range(S - 1)] for _ in range(nb + 1)]
    fze = [None] * (nb + 1)
    for i in range(nb):
        b = in_[i]
        pv = {}
        for n_, k in enumerate(eq_idx):
            pv[k] = g.iszero(f"{pre}.eq[{n_}][{i}].isz", lc_add(const(prims[k][1]), b, -1))
        for n_, k in enumerate(rg_idx):
            lo, hi = prims[k][1], prims[k][2]
            a = g.less_than(f"{pre}.lt[{2 * n_}][{i}]", 8, const(lo - 1), b)
            bb = g.less_than(f"{pre}.lt[{2 * n_ + 1}][{i}]", 8, b, const(hi + 1))
            pv[k] = g.and_(f"{pre}.and_rng[{n_}][{i}]", a, bb)
        cv = []
        for k, cdef in enumerate(classes):
            if len(cdef["members"]) > 1:
                v = multi_or(g, f"{pre}.cls_or[{multi_cls.index(k)}][{i}]", [pv[m] for m in cdef["members"]])
            else:
                v = pv[cdef["members"][0]]
            cv.append(one_minus(v) if cdef["neg"] else v)
        and_t = [None] * len(trans)
        for t_i, (f, to, cid) in enumerate(trans):
            if f != 0:
                and_t[t_i] = g.and_(f"{pre}.and[{t_i}][{i}]", states[i][f], cv[cid])
        tmps = [{} for _ in range(S)]
        for d in range(1, S):
            inc = [t for t in incoming[d] if trans[t][0] != 0]
            if len(inc) == 1:
                tmps[d] = and_t[inc[0]]
            elif len(inc) > 1:
                tmps[d] = multi_or(g, f"{pre}.tmp_or[{tmp_multi.index(d)}][{i}]", [and_t[t] for t in inc])
        acc = {}
        for d in range(1, S):
            acc = lc_add(acc, tmps[d])
        fze[i] = g.iszero(f"{pre}.fze[{i}].is_zero", acc)          # MultiNOR
        for t_i, (f, to, cid) in enumerate(trans):
            if f == 0:
                and_t[t_i] = g.and_(f"{pre}.and[{t_i}][{i}]", fze[i], cv[cid])
        for d in range(1, S):
            z = [t for t in incoming[d] if trans[t][0] == 0]
            if not z:
                states[i + 1][d] = tmps[d]
            elif d in st_multi:
                states[i + 1][d] = multi_or(g, f"{pre}.st_or[{st_multi.index(d)}][{i}]", [tmps[d], and_t[z[0]]])
            else:
                states[i + 1][d] = and_t[z[0]]
    fze[nb] = {}
    out = multi_or(g, f"{pre}.is_accepted", [states[j][ACC] for j in range(nb + 1)])
    live_c1, live_t = g.arr(pre + ".live_c1", nb), g.arr(pre + ".live_t", nb)
    lv = [{} for _ in range(nb + 2)]
    for j in range(nb, 0, -1):
        if j < nb:
            g.cons.append((lv[j + 1], one_minus(fze[j]), wire(live_c1[j - 1])))
        else:
            g.lin(wire(live_c1[j - 1]))
        g.cons.append((one_minus(states[j][ACC]), wire(live_c1[j - 1]), wire(live_t[j - 1])))
        lv[j] = lc_add(states[j][ACC], wire(live_t[j - 1]))
    prev = g.arr(pre + ".prev_states0", len(public) * msg_bytes)
    isrev = g.arr(pre + ".is_reveal0", msg_bytes)
    reveal = g.arr(pre + ".reveal0", msg_bytes)
    for i in range(msg_bytes):
        pvals = []
        for k, (s_, d_) in enumerate(public):
            w_ = prev[k * msg_bytes + i]
            g.cons.append((states[i + 1][s_], states[i + 2][d_], wire(w_)))
            pvals.append(wire(w_))
        so = multi_or(g, f"{pre}.substr_or[{i}]", pvals)
        g.cons.append((so, lv[i + 2], wire(isrev[i])))
        g.cons.append((in_[i + 1], wire(isrev[i]), wire(reveal[i])))
    return out, [wire(s) for s in reveal]


def assert_zero_padding(g, pre, in_wires, start):
    n = len(in_wires)
    bl = log2ceil(n)
    for i in range(n):
        lt = g.less_than(f"{pre}.lessThans[{i}]", bl, lc_add(start, const(-1)), const(i))
        g.cons.append((lt, in_wires[i], {}))


def select_regex_reveal(g, pre, in_, start, max_reveal):
    n = len(in_)
    bl = log2ceil(n + max_reveal - 1)
    for i in range(n):
        is_start = g.iszero(f"{pre}.anon_IsEqual[{i}].isz", lc_add(start, const(-i)))
        is_zero = g.iszero(f"{pre}.anon_IsZero[{i}]", in_[i])
        is_prev_zero = g.iszero(f"{pre}.anon_IsPrevZero[{i}]", in_[i - 1]) if i else const(1)
        # GreaterThan(bl)([i, startIndex + maxRevealLen - 1]) = LessThan(bl)(startIndex + maxRevealLen - 1, i)
        is_above = g.less_than(f"{pre}.anon_GreaterThan[{i}].lt", bl, lc_add(start, const(max_reveal - 1)), const(i))
        g.cons.append((is_start, is_zero, {}))
        g.cons.append((is_start, one_minus(is_prev_zero), {}))
        g.cons.append((is_above, one_minus(is_zero), {}))
    # VarShiftLeft(n, max_reveal)
    vs = pre + ".anon_VarShiftLeft"
    blv = log2ceil(n)
    tmp = g.arr(vs + ".tmp", blv * n)
    prev = list(in_)
    layers = []
    for j in range(blv):
        cur = [wire(tmp[j * n + i]) for i in range(n)]
        layers.append((prev, cur))
        prev = cur
    bits = g.num2bits(vs + ".n2b", start, blv)
    for j, (pv, cur) in enumerate(layers):
        for i in range(n):
            off = (i + (1 << j)) % n
            g.cons.append((bits[j], lc_add(pv[off], pv[i], -1), lc_add(cur[i], pv[i], -1)))
    return prev[:max_reveal]


def base64_lookup(g, pre, x):
    le_Z = g.less_than(pre + ".le_Z", 8, x, const(91))
    ge_A = g.less_than(pre + ".ge_A.lt", 8, const(64), x)
    w = lambda nm: wire(g.s(pre + "." + nm))
    g.cons.append((ge_A, le_Z, w("range_AZ")))
    g.cons.append((w("range_AZ"), lc_add(x, const(-65)), w("sum_AZ")))
    le_z = g.less_than(pre + ".le_z", 8, x, const(123))
    ge_a = g.less_than(pre + ".ge_a.lt", 8, const(96), x)
    g.cons.append((ge_a, le_z, w("range_az")))
    g.cons.append((w("range_az"), lc_add(x, const(-71)), lc_add(w("sum_az"), w("sum_AZ"), -1)))
    le_9 = g.less_than(pre + ".le_9", 8, x, const(58))
    ge_0 = g.less_than(pre + ".ge_0.lt", 8, const(47), x)
    g.cons.append((ge_0, le_9, w("range_09")))
    g.cons.append((w("range_09"), lc_add(x, const(4)), lc_add(w("sum_09"), w("sum_az"), -1)))
    eqp = g.iszero(pre + ".equal_plus", lc_add(x, const(-43)))
    g.cons.append((eqp, lc_add(x, const(19)), lc_add(w("sum_plus"), w("sum_09"), -1)))
    eqs = g.iszero(pre + ".equal_slash", lc_add(x, const(-47)))
    g.cons.append((eqs, lc_add(x, const(16)), lc_add(w("sum_slash"), w("sum_plus"), -1)))
    eqe = g.iszero(pre + ".equal_eqsign", lc_add(x, const(-61)))
    tot = const(-1)
    for t in (w("range_AZ"), w("range_az"), w("range_09"), eqp, eqs, eqe):
        tot = lc_add(tot, t)
    g.lin(tot)
    return w("sum_slash")


def base64_decode(g, pre, chars, byte_length=32):
    out = [None] * byte_length
    idx = 0
    for gi in range(len(chars) // 4):
        bi = []
        for j in range(4):
            v = base64_lookup(g, f"{pre}.translate[{gi}][{j}]", chars[4 * gi + j])
            bi.append(g.num2bits(f"{pre}.bitsIn[{gi}][{j}]", v, 6))
        bo0 = [bi[1][4], bi[1][5]] + [bi[0][j] for j in range(6)]
        bo1 = [bi[2][j + 2] for j in range(4)] + [bi[1][j] for j in range(4)]
        bo2 = [bi[3][j] for j in range(6)] + [bi[2][0], bi[2][1]]
        for j, bo in enumerate((bo0, bo1, bo2)):
            if idx + j < byte_length:
                acc = {}
                for k in range(8):
                    acc = lc_add(acc, bo[k], 1 << k)
                out[idx + j] = acc
        idx += 3
    return out


def email_verifier_constraints(symbols, N, M, enable_header_masking=0, enable_body_masking=0, remove_soft_line_breaks_flag=0,
                               ignore_body_hash_check=0):
    slot_of = {n: s for s, n in symbols}
    body_on = not ignore_body_hash_check
    b = Builder(slot_of)
    _, sha = sha256_bytes_constraints(symbols, N, "main.anon_Sha256Bytes", "main.emailHeader", builder=b,
                                               length="main.emailHeaderLength")
    if body_on:
        _, body_sha = sha256_bytes_constraints(symbols, M, "main.anon_Sha256BytesPartial", "main.emailBody",
                                               pre="main.precomputedSHA", builder=b, length="main.emailBodyLength")
    cons = b.cons
    # PackBits(256, 128) (utils/bytes.circom:194-210 via email-verifier.circom:68-71): chunk 0 -> shaHi, chunk 1 ->
    # shaLo, each chunk big-endian: out[i] = sum_j in[128 i + j] * 2^(127 - j)
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
    g = RsaBuilder(slot_of, "main.rsaVerifier")
    g.cons = cons
    arr = lambda nm, k: [wire(slot_of[f"main.{nm}[{i}]"]) for i in range(k)]
    g.rsa_verifier(message, arr("signature", 17), arr("pubkey", 17))
    header = arr("emailHeader", N)
    hlen = wire(slot_of["main.emailHeaderLength"])
    g.num2bits("main.n2bHeaderLength", hlen, log2ceil(N))
    assert_zero_padding(g, "main.anon_AssertZeroPadding_header", header, hlen)
    if not body_on:      # email-verifier.circom:108: everything below sits inside `if (ignoreBodyHashCheck != 1)`
        if enable_header_masking:
            _byte_mask(g, slot_of, header, arr("headerMask", N), "maskedHeader", "byteMask_header")
        cons += poseidon9_constraints(symbols)
        return cons
    body = arr("emailBody", M)
    blen = wire(slot_of["main.emailBodyLength"])
    bhi = wire(slot_of["main.bodyHashIndex"])
    g.num2bits("main.n2bBodyLength", blen, log2ceil(M))
    assert_zero_padding(g, "main.anon_AssertZeroPadding_body", body, blen)
    match, reveal = body_hash_regex_v1(g, "main.anon_BodyHashRegex", header)
    g.lin(lc_add(match, const(-1)))                                   # bhRegexMatch === 1
    chars = select_regex_reveal(g, "main.anon_SelectRegexReveal", reveal, bhi, 44)
    hbh = base64_decode(g, "main.anon_Base64Decode", chars)
    for i in range(32):                                               # computedBodyHashInts[i].out === headerBodyHash[i]
        acc = lc_add({}, hbh[i], -1)
        for j in range(8):
            acc = lc_add(acc, body_sha[8 * i + j], 1 << (7 - j))
        g.lin(acc)
    if remove_soft_line_breaks_flag:       # email-verifier.circom:148-156
        valid = remove_soft_line_breaks(g, "main.qpEncodingChecker", body, arr("decodedEmailBodyIn", M))
        g.lin(lc_add(valid, const(-1)))
    if enable_header_masking:
        _byte_mask(g, slot_of, header, arr("headerMask", N), "maskedHeader", "byteMask_header")
    if enable_body_masking:
        _byte_mask(g, slot_of, body, arr("bodyMask", M), "maskedBody", "byteMask_body")
    cons += poseidon9_constraints(symbols)
    return cons


def _byte_mask(g, slot_of, data, mask, outp, comp):
    """ByteMask (utils/bytes.circom:173-185): AssertBit(mask[i]); out[i] <== in[i] * mask[i]; the main output array
    aliases the component's outputs"""
    for i in range(len(data)):
        g.cons.append((mask[i], lc_add(mask[i], const(-1)), {}))
        o = wire(slot_of[f"main.{comp}.out[{i}]"])
        g.cons.append((data[i], mask[i], o))
        g.lin(lc_add(o, wire(slot_of[f"main.{outp}[{i}]"]), -1))


# ------------------------------------------------------------------ Poseidon(9) block, `.r1cs` writer, CLI
N_ROUNDS_P = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]   # t = 2..17, R_F = 8


def poseidon_constants(t):
    """(C, M) of circomlib's Poseidon for state width t, regenerated by the published procedure (Grain LFSR seeded
    with field = prime, sbox = x^5, n = 254, t, R_F = 8, R_P; round constants by rejection sampling; MDS = Cauchy
    matrix 1 / (x_i + y_j) from the next 2t values) -- the same procedure as csrc/zkwg_build.h."""
    r_f, r_p = 8, N_ROUNDS_P[t - 2]
    bits = []
    for v, w_ in ((1, 2), (0, 4), (254, 12), (t, 12), (r_f, 10), (r_p, 10)):
        bits.extend(int(c) for c in bin(v)[2:].zfill(w_))
    state = bits + [1] * 30

    def step():
        nb = state[62] ^ state[51] ^ state[38] ^ state[23] ^ state[13] ^ state[0]
        state.pop(0)
        state.append(nb)
        return nb

    for _ in range(160):
        step()

    def next_bit():
        nb = step()
        while nb == 0:
            step()
            nb = step()
        return step()

    def next_254():
        v = 0
        for _ in range(254):
            v = (v << 1) | next_bit()
        return v

    C = []
    while len(C) < (r_f + r_p) * t:
        v = next_254()
        if v < P:
            C.append(v)
    xy = [next_254() % P for _ in range(2 * t)]
    M = [[pow(xy[i] + xy[t + j], P - 2, P) for j in range(t)] for i in range(t)]
    return C, M


def poseidon_constraints(cons, slot, pre, inputs, t=None):
    """circomlib Poseidon(len(inputs)) instance `pre` (its PoseidonEx is `pre`.pEx): S-box constraints through
    linear combinations of earlier wires, textbook rounds; returns out as a linear combination."""
    t = t or len(inputs) + 1
    rp = N_ROUNDS_P[t - 2]
    Cc, M = poseidon_constants(t)
    state = [{}] + list(inputs)

    def sbox(lc, name):
        o, i2, i4 = slot[name + ".out"], slot[name + ".in2"], slot[name + ".in4"]
        cons.extend([(lc, lc, {i2: 1}), ({i2: 1}, {i2: 1}, {i4: 1}), ({i4: 1}, lc, {o: 1})])
        return {o: 1}

    fr = 0
    for r in range(8 + rp):
        state = [lc_add(state[j], const(Cc[r * t + j])) for j in range(t)]
        if r < 4 or r >= 4 + rp:
            state = [sbox(state[j], f"{pre}.pEx.sigmaF[{fr}][{j}]") for j in range(t)]
            fr += 1
        else:
            state[0] = sbox(state[0], f"{pre}.pEx.sigmaP[{r - 4}]")
        new = []
        for i in range(t):
            acc = {}
            for j in range(t):
                acc = lc_add(acc, state[j], M[i][j])
            new.append(acc)
        state = new
    return state[0]


_pos_cache = {}
_poseidon_constants_raw = poseidon_constants


def poseidon_constants(t):   # noqa: F811  (memoised)
    if t not in _pos_cache:
        _pos_cache[t] = _poseidon_constants_raw(t)
    return _pos_cache[t]


def poseidon9_constraints(symbols):
    """The PoseidonLarge(121,17) -> Poseidon(9) block of EmailVerifier (utils/hash.circom:15-39): the last
    constraint ties the result to main.pubkeyHash."""
    slot = {n: s for s, n in symbols}
    pk = [slot[f"main.pubkey[{i}]"] for i in range(17)]
    inputs = [({pk[2 * i]: 1, pk[2 * i + 1]: 1 << 121} if i < 8 else {pk[16]: 1}) for i in range(9)]
    cons = []
    out = poseidon_constraints(cons, slot, "main.anon_PoseidonLarge.anon_Poseidon", inputs)
    cons.append((lc_add(out, wire(slot["main.pubkeyHash"]), -1), const(1), {}))
    return cons


def remove_soft_line_breaks(g, pre, enc, dec):
    """helpers/remove-soft-line-breaks.circom:14-126 as the component `pre`; enc / dec: lists of byte wires.
    Returns isValid as a linear combination."""
    M = len(enc)
    slot = g.slot
    # r = PoseidonModular(2M)(encoded || decoded) (utils/hash.circom:49-82)
    data = list(enc) + list(dec)
    r = None
    for c in range(len(data) // 16):
        h = poseidon_constraints(g.cons, slot, f"{pre}.rHasher.anon_Poseidon_chunk[{c}]", data[16 * c:16 * c + 16])
        r = h if c == 0 else poseidon_constraints(g.cons, slot, f"{pre}.rHasher.anon_Poseidon_merge[{c}]", [r, h])
    is_eq = [g.iszero(f"{pre}.anon_IsEqual_eq[{i}].isz", lc_add(const(61), enc[i], -1)) for i in range(M)]
    is_cr = [g.iszero(f"{pre}.anon_IsEqual_cr[{i}].isz", lc_add(const(13), enc[i + 1], -1)) for i in range(M - 1)] + [{}]
    is_lf = [g.iszero(f"{pre}.anon_IsEqual_lf[{i}].isz", lc_add(const(10), enc[i + 2], -1)) for i in range(M - 2)] + [{}, {}]
    tsb, sb = g.arr(pre + ".tempSoftBreak", M - 2), g.arr(pre + ".isSoftBreak", M - 2)
    is_sb = []
    for i in range(M - 2):
        g.cons.append((is_eq[i], is_cr[i], wire(tsb[i])))
        g.cons.append((wire(tsb[i]), is_lf[i], wire(sb[i])))
        is_sb.append(wire(sb[i]))
    is_sb += [{}, {}]
    sz = []
    for i in range(M):
        v = is_sb[i] if i < M - 1 else {}
        if i == M - 1:
            v = lc_add(is_sb[i - 1], is_sb[i - 2])
        else:
            if i >= 1:
                v = lc_add(v, is_sb[i - 1])
            if i >= 2:
                v = lc_add(v, is_sb[i - 2])
        sz.append(v)
    proc = g.arr(pre + ".processed", M)
    for i in range(M):
        g.cons.append((one_minus(sz[i]), enc[i], wire(proc[i])))
    # muxEnc[i] = Mux1 -> MultiMux1(1): out = (c[1] - c[0]) * s + c[0]
    r_enc = []
    for i in range(M):
        if i == 0:
            c0, c1 = r, const(1)
        else:
            c0w = slot[f"{pre}.muxEnc[{i}].c[0]"]
            g.cons.append((r_enc[i - 1], r, wire(c0w)))
            c0, c1 = wire(c0w), r_enc[i - 1]
        o = slot[f"{pre}.muxEnc[{i}].mux.out[0]"]
        g.cons.append((lc_add(c1, c0, -1), sz[i], lc_add(wire(o), c0, -1)))
        r_enc.append(wire(o))
    r_dec = [r]
    for i in range(1, M):
        w_ = slot[f"{pre}.rDec[{i}]"]
        g.cons.append((r_dec[i - 1], r, wire(w_)))
        r_dec.append(wire(w_))
    s_enc, s_dec = g.arr(pre + ".sumEnc", M), g.arr(pre + ".sumDec", M)
    for i in range(M):
        prev_e = wire(s_enc[i - 1]) if i else {}
        prev_d = wire(s_dec[i - 1]) if i else {}
        g.cons.append((r_enc[i], wire(proc[i]), lc_add(wire(s_enc[i]), prev_e, -1)))
        g.cons.append((r_dec[i], dec[i], lc_add(wire(s_dec[i]), prev_d, -1)))
    return g.iszero(pre + ".anon_IsEqual_final.isz", lc_add(wire(s_dec[M - 1]), wire(s_enc[M - 1]), -1))


def write_r1cs(n_wires, constraints, n_pub_out=0, n_pub_in=0, n_prv_in=0, header_last=False):
    """constraints: list of (A, B, C), each a dict wire -> coefficient (ints mod P)."""
    def lc(d):
        items = sorted((w, c % P) for w, c in d.items() if c % P)
        return struct.pack("<I", len(items)) + b"".join(struct.pack("<I", w) + c.to_bytes(32, "little") for w, c in items)
    hdr = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIIIQI", n_wires, n_pub_out, n_pub_in, n_prv_in,
                                                                       n_wires, len(constraints))
    cons = b"".join(lc(a) + lc(b) + lc(c) for a, b, c in constraints)
    w2l = b"".join(struct.pack("<Q", i) for i in range(n_wires))
    secs = [(1, hdr), (2, cons), (3, w2l)]
    if header_last:
        secs = [secs[1], secs[2], secs[0]]
    out = b"r1cs" + struct.pack("<II", 1, len(secs))
    for t, d in secs:
        out += struct.pack("<IQ", t, len(d)) + d
    return out




def append_public_rows(constraints, n_public):
    """the system a snarkjs zkey is built over: zkey_new.js appends nPublic + 1 rows to the A matrix, row nConstraints + s =
    the wire s alone (s = 0 .. nPublic; B and C empty there), which keeps the public wires' polynomials independent.  Attached
    like this (zkwg.Circuit.attach_r1cs), zkwg_expand_abc_device writes the A.w | B.w | C.w that groth16.prove's buildABC1 forms
    and zkwg_h_evaluations_device the H evaluations of a real zkey's domain."""
    return list(constraints) + [({s: 1}, {}, {}) for s in range(n_public + 1)]


def email_verifier_r1cs(symbols, N, M, enable_header_masking=0, enable_body_masking=0, remove_soft_line_breaks_flag=0,
                        ignore_body_hash_check=0, public_rows=False):
    """bytes of the `.r1cs` file of EmailVerifier(N, M, 121, 17, 0, flags...) over the kept-v1 wires `symbols`
    ([(slot, name)], zkwg.Circuit.symbols()): public outputs (pubkeyHash, shaHi, shaLo, masked arrays), 17 public
    inputs (pubkey), the rest private."""
    cons = email_verifier_constraints(symbols, N, M, enable_header_masking, enable_body_masking, remove_soft_line_breaks_flag,
                                      ignore_body_hash_check)
    if ignore_body_hash_check:
        n_out = 3 + (N if enable_header_masking else 0)
        n_prv = N + 1 + 17 + (N if enable_header_masking else 0)
        if public_rows:
            cons = append_public_rows(cons, n_out + 17)
        return write_r1cs(len(symbols), cons, n_pub_out=n_out, n_pub_in=17, n_prv_in=n_prv)
    n_out = 3 + (N if enable_header_masking else 0) + (M if enable_body_masking else 0)
    if public_rows:      # the system a snarkjs zkey is built over (the prover attaches this one)
        cons = append_public_rows(cons, n_out + 17)
    n_prv = N + 1 + 17 + 1 + 32 + M + 1 + (N if enable_header_masking else 0) + (M if enable_body_masking else 0) + \
        (M if remove_soft_line_breaks_flag else 0)
    return write_r1cs(len(symbols), cons, n_pub_out=n_out, n_pub_in=17, n_prv_in=n_prv)


def _main():
    import argparse
    import zkwg
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--main", choices=["email-verifier", "sha256-bytes", "rsa-verifier"], default="email-verifier",
                    help="main component (tests/test-circuits/{email-verifier,sha,rsa}-test.circom)")
    ap.add_argument("--max-header", type=int, default=1024)
    ap.add_argument("--max-body", type=int, default=1536)
    ap.add_argument("--ignore-body-hash-check", type=int, default=0)
    ap.add_argument("--enable-header-masking", type=int, default=0)
    ap.add_argument("--enable-body-masking", type=int, default=0)
    ap.add_argument("--remove-soft-line-breaks", type=int, default=0)
    ap.add_argument("-o", "--output", required=True, help=".r1cs file to write")
    ap.add_argument("--sym", help="also write the layout's .sym file here")
    ap.add_argument("--public-rows", type=int, default=0,
                    help="1: append the nPublic + 1 rows snarkjs' zkey_new adds to A (the system a prover attaches; email-verifier only)")
    a = ap.parse_args()
    if a.main == "sha256-bytes":
        c = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=a.max_header, max_body=0, device=-1)
        sym = c.symbols()
        data = write_r1cs(len(sym), sha256_main_constraints(sym, a.max_header), n_pub_out=256, n_pub_in=a.max_header + 1)
    elif a.main == "rsa-verifier":
        c = zkwg.Circuit(zkwg.MAIN_RSA_VERIFIER, max_header=0, max_body=0, device=-1)
        sym = c.symbols()
        data = write_r1cs(len(sym), rsa_main_constraints(sym), n_pub_out=0, n_pub_in=17, n_prv_in=34)
    else:
        c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=a.max_header, max_body=a.max_body, device=-1,
                         ignore_body_hash_check=a.ignore_body_hash_check,
                         enable_header_masking=a.enable_header_masking, enable_body_masking=a.enable_body_masking,
                         remove_soft_line_breaks=a.remove_soft_line_breaks)
        sym = c.symbols()
        data = email_verifier_r1cs(sym, a.max_header, a.max_body, a.enable_header_masking, a.enable_body_masking,
                                   a.remove_soft_line_breaks, a.ignore_body_hash_check, public_rows=bool(a.public_rows))
    with open(a.output, "wb") as f:
        f.write(data)
    if a.sym:
        with open(a.sym, "w") as f:
            f.write("".join(f"{s},{s},0,{n}\n" for s, n in sym[1:]))
    print(f"{a.output}: {len(sym)} wires, {len(data)} bytes")


if __name__ == "__main__":
    _main()
