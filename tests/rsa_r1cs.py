"""Test-only: the R1CS of `RSAVerifier65537(121, 17)` over the wires of the kept layout, derived from the
reference templates (lib/rsa.circom:13-181, lib/fp.circom:16-81, lib/bigint.circom:16-94 and the circomlib
comparators / bitify they instantiate): RSAPad, the signature range checks, BigLessThan, the 17 FpMul of
FpPow65537Mod with their polynomial identities and CheckCarryToZero, and the final equality.  Aliases and linear
signals are resolved symbolically into linear combinations of kept wires.  Used to run `checkConstraints` on
device witnesses, independently of the witness kernels and of the oracle's evaluator."""
from sha_r1cs import P, lc_add, const, wire

N_, K_ = 121, 17


def lc_scale(a, k):
    return {w: c * k % P for w, c in a.items() if c * k % P}


class RsaBuilder:
    def __init__(self, slot_of, prefix):
        self.slot = slot_of
        self.cons = []
        self.pre = prefix
        self._interp = None

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
        k = K_
        lt = [self.less_than(f"{pre}.lt[{i}]", N_, a[i], b[i]) for i in range(k)]
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
        # t = poly_interp(v) is linear in v (lib/bigint-func.circom:65-103): its matrix from unit vectors
        if self._interp is None:
            from oracle.pyref import bigint_func as bf
            cols = [bf.poly_interp(n, [1 if y == x else 0 for y in range(n)]) for x in range(n)]
            self._interp = [[cols[x][i] % P for x in range(n)] for i in range(n)]
        return self._interp

    def fp_mul(self, pre, a, b, p):
        k, m = K_, 2 * K_ - 1
        v_ab = self.arr(pre + ".v_ab", m)
        q = [wire(s) for s in self.arr(pre + ".q", k)]
        r = [wire(s) for s in self.arr(pre + ".r", k)]
        v_pq_r = self.arr(pre + ".v_pq_r", m)
        for x in range(m):
            self.cons.append((self.poly(a, x), self.poly(b, x), wire(v_ab[x])))
        for i in range(k):
            self.num2bits(f"{pre}.q_range_check[{i}]", q[i], N_)
        for i in range(k):
            self.num2bits(f"{pre}.r_range_check[{i}]", r[i], N_)
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
            self.lin(lc_add(lhs, wire(carry[i]), -(1 << N_)))
            self.num2bits(f"{pre}.tCheck.carryRangeChecks[{i}]", lc_add(wire(carry[i]), const(1 << 130)), 131)
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
