"""circomlib 2.0.5 templates restated (test infrastructure).

[EXT] circomlib is a yarn dependency of the reference (yarn.lock:3619-3624,
`circomlib@^2.0.5`) whose source is absent from /root/reference; the
templates below restate its published circuits (bitify, comparators, gates,
sha256/*).  Reference call sites: packages/circuits/lib/sha.circom:3-6,
lib/fp.circom:3-5, lib/bigint.circom:3-5, lib/base64.circom:3,
utils/array.circom:3-4, email-verifier.circom:3-4.

Every template function takes input *values* and returns the evaluated `Comp`.
Inputs are flagged 'L' (the parent wires them with a linear `<==`); callers
that wire a quadratic expression into a sub-component input re-flag it 'Q'.
"""
from .comp import Comp, P

# ---------------------------------------------------------------- bitify.circom


def Num2Bits(n, x):
    """bitify.circom Num2Bits(n): out[i] <-- (in >> i) & 1; out[i]*(out[i]-1)===0; sum === in."""
    c = Comp(f"Num2Bits({n})")
    out = c.out("out", n)
    i_ = c.inp("in")
    x %= P
    i_.set(x, "L")
    lc1 = 0
    for i in range(n):
        b = (x >> i) & 1
        out.v[i] = b
        out.k[i] = "H"
        lc1 += b << i
    c.eq(lc1, x, "lc1 === in")
    c.o = out.v
    return c


def Bits2Num(n, bits):
    """bitify.circom Bits2Num(n): out <== sum in[i]*2^i (linear)."""
    c = Comp(f"Bits2Num({n})")
    out = c.out("out")
    i_ = c.inp("in", n)
    i_.setall(bits, "L")
    lc1 = 0
    for i in range(n):
        lc1 += i_.v[i] << i
    out.set(lc1, "L")
    c.o = out.v[0]
    return c


# ----------------------------------------------------------- comparators.circom


def IsZero(x):
    """comparators.circom IsZero: inv <-- in!=0 ? 1/in : 0; out <== -in*inv+1; in*out === 0."""
    c = Comp("IsZero")
    out = c.out("out")
    i_ = c.inp("in")
    inv = c.mid("inv")
    x %= P
    i_.set(x, "L")
    iv = pow(x, P - 2, P) if x != 0 else 0
    inv.set(iv, "H")
    o = out.set(-x * iv + 1, "Q")
    c.eq(x * o, 0, "in*out === 0")
    c.o = o
    return c


def IsEqual(a, b):
    """comparators.circom IsEqual: isz.in <== in[1]-in[0]; out <== isz.out."""
    c = Comp("IsEqual")
    out = c.out("out")
    i_ = c.inp("in", 2)
    i_.setall([a, b], "L")
    isz = c.sub("isz", IsZero(i_.v[1] - i_.v[0]))
    c.o = out.set(isz.o, "L")
    return c


def LessThan(n, a, b):
    """comparators.circom LessThan(n): n2b=Num2Bits(n+1)(in[0]+(1<<n)-in[1]); out <== 1-n2b.out[n]."""
    assert n <= 252
    c = Comp(f"LessThan({n})")
    out = c.out("out")
    i_ = c.inp("in", 2)
    i_.setall([a, b], "L")
    n2b = c.sub("n2b", Num2Bits(n + 1, i_.v[0] + (1 << n) - i_.v[1]))
    c.o = out.set(1 - n2b.o[n], "L")
    return c


def LessEqThan(n, a, b):
    """comparators.circom LessEqThan(n): lt = LessThan(n)(in[0], in[1]+1)."""
    c = Comp(f"LessEqThan({n})")
    out = c.out("out")
    i_ = c.inp("in", 2)
    i_.setall([a, b], "L")
    lt = c.sub("lt", LessThan(n, i_.v[0], i_.v[1] + 1))
    c.o = out.set(lt.o, "L")
    return c


def GreaterThan(n, a, b):
    """comparators.circom GreaterThan(n): lt = LessThan(n)(in[1], in[0])."""
    c = Comp(f"GreaterThan({n})")
    out = c.out("out")
    i_ = c.inp("in", 2)
    i_.setall([a, b], "L")
    lt = c.sub("lt", LessThan(n, i_.v[1], i_.v[0]))
    c.o = out.set(lt.o, "L")
    return c


# ------------------------------------------------------------------ gates.circom


def AND(a, b):
    c = Comp("AND")
    out = c.out("out")
    c.inp("a").set(a, "L")
    c.inp("b").set(b, "L")
    c.o = out.set(a * b, "Q")
    return c


def OR(a, b):
    c = Comp("OR")
    out = c.out("out")
    c.inp("a").set(a, "L")
    c.inp("b").set(b, "L")
    c.o = out.set(a + b - a * b, "Q")
    return c


# ------------------------------------------------------------------ sha256/*

_H = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
_K = [
    0x428A2F98, 0x71374491, 0xB5C0FBCF, 0xE9B5DBA5, 0x3956C25B, 0x59F111F1, 0x923F82A4, 0xAB1C5ED5,
    0xD807AA98, 0x12835B01, 0x243185BE, 0x550C7DC3, 0x72BE5D74, 0x80DEB1FE, 0x9BDC06A7, 0xC19BF174,
    0xE49B69C1, 0xEFBE4786, 0x0FC19DC6, 0x240CA1CC, 0x2DE92C6F, 0x4A7484AA, 0x5CB0A9DC, 0x76F988DA,
    0x983E5152, 0xA831C66D, 0xB00327C8, 0xBF597FC7, 0xC6E00BF3, 0xD5A79147, 0x06CA6351, 0x14292967,
    0x27B70A85, 0x2E1B2138, 0x4D2C6DFC, 0x53380D13, 0x650A7354, 0x766A0ABB, 0x81C2C92E, 0x92722C85,
    0xA2BFE8A1, 0xA81A664B, 0xC24B8B70, 0xC76C51A3, 0xD192E819, 0xD6990624, 0xF40E3585, 0x106AA070,
    0x19A4C116, 0x1E376C08, 0x2748774C, 0x34B0BCB5, 0x391C0CB3, 0x4ED8AA4A, 0x5B9CCA4F, 0x682E6FF3,
    0x748F82EE, 0x78A5636F, 0x84C87814, 0x8CC70208, 0x90BEFFFA, 0xA4506CEB, 0xBEF9A3F7, 0xC67178F2,
]


def _const_bits(template, word):
    """sha256/constants.circom H(x)/K(x): out[i] <== (c >> i) & 1  (32 bits, LSB first)."""
    c = Comp(template)
    out = c.out("out", 32)
    out.setall([(word >> i) & 1 for i in range(32)], "L")
    c.o = out.v
    return c


def H(x):
    return _const_bits(f"H({x})", _H[x])


def K(x):
    return _const_bits(f"K({x})", _K[x])


def _alias_comp(template, in_bits, out_bits):
    c = Comp(template)
    o = c.out("out", len(out_bits))
    i_ = c.inp("in", len(in_bits))
    i_.setall(in_bits, "L")
    o.setall(out_bits, "L")
    c.o = o.v
    return c


def RotR(n, r, x):
    """sha256/rotate.circom RotR(n,r): out[i] <== in[(i+r)%n]."""
    return _alias_comp(f"RotR({n},{r})", x, [x[(i + r) % n] for i in range(n)])


def ShR(n, r, x):
    """sha256/shift.circom ShR(n,r): out[i] <== (i+r >= n) ? 0 : in[i+r]."""
    return _alias_comp(f"ShR({n},{r})", x, [0 if i + r >= n else x[i + r] for i in range(n)])


def Xor3(n, a, b, cc):
    """sha256/xor3.circom: mid[k] <== b[k]*c[k]; out[k] <== a[k]*(1-2b-2c+4mid)+b+c-2mid."""
    c = Comp(f"Xor3({n})")
    out = c.out("out", n)
    c.inp("a", n).setall(a, "L")
    c.inp("b", n).setall(b, "L")
    c.inp("c", n).setall(cc, "L")
    mid = c.mid("mid", n)
    for k in range(n):
        m = b[k] * cc[k]
        mid.v[k] = m
        mid.k[k] = "Q"
        out.v[k] = (a[k] * (1 - 2 * b[k] - 2 * cc[k] + 4 * m) + b[k] + cc[k] - 2 * m) % P
        out.k[k] = "Q"
    c.o = out.v
    return c


def Ch_t(n, a, b, cc):
    """sha256/ch.circom Ch_t: out[k] <== a[k]*(b[k]-c[k]) + c[k]."""
    c = Comp(f"Ch_t({n})")
    out = c.out("out", n)
    c.inp("a", n).setall(a, "L")
    c.inp("b", n).setall(b, "L")
    c.inp("c", n).setall(cc, "L")
    for k in range(n):
        out.v[k] = (a[k] * (b[k] - cc[k]) + cc[k]) % P
        out.k[k] = "Q"
    c.o = out.v
    return c


def Maj_t(n, a, b, cc):
    """sha256/maj.circom Maj_t: mid[k] <== b[k]*c[k]; out[k] <== a[k]*(b[k]+c[k]-2mid[k]) + mid[k]."""
    c = Comp(f"Maj_t({n})")
    out = c.out("out", n)
    c.inp("a", n).setall(a, "L")
    c.inp("b", n).setall(b, "L")
    c.inp("c", n).setall(cc, "L")
    mid = c.mid("mid", n)
    for k in range(n):
        m = b[k] * cc[k]
        mid.v[k] = m
        mid.k[k] = "Q"
        out.v[k] = (a[k] * (b[k] + cc[k] - 2 * m) + m) % P
        out.k[k] = "Q"
    c.o = out.v
    return c


def _nbits(a):
    n = 1
    r = 0
    while n - 1 < a:
        r += 1
        n *= 2
    return r


def BinSum(n, ops, ins):
    """binsum.circom BinSum(n,ops): out[k] <-- (lin >> k)&1, boolean, lin === lout."""
    nout = _nbits(((1 << n) - 1) * ops)
    c = Comp(f"BinSum({n},{ops})")
    out = c.out("out", nout)
    i_ = c.inp("in", ops * n)
    flat = []
    for j in range(ops):
        flat += list(ins[j])
    i_.setall(flat, "L")
    lin = 0
    for k in range(n):
        e2 = 1 << k
        for j in range(ops):
            lin += ins[j][k] * e2
    lout = 0
    for k in range(nout):
        b = (lin >> k) & 1
        out.v[k] = b
        out.k[k] = "H"
        lout += b << k
    c.eq(lin, lout, "lin === lout")
    c.o = out.v
    return c


def SmallSigma(ra, rb, rc, x):
    """sha256/sigma.circom SmallSigma: Xor3(RotR(ra), RotR(rb), ShR(rc))."""
    c = Comp(f"SmallSigma({ra},{rb},{rc})")
    out = c.out("out", 32)
    c.inp("in", 32).setall(x, "L")
    rota = c.sub("rota", RotR(32, ra, x))
    rotb = c.sub("rotb", RotR(32, rb, x))
    shrc = c.sub("shrc", ShR(32, rc, x))
    xor3 = c.sub("xor3", Xor3(32, rota.o, rotb.o, shrc.o))
    out.setall(xor3.o, "L")
    c.o = out.v
    return c


def BigSigma(ra, rb, rc, x):
    """sha256/sigma.circom BigSigma: Xor3 of three RotR."""
    c = Comp(f"BigSigma({ra},{rb},{rc})")
    out = c.out("out", 32)
    c.inp("in", 32).setall(x, "L")
    rota = c.sub("rota", RotR(32, ra, x))
    rotb = c.sub("rotb", RotR(32, rb, x))
    rotc = c.sub("rotc", RotR(32, rc, x))
    xor3 = c.sub("xor3", Xor3(32, rota.o, rotb.o, rotc.o))
    out.setall(xor3.o, "L")
    c.o = out.v
    return c


def SigmaPlus(in2, in7, in15, in16):
    """sha256/sigmaplus.circom: sigma1(in2) + in7 + sigma0(in15) + in16 via BinSum(32,4)."""
    c = Comp("SigmaPlus")
    out = c.out("out", 32)
    c.inp("in2", 32).setall(in2, "L")
    c.inp("in7", 32).setall(in7, "L")
    c.inp("in15", 32).setall(in15, "L")
    c.inp("in16", 32).setall(in16, "L")
    sigma1 = c.sub("sigma1", SmallSigma(17, 19, 10, in2))
    sigma0 = c.sub("sigma0", SmallSigma(7, 18, 3, in15))
    s = c.sub("sum", BinSum(32, 4, [sigma1.o, in7, sigma0.o, in16]))
    out.setall(s.o[:32], "L")
    c.o = out.v
    return c


def T1(h, e, f, g, k, w):
    """sha256/t1.circom: BinSum(32,5)(h, BigSigma(6,11,25)(e), Ch(e,f,g), k, w)."""
    c = Comp("T1")
    out = c.out("out", 32)
    for nm, v in (("h", h), ("e", e), ("f", f), ("g", g), ("k", k), ("w", w)):
        c.inp(nm, 32).setall(v, "L")
    ch = c.sub("ch", Ch_t(32, e, f, g))
    bigsigma1 = c.sub("bigsigma1", BigSigma(6, 11, 25, e))
    s = c.sub("sum", BinSum(32, 5, [h, bigsigma1.o, ch.o, k, w]))
    out.setall(s.o[:32], "L")
    c.o = out.v
    return c


def T2(a, b, cc):
    """sha256/t2.circom: BinSum(32,2)(BigSigma(2,13,22)(a), Maj(a,b,c))."""
    c = Comp("T2")
    out = c.out("out", 32)
    for nm, v in (("a", a), ("b", b), ("c", cc)):
        c.inp(nm, 32).setall(v, "L")
    bigsigma0 = c.sub("bigsigma0", BigSigma(2, 13, 22, a))
    maj = c.sub("maj", Maj_t(32, a, b, cc))
    s = c.sub("sum", BinSum(32, 2, [bigsigma0.o, maj.o]))
    out.setall(s.o[:32], "L")
    c.o = out.v
    return c


def Sha256compression(hin, inp):
    """sha256/sha256compression.circom.

    hin[256]: 8 words, each LSB-first; inp[512]: message block, big-endian bit
    order; out[256]: each word MSB-first (`out[32j+31-k] === fsum[j].out[k]`,
    `out <-- sha256compression(hin, inp)` hint => alias class, flagged 'L').
    Component creation order: sigmaPlus[48], ct_k[64], t1[64], t2[64],
    suma[64], sume[64], fsum[8].
    """
    c = Comp("Sha256compression")
    out = c.out("out", 256)
    c.inp("hin", 256).setall(hin, "L")
    c.inp("inp", 512).setall(inp, "L")
    regs = {nm: c.mid(nm, 65 * 32) for nm in "abcdefgh"}
    wsig = c.mid("w", 64 * 32)

    w = [None] * 64
    for t in range(16):
        w[t] = [inp[t * 32 + 31 - k] for k in range(32)]
    for t in range(16, 64):
        sp = c.sub(f"sigmaPlus[{t-16}]", SigmaPlus(w[t - 2], w[t - 7], w[t - 15], w[t - 16]))
        w[t] = sp.o
    ct_k = [c.sub(f"ct_k[{t}]", K(t)) for t in range(64)]

    st = {nm: [None] * 65 for nm in "abcdefgh"}
    for j, nm in enumerate("abcdefgh"):
        st[nm][0] = [hin[32 * j + k] for k in range(32)]

    t1s, t2s, sumas, sumes = [], [], [], []
    for t in range(64):
        t1 = T1(st["h"][t], st["e"][t], st["f"][t], st["g"][t], ct_k[t].o, w[t])
        t2 = T2(st["a"][t], st["b"][t], st["c"][t])
        sume = BinSum(32, 2, [st["d"][t], t1.o])
        suma = BinSum(32, 2, [t1.o, t2.o])
        t1s.append(t1); t2s.append(t2); sumas.append(suma); sumes.append(sume)
        st["h"][t + 1] = st["g"][t]
        st["g"][t + 1] = st["f"][t]
        st["f"][t + 1] = st["e"][t]
        st["e"][t + 1] = sume.o[:32]
        st["d"][t + 1] = st["c"][t]
        st["c"][t + 1] = st["b"][t]
        st["b"][t + 1] = st["a"][t]
        st["a"][t + 1] = suma.o[:32]
    for t in range(64):
        c.sub(f"t1[{t}]", t1s[t])
    for t in range(64):
        c.sub(f"t2[{t}]", t2s[t])
    for t in range(64):
        c.sub(f"suma[{t}]", sumas[t])
    for t in range(64):
        c.sub(f"sume[{t}]", sumes[t])

    outv = [0] * 256
    for j, nm in enumerate("abcdefgh"):
        fs = c.sub(f"fsum[{j}]", BinSum(32, 2, [[hin[32 * j + k] for k in range(32)], st[nm][64]]))
        for k in range(32):
            outv[32 * j + 31 - k] = fs.o[k]
    out.setall(outv, "L")

    for nm in "abcdefgh":
        flat = []
        for t in range(65):
            flat += st[nm][t]
        regs[nm].setall(flat, "L")
    flat = []
    for t in range(64):
        flat += w[t]
    wsig.setall(flat, "L")
    c.o = out.v
    return c
