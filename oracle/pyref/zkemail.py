"""Literal restatement of the reference's own circom templates (test infrastructure).

Follows packages/circuits/{email-verifier.circom, lib/*.circom, utils/*.circom}
statement by statement; each function cites the reference file:line.
"""
from .comp import Comp, P, AssertFailed
from . import circomlib as cl
from . import bigint_func as bf


def log2Ceil(a):  # utils/functions.circom:7-17
    n = a - 1
    r = 0
    while n > 0:
        r += 1
        n //= 2
    return r


# ------------------------------------------------------------ utils/array.circom


def CalculateTotal(n, nums, how=None):
    """utils/array.circom:51-64. sums[] and sum are linear."""
    c = Comp(f"CalculateTotal({n})")
    s_out = c.out("sum")
    i_ = c.inp("nums", n)
    i_.setall(nums, "L")
    if how is not None:
        i_.k = list(how)
    sums = c.mid("sums", n)
    acc = i_.v[0]
    sums.set(acc, "L", 0)
    for i in range(1, n):
        acc = (acc + i_.v[i]) % P
        sums.set(acc, "L", i)
    c.o = s_out.set(acc, "L")
    return c


def ItemAtIndex(n, in_, index):
    """utils/array.circom:16-43."""
    c = Comp(f"ItemAtIndex({n})")
    out = c.out("out")
    c.inp("in", n).setall(in_, "L")
    c.inp("index").set(index, "L")
    eqs = [cl.IsEqual(i, index) for i in range(n)]
    calcTotalValue = c.sub("calcTotalValue", CalculateTotal(n, [eqs[i].o * in_[i] for i in range(n)], ["Q"] * n))
    calcTotalIndex = c.sub("calcTotalIndex", CalculateTotal(n, [eqs[i].o for i in range(n)]))
    for i in range(n):
        c.sub(f"eqs[{i}]", eqs[i])
    c.eq(calcTotalIndex.o, 1, "calcTotalIndex.sum === 1 (array.circom:40)")
    c.o = out.set(calcTotalValue.o, "L")
    return c


def VarShiftLeft(maxArrayLen, maxOutArrayLen, in_, shift):
    """utils/array.circom:111-141."""
    assert maxOutArrayLen <= maxArrayLen
    bitLength = log2Ceil(maxArrayLen)
    c = Comp(f"VarShiftLeft({maxArrayLen},{maxOutArrayLen})")
    out = c.out("out", maxOutArrayLen)
    c.inp("in", maxArrayLen).setall(in_, "L")
    c.inp("shift").set(shift, "L")
    tmp = c.mid("tmp", bitLength * maxArrayLen)
    n2b = c.sub("n2b", cl.Num2Bits(bitLength, shift))
    prev = [x % P for x in in_]
    for j in range(bitLength):
        cur = [0] * maxArrayLen
        bit = n2b.o[j]
        for i in range(maxArrayLen):
            offset = (i + (1 << j)) % maxArrayLen
            cur[i] = (bit * (prev[offset] - prev[i]) + prev[i]) % P
        tmp.v[j * maxArrayLen:(j + 1) * maxArrayLen] = cur
        tmp.k[j * maxArrayLen:(j + 1) * maxArrayLen] = ["Q"] * maxArrayLen
        prev = cur
    out.setall(prev[:maxOutArrayLen], "L")
    c.o = out.v
    return c


def AssertZeroPadding(maxArrayLen, in_, startIndex):
    """utils/array.circom:149-164."""
    bitLength = log2Ceil(maxArrayLen)
    c = Comp(f"AssertZeroPadding({maxArrayLen})")
    c.inp("in", maxArrayLen).setall(in_, "L")
    c.inp("startIndex").set(startIndex, "L")
    for i in range(maxArrayLen):
        lt = c.sub(f"lessThans[{i}]", cl.LessThan(bitLength, startIndex - 1, i))
        c.eq(lt.o * in_[i], 0, f"lessThans[{i}].out * in[{i}] === 0 (array.circom:162)")
    return c


# ------------------------------------------------------------ utils/bytes.circom


def PackBits(numBits, bitsPerElement, in_):
    """utils/bytes.circom:194-210 (big-endian packing; linear)."""
    numElements = (numBits + bitsPerElement - 1) // bitsPerElement
    c = Comp(f"PackBits({numBits},{bitsPerElement})")
    out = c.out("out", numElements)
    c.inp("in", numBits).setall(in_, "L")
    for i in range(numElements):
        s = 0
        for j in range(bitsPerElement):
            idx = i * bitsPerElement + j
            if idx < numBits:
                s += in_[idx] * (1 << (bitsPerElement - 1 - j))
        out.set(s, "L", i)
    c.o = out.v
    return c


def ByteMask(maxLength, in_, mask):
    """utils/bytes.circom:173-185 (+AssertBit :155-158)."""
    c = Comp(f"ByteMask({maxLength})")
    out = c.out("out", maxLength)
    c.inp("in", maxLength).setall(in_, "L")
    c.inp("mask", maxLength).setall(mask, "L")
    for i in range(maxLength):
        ab = Comp("AssertBit")
        ab.inp("in").set(mask[i], "L")
        ab.eq(mask[i] * (mask[i] - 1), 0, "in*(in-1) === 0 (bytes.circom:157)")
        c.sub(f"bit_check[{i}]", ab)
        out.set(in_[i] * mask[i], "Q", i)
    c.o = out.v
    return c


# ------------------------------------------------------------ utils/regex.circom


def SelectRegexReveal(maxArrayLen, maxRevealLen, in_, startIndex):
    """utils/regex.circom:17-52."""
    c = Comp(f"SelectRegexReveal({maxArrayLen},{maxRevealLen})")
    out = c.out("out", maxRevealLen)
    c.inp("in", maxArrayLen).setall(in_, "L")
    c.inp("startIndex").set(startIndex, "L")
    bitLength = log2Ceil(maxArrayLen + maxRevealLen - 1)
    isStartIndex = c.mid("isStartIndex", maxArrayLen)
    isZero = c.mid("isZero", maxArrayLen)
    isPreviousZero = c.mid("isPreviousZero", maxArrayLen)
    isAbove = c.mid("isAboveMaxRevealLen", maxArrayLen)
    isPreviousZero.set(1, "L", 0)
    for i in range(maxArrayLen):
        a = c.sub(f"anon_IsEqual[{i}]", cl.IsEqual(i, startIndex))
        isStartIndex.set(a.o, "L", i)
        z = c.sub(f"anon_IsZero[{i}]", cl.IsZero(in_[i]))
        isZero.set(z.o, "L", i)
        if i > 0:
            pz = c.sub(f"anon_IsPrevZero[{i}]", cl.IsZero(in_[i - 1]))
            isPreviousZero.set(pz.o, "L", i)
        g = c.sub(f"anon_GreaterThan[{i}]", cl.GreaterThan(bitLength, i, startIndex + maxRevealLen - 1))
        isAbove.set(g.o, "L", i)
        c.eq(isStartIndex.v[i] * isZero.v[i], 0, "regex.circom:39")
        c.eq(isStartIndex.v[i] * (1 - isPreviousZero.v[i]), 0, "regex.circom:44")
        c.eq(isAbove.v[i] * (1 - isZero.v[i]), 0, "regex.circom:47")
    vs = c.sub("anon_VarShiftLeft", VarShiftLeft(maxArrayLen, maxRevealLen, in_, startIndex))
    out.setall(vs.o, "L")
    c.o = out.v
    return c


# ------------------------------------------------------------ lib/base64.circom


def Base64Lookup(x):
    """lib/base64.circom:71-128."""
    c = Comp("Base64Lookup")
    out = c.out("out")
    c.inp("in").set(x, "L")
    x %= P
    le_Z = c.sub("le_Z", cl.LessThan(8, x, 90 + 1))
    ge_A = c.sub("ge_A", cl.GreaterThan(8, x, 65 - 1))
    range_AZ = c.mid("range_AZ").set(ge_A.o * le_Z.o, "Q")
    sum_AZ = c.mid("sum_AZ").set(range_AZ * (x - 65), "Q")
    le_z = c.sub("le_z", cl.LessThan(8, x, 122 + 1))
    ge_a = c.sub("ge_a", cl.GreaterThan(8, x, 97 - 1))
    range_az = c.mid("range_az").set(ge_a.o * le_z.o, "Q")
    sum_az = c.mid("sum_az").set(sum_AZ + range_az * (x - 71), "Q")
    le_9 = c.sub("le_9", cl.LessThan(8, x, 57 + 1))
    ge_0 = c.sub("ge_0", cl.GreaterThan(8, x, 48 - 1))
    range_09 = c.mid("range_09").set(ge_0.o * le_9.o, "Q")
    sum_09 = c.mid("sum_09").set(sum_az + range_09 * (x + 4), "Q")
    equal_plus = c.sub("equal_plus", cl.IsZero(x - 43))
    sum_plus = c.mid("sum_plus").set(sum_09 + equal_plus.o * (x + 19), "Q")
    equal_slash = c.sub("equal_slash", cl.IsZero(x - 47))
    sum_slash = c.mid("sum_slash").set(sum_plus + equal_slash.o * (x + 16), "Q")
    c.o = out.set(sum_slash, "L")
    equal_eqsign = c.sub("equal_eqsign", cl.IsZero(x - 61))
    c.eq(1, range_AZ + range_az + range_09 + equal_plus.o + equal_slash.o + equal_eqsign.o,
         "base64.circom:127")
    return c


def Base64Decode(byteLength, in_):
    """lib/base64.circom:14-64."""
    charLength = 4 * ((byteLength + 2) // 3)
    c = Comp(f"Base64Decode({byteLength})")
    out = c.out("out", byteLength)
    c.inp("in", charLength).setall(in_, "L")
    idx = 0
    groups = charLength // 4
    bitsIn = [[None] * 4 for _ in range(groups)]
    bitsOut = [[None] * 3 for _ in range(groups)]
    translate = [[None] * 4 for _ in range(groups)]
    for i in range(0, charLength, 4):
        g = i // 4
        for j in range(4):
            translate[g][j] = Base64Lookup(in_[i + j])
            bitsIn[g][j] = cl.Num2Bits(6, translate[g][j].o)
        bo0 = [0] * 8
        for j in range(6):
            bo0[j + 2] = bitsIn[g][0].o[j]
        bo0[0] = bitsIn[g][1].o[4]
        bo0[1] = bitsIn[g][1].o[5]
        bo1 = [0] * 8
        for j in range(4):
            bo1[j + 4] = bitsIn[g][1].o[j]
        for j in range(4):
            bo1[j] = bitsIn[g][2].o[j + 2]
        bo2 = [0] * 8
        bo2[6] = bitsIn[g][2].o[0]
        bo2[7] = bitsIn[g][2].o[1]
        for j in range(6):
            bo2[j] = bitsIn[g][3].o[j]
        bitsOut[g][0] = cl.Bits2Num(8, bo0)
        bitsOut[g][1] = cl.Bits2Num(8, bo1)
        bitsOut[g][2] = cl.Bits2Num(8, bo2)
        for j in range(3):
            if idx + j < byteLength:
                out.set(bitsOut[g][j].o, "L", idx + j)
        idx += 3
    # component arrays in declaration order: bitsIn, bitsOut, translate (base64.circom:20-22)
    for g in range(groups):
        for j in range(4):
            c.sub(f"bitsIn[{g}][{j}]", bitsIn[g][j])
    for g in range(groups):
        for j in range(3):
            c.sub(f"bitsOut[{g}][{j}]", bitsOut[g][j])
    for g in range(groups):
        for j in range(4):
            c.sub(f"translate[{g}][{j}]", translate[g][j])
    c.o = out.v
    return c


# ------------------------------------------------------------ lib/bigint.circom


def BigLessThan(n, k, a, b):
    """lib/bigint.circom:16-60."""
    c = Comp(f"BigLessThan({n},{k})")
    out = c.out("out")
    c.inp("a", k).setall(a, "L")
    c.inp("b", k).setall(b, "L")
    lt = [cl.LessThan(n, a[i], b[i]) for i in range(k)]
    eq = [cl.IsEqual(a[i], b[i]) for i in range(k)]
    ors = [None] * (k - 1)
    ands = [None] * (k - 1)
    eq_ands = [None] * (k - 1)
    for i in range(k - 2, -1, -1):
        if i == k - 2:
            ands[i] = cl.AND(eq[k - 1].o, lt[k - 2].o)
            eq_ands[i] = cl.AND(eq[k - 1].o, eq[k - 2].o)
            ors[i] = cl.OR(lt[k - 1].o, ands[i].o)
        else:
            ands[i] = cl.AND(eq_ands[i + 1].o, lt[i].o)
            eq_ands[i] = cl.AND(eq_ands[i + 1].o, eq[i].o)
            ors[i] = cl.OR(ors[i + 1].o, ands[i].o)
    for i in range(k):
        c.sub(f"lt[{i}]", lt[i])
    for i in range(k):
        c.sub(f"eq[{i}]", eq[i])
    for i in range(k - 1):
        c.sub(f"ors[{i}]", ors[i])
    for i in range(k - 1):
        c.sub(f"ands[{i}]", ands[i])
    for i in range(k - 1):
        c.sub(f"eq_ands[{i}]", eq_ands[i])
    c.o = out.set(ors[0].o, "L")
    return c


def CheckCarryToZero(n, m, k, in_):
    """lib/bigint.circom:69-94.  carry[k-1] is declared but never assigned (reads 0)."""
    assert k >= 2
    EPSILON = 3
    assert m + EPSILON <= 253
    c = Comp(f"CheckCarryToZero({n},{m},{k})")
    c.inp("in", k).setall(in_, "L")
    carry = c.mid("carry", k)
    for i in range(k - 1):
        if i == 0:
            cv = bf.fdiv(in_[i], bf.shl(1, n))
            carry.set(cv, "H", i)
            c.eq(in_[i], carry.v[i] * (1 << n), "bigint.circom:84")
        else:
            cv = bf.fdiv(in_[i] + carry.v[i - 1], bf.shl(1, n))
            carry.set(cv, "H", i)
            c.eq(in_[i] + carry.v[i - 1], carry.v[i] * (1 << n), "bigint.circom:88")
        c.sub(f"carryRangeChecks[{i}]",
              cl.Num2Bits(m + EPSILON - n, carry.v[i] + (1 << (m + EPSILON - n - 1))))
    c.eq(in_[k - 1] + carry.v[k - 2], 0, "bigint.circom:93")
    return c


# ------------------------------------------------------------ lib/fp.circom


def FpMul(n, k, a, b, p, qr_override=None):
    """lib/fp.circom:16-81.  `qr_override=(q, r)` reproduces the test-only template
    tests/test-circuits/fp-mul-test-range-check.circom that takes q, r as inputs."""
    assert n + n + bf.log_ceil(k) + 2 <= 252
    c = Comp(f"FpMul({n},{k})")
    out = c.out("out", k)
    c.inp("a", k).setall(a, "L")
    c.inp("b", k).setall(b, "L")
    c.inp("p", k).setall(p, "L")
    a = [x % P for x in a]
    b = [x % P for x in b]
    p = [x % P for x in p]
    v_ab = c.mid("v_ab", 2 * k - 1)
    for x in range(2 * k - 1):
        v_a = bf.poly_eval(k, a, x)
        v_b = bf.poly_eval(k, b, x)
        v_ab.set(v_a * v_b, "Q", x)
    ab = bf.poly_interp(2 * k - 1, v_ab.v)
    ab_proper = bf.getProperRepresentation(n + n + bf.log_ceil(k), n, 2 * k - 1, ab)
    long_div_out = bf.long_div(n, k, k, ab_proper, p)

    q = c.mid("q", k)
    r = c.mid("r", k)
    for i in range(k):
        q.set(long_div_out[0][i] if qr_override is None else qr_override[0][i], "H", i)
        r.set(long_div_out[1][i] if qr_override is None else qr_override[1][i], "H", i)
    q_rc = [cl.Num2Bits(n, q.v[i]) for i in range(k)]
    r_rc = [cl.Num2Bits(n, r.v[i]) for i in range(k)]
    for i in range(k):
        c.sub(f"q_range_check[{i}]", q_rc[i])
    for i in range(k):
        c.sub(f"r_range_check[{i}]", r_rc[i])
    r_p_lt = c.sub("r_p_lt_check", BigLessThan(n, k, r.v, p))
    c.eq(r_p_lt.o, 1, "r_p_lt_check.out === 1 (fp.circom:57)")

    v_pq_r = c.mid("v_pq_r", 2 * k - 1)
    for x in range(2 * k - 1):
        v_p = bf.poly_eval(k, p, x)
        v_q = bf.poly_eval(k, q.v, x)
        v_r = bf.poly_eval(k, r.v, x)
        v_pq_r.set(v_p * v_q + v_r, "Q", x)
    v_t = c.mid("v_t", 2 * k - 1)
    for x in range(2 * k - 1):
        v_t.set(v_ab.v[x] - v_pq_r.v[x], "L", x)
    t = bf.poly_interp(2 * k - 1, v_t.v)
    c.sub("tCheck", CheckCarryToZero(n, n + n + bf.log_ceil(k) + 2, 2 * k - 1, t[:2 * k - 1]))
    out.setall(r.v, "L")
    c.o = out.v
    return c


# ------------------------------------------------------------ lib/rsa.circom


def RSAPad(n, k, modulus, message):
    """lib/rsa.circom:101-181."""
    c = Comp(f"RSAPad({n},{k})")
    out = c.out("out", k)
    c.inp("modulus", k).setall(modulus, "L")
    c.inp("message", k).setall(message, "L")
    baseLen = 408
    msgLen = 256
    paddedMessageBits = c.mid("paddedMessageBits", n * k)
    modulusBits = c.mid("modulusBits", n * k)
    messageBits = c.mid("messageBits", n * k)
    modulusN2B = [None] * k
    messageN2B = [None] * k
    for i in range(k):
        messageN2B[i] = cl.Num2Bits(n, message[i])
        for j in range(n):
            messageBits.set(messageN2B[i].o[j], "L", i * n + j)
        modulusN2B[i] = cl.Num2Bits(n, modulus[i])
        for j in range(n):
            modulusBits.set(modulusN2B[i].o[j], "L", i * n + j)
    for i in range(k):
        c.sub(f"modulusN2B[{i}]", modulusN2B[i])
    for i in range(k):
        c.sub(f"messageN2B[{i}]", messageN2B[i])
    for i in range(msgLen, n * k):
        c.eq(messageBits.v[i], 0, "messageBits[i] === 0 (rsa.circom:128)")
    for i in range(msgLen):
        paddedMessageBits.set(messageBits.v[i], "L", i)
    for i in range(baseLen, baseLen + 8):
        paddedMessageBits.set(0, "L", i)
    for i in range(msgLen, baseLen):
        paddedMessageBits.set((0x3031300D060960864801650304020105000420 >> (i - msgLen)) & 1, "L", i)
    nz = (n * k + 7 - (baseLen + 8)) // 8
    modulusZero = [None] * nz
    modulusPrefix = 0
    for i in range(n * k - 1, baseLen + 8 - 1, -1):
        if i + 8 < n * k:
            modulusPrefix += modulusBits.v[i + 8]
            if i % 8 == 0:
                idx = (i - (baseLen + 8)) // 8
                modulusZero[idx] = cl.IsZero(modulusPrefix)
                paddedMessageBits.set(1 - modulusZero[idx].o, "L", i)
            else:
                paddedMessageBits.set(paddedMessageBits.v[i + 1], "L", i)
        else:
            paddedMessageBits.set(0, "L", i)
    for idx in range(nz):
        if modulusZero[idx] is not None:
            c.sub(f"modulusZero[{idx}]", modulusZero[idx])
    assert baseLen + 8 + 65 <= n * k
    for i in range(baseLen + 8, baseLen + 8 + 65):
        c.eq(paddedMessageBits.v[i], 1, "paddedMessageBits[i] === 1 (rsa.circom:170)")
    for i in range(k):
        b2n = c.sub(f"passedMessageB2N[{i}]", cl.Bits2Num(n, paddedMessageBits.v[i * n:(i + 1) * n]))
        out.set(b2n.o, "L", i)
    c.o = out.v
    return c


def FpPow65537Mod(n, k, base, modulus):
    """lib/rsa.circom:57-92: 16 squarings + 1 multiply."""
    c = Comp(f"FpPow65537Mod({n},{k})")
    out = c.out("out", k)
    c.inp("base", k).setall(base, "L")
    c.inp("modulus", k).setall(modulus, "L")
    doublers = [None] * 16
    cur = list(base)
    for i in range(16):
        doublers[i] = FpMul(n, k, cur, cur, modulus)
        cur = doublers[i].o
    for i in range(16):
        c.sub(f"doublers[{i}]", doublers[i])
    adder = c.sub("adder", FpMul(n, k, base, doublers[15].o, modulus))
    out.setall(adder.o, "L")
    c.o = out.v
    return c


def RSAVerifier65537(n, k, message, signature, modulus, is_main=False):
    """lib/rsa.circom:13-46."""
    c = Comp(f"RSAVerifier65537({n},{k})", is_main=is_main)
    c.inp("message", k).setall(message, "L")
    c.inp("signature", k).setall(signature, "L")
    c.inp("modulus", k).setall(modulus, "L")
    if is_main:
        c.public = {"modulus"}  # tests/test-circuits/rsa-test.circom:5
    padder = c.sub("padder", RSAPad(n, k, modulus, message))
    src = [cl.Num2Bits(n, signature[i]) for i in range(k)]
    for i in range(k):
        c.sub(f"signatureRangeCheck[{i}]", src[i])
    blt = c.sub("bigLessThan", BigLessThan(n, k, signature, modulus))
    c.eq(blt.o, 1, "bigLessThan.out === 1 (rsa.circom:33)")
    bigPow = c.sub("bigPow", FpPow65537Mod(n, k, signature, modulus))
    for i in range(k):
        c.eq(bigPow.o[i], padder.o[i], "bigPow.out[i] === padder.out[i] (rsa.circom:44)")
    return c


# ------------------------------------------------------------ lib/sha.circom


def _sha_core(c, maxBitLength, paddedIn, paddedInLength, first_hin):
    """Shared body of Sha256General (lib/sha.circom:89-203) and Sha256Partial (:212-292)."""
    assert maxBitLength % 512 == 0
    maxBitsPaddedBits = log2Ceil(maxBitLength)
    maxBlocks = maxBitLength // 512
    inBlockIndex = c.mid("inBlockIndex")
    ibi = inBlockIndex.set(bf.shr(paddedInLength, 9), "H")
    c.eq(paddedInLength, ibi * 512, "paddedInLength === inBlockIndex * 512 (sha.circom:112)")
    blv = c.sub("bitLengthVerifier", cl.LessEqThan(maxBitsPaddedBits, paddedInLength, maxBitLength))
    c.eq(blv.o, 1, "bitLengthVerifier.out === 1 (sha.circom:129)")
    hin = first_hin(c)
    comps = []
    for i in range(maxBlocks):
        sc = c.sub(f"sha256compression[{i}]", cl.Sha256compression(hin, paddedIn[i * 512:(i + 1) * 512]))
        comps.append(sc)
        hin = [0] * 256
        for j in range(8):
            for k in range(32):
                hin[32 * j + k] = sc.o[32 * j + 31 - k]
    outv = [0] * 256
    for k in range(256):
        sel = c.sub(f"arraySelectors[{k}]",
                    ItemAtIndex(maxBlocks, [comps[j].o[k] for j in range(maxBlocks)], ibi - 1))
        outv[k] = sel.o
    return outv


def Sha256General(maxBitLength, paddedIn, paddedInLength):
    """lib/sha.circom:89-203."""
    c = Comp(f"Sha256General({maxBitLength})")
    out = c.out("out", 256)
    c.inp("paddedIn", maxBitLength).setall(paddedIn, "L")
    c.inp("paddedInLength").set(paddedInLength, "L")

    def first_hin(cc):
        hs = [cc.sub(f"h{'abcdefgh'[j]}0", cl.H(j)) for j in range(8)]
        hin = []
        for j in range(8):
            hin += hs[j].o
        return hin

    out.setall(_sha_core(c, maxBitLength, paddedIn, paddedInLength % P, first_hin), "L")
    c.o = out.v
    return c


def Sha256Partial(maxBitLength, paddedIn, paddedInLength, preHash):
    """lib/sha.circom:212-292."""
    c = Comp(f"Sha256Partial({maxBitLength})")
    out = c.out("out", 256)
    c.inp("paddedIn", maxBitLength).setall(paddedIn, "L")
    c.inp("paddedInLength").set(paddedInLength, "L")
    c.inp("preHash", 256).setall(preHash, "L")

    def first_hin(cc):
        hin = [0] * 256
        for j in range(8):
            for k in range(32):
                hin[32 * j + k] = preHash[32 * j + 31 - k]
        return hin

    out.setall(_sha_core(c, maxBitLength, paddedIn, paddedInLength % P, first_hin), "L")
    c.o = out.v
    return c


def Sha256Bytes(maxByteLength, paddedIn, paddedInLength, is_main=False):
    """lib/sha.circom:17-38."""
    c = Comp(f"Sha256Bytes({maxByteLength})", is_main=is_main)
    out = c.out("out", 256)
    c.inp("paddedIn", maxByteLength).setall(paddedIn, "L")
    c.inp("paddedInLength").set(paddedInLength, "L")
    if is_main:
        c.public = {"paddedIn", "paddedInLength"}  # tests/test-circuits/sha-test.circom:5
    bytes_ = [cl.Num2Bits(8, paddedIn[i]) for i in range(maxByteLength)]
    bits = []
    for i in range(maxByteLength):
        bits += [bytes_[i].o[7 - j] for j in range(8)]
    sha = c.sub("sha", Sha256General(maxByteLength * 8, bits, paddedInLength * 8))
    for i in range(maxByteLength):
        c.sub(f"bytes[{i}]", bytes_[i])
    out.setall(sha.o, "L")
    c.o = out.v
    return c


def Sha256BytesPartial(maxByteLength, paddedIn, paddedInLength, preHash):
    """lib/sha.circom:47-80."""
    assert maxByteLength % 32 == 0
    c = Comp(f"Sha256BytesPartial({maxByteLength})")
    out = c.out("out", 256)
    c.inp("paddedIn", maxByteLength).setall(paddedIn, "L")
    c.inp("paddedInLength").set(paddedInLength, "L")
    c.inp("preHash", 32).setall(preHash, "L")
    bytes_ = [cl.Num2Bits(8, paddedIn[i]) for i in range(maxByteLength)]
    bits = []
    for i in range(maxByteLength):
        bits += [bytes_[i].o[7 - j] for j in range(8)]
    states = [cl.Num2Bits(8, preHash[i]) for i in range(32)]
    pre = []
    for i in range(32):
        pre += [states[i].o[7 - j] for j in range(8)]
    sha = c.sub("sha", Sha256Partial(maxByteLength * 8, bits, paddedInLength * 8, pre))
    for i in range(maxByteLength):
        c.sub(f"bytes[{i}]", bytes_[i])
    for i in range(32):
        c.sub(f"states[{i}]", states[i])
    out.setall(sha.o, "L")
    c.o = out.v
    return c


# ------------------------------------------------------------ utils/hash.circom


def PoseidonLarge(bitsPerChunk, chunkSize, in_):
    """utils/hash.circom:15-39."""
    from . import poseidon as pos
    assert chunkSize > 16 and chunkSize <= 32 and bitsPerChunk * 2 < 251
    halfChunkSize = chunkSize >> 1
    if chunkSize % 2 == 1:
        halfChunkSize += 1
    c = Comp(f"PoseidonLarge({bitsPerChunk},{chunkSize})")
    out = c.out("out")
    c.inp("in", chunkSize).setall(in_, "L")
    poseidonInput = c.mid("poseidonInput", halfChunkSize)
    for i in range(halfChunkSize):
        if i == halfChunkSize - 1 and chunkSize % 2 == 1:
            poseidonInput.set(in_[2 * i], "L", i)
        else:
            poseidonInput.set(in_[2 * i] + (1 << bitsPerChunk) * in_[2 * i + 1], "L", i)
    h = c.sub("anon_Poseidon", pos.Poseidon(halfChunkSize, poseidonInput.v))
    c.o = out.set(h.o, "L")
    return c


def Slice(n, start, end, in_):
    """utils/array.circom:175-186: out[i - start] <== in[i]."""
    assert n >= end and start >= 0 and end >= start
    c = Comp(f"Slice({n},{start},{end})")
    out = c.out("out", end - start)
    c.inp("in", n).setall(in_, "L")
    out.setall(in_[start:end], "L")
    c.o = out.v
    return c


def PoseidonModular(numElements, in_):
    """utils/hash.circom:49-82: Poseidon(16) per chunk, chained through Poseidon(2)."""
    from . import poseidon as pos
    c = Comp(f"PoseidonModular({numElements})")
    out = c.out("out")
    c.inp("in", numElements).setall(in_, "L")
    chunks = numElements // 16
    last_chunk_size = numElements % 16
    if last_chunk_size != 0:
        chunks += 1
    _out = 0
    for i in range(chunks):
        start = i * 16
        end = start + 16
        if end > numElements:
            end = numElements
            sl = c.sub(f"anon_Slice[{i}]", Slice(numElements, start, end, in_))
            ch = c.sub(f"anon_Poseidon_chunk[{i}]", pos.Poseidon(last_chunk_size, sl.o))
        else:
            sl = c.sub(f"anon_Slice[{i}]", Slice(numElements, start, end, in_))
            ch = c.sub(f"anon_Poseidon_chunk[{i}]", pos.Poseidon(16, sl.o))
        if i == 0:
            _out = ch.o
        else:
            _out = c.sub(f"anon_Poseidon_merge[{i}]", pos.Poseidon(2, [_out, ch.o])).o
    c.o = out.set(_out, "L")
    return c


def Mux1(c0, c1, s, how0="L"):
    """circomlib mux1.circom [EXT]: Mux1 wraps MultiMux1(1); out <== (c[1] - c[0])*s + c[0].
    `how0`: how the parent assigned c[0] (RemoveSoftLineBreaks feeds a product into it)."""
    c = Comp("Mux1")
    out = c.out("out")
    ci = c.inp("c", 2)
    ci.set(c0, how0, 0)
    ci.set(c1, "L", 1)
    c.inp("s").set(s, "L")
    m = Comp("MultiMux1(1)")
    mout = m.out("out", 1)
    m.inp("c", 2).setall([c0, c1], "L")      # c[0][0], c[0][1]
    m.inp("s").set(s, "L")
    mout.set((c1 - c0) * s + c0, "Q", 0)
    c.sub("mux", m)
    c.o = out.set(mout.v[0], "L")
    return c


def RemoveSoftLineBreaks(maxLength, encoded, decoded, is_main=False):
    """helpers/remove-soft-line-breaks.circom:14-126."""
    c = Comp(f"RemoveSoftLineBreaks({maxLength})", is_main=is_main)
    M = maxLength
    isValid = c.out("isValid")
    c.inp("encoded", M).setall(encoded, "L")
    c.inp("decoded", M).setall(decoded, "L")
    encoded = [x % P for x in encoded]
    decoded = [x % P for x in decoded]
    r_s = c.mid("r")
    processed = c.mid("processed", M)
    isEquals = c.mid("isEquals", M)
    isCr = c.mid("isCr", M)
    isLf = c.mid("isLf", M)
    tempSoftBreak = c.mid("tempSoftBreak", M - 2)
    isSoftBreak = c.mid("isSoftBreak", M)
    shouldZero = c.mid("shouldZero", M)
    rEnc = c.mid("rEnc", M)
    sumEnc = c.mid("sumEnc", M)
    rDec = c.mid("rDec", M)
    sumDec = c.mid("sumDec", M)
    # `component muxEnc[maxLength]` is declared before rHasher (:33-36)
    mux_slots = [None] * M
    for i in range(M):
        c.subs.append([f"muxEnc[{i}]", None])
        mux_slots[i] = c.subs[-1]
    rHasher = c.sub("rHasher", PoseidonModular(2 * M, encoded + decoded))
    r = r_s.set(rHasher.o, "L")
    for i in range(M):
        isEquals.set(c.sub(f"anon_IsEqual_eq[{i}]", cl.IsEqual(encoded[i], 61)).o, "L", i)
    for i in range(M - 1):
        isCr.set(c.sub(f"anon_IsEqual_cr[{i}]", cl.IsEqual(encoded[i + 1], 13)).o, "L", i)
    isCr.set(0, "L", M - 1)
    for i in range(M - 2):
        isLf.set(c.sub(f"anon_IsEqual_lf[{i}]", cl.IsEqual(encoded[i + 2], 10)).o, "L", i)
    isLf.set(0, "L", M - 2)
    isLf.set(0, "L", M - 1)
    for i in range(M - 2):
        tempSoftBreak.set(isEquals.v[i] * isCr.v[i], "Q", i)
        isSoftBreak.set(tempSoftBreak.v[i] * isLf.v[i], "Q", i)
    isSoftBreak.set(0, "L", M - 2)
    isSoftBreak.set(0, "L", M - 1)
    for i in range(M):
        if i == 0:
            v = isSoftBreak.v[i]
        elif i == 1:
            v = isSoftBreak.v[i] + isSoftBreak.v[i - 1]
        elif i == M - 1:
            v = isSoftBreak.v[i - 1] + isSoftBreak.v[i - 2]
        else:
            v = isSoftBreak.v[i] + isSoftBreak.v[i - 1] + isSoftBreak.v[i - 2]
        shouldZero.set(v, "L", i)
    for i in range(M):
        processed.set((1 - shouldZero.v[i]) * encoded[i], "Q", i)
    m0 = Mux1(r, 1, shouldZero.v[0], "L")
    mux_slots[0][1] = m0
    rEnc.set(m0.o, "L", 0)
    for i in range(1, M):
        mi = Mux1(rEnc.v[i - 1] * r % P, rEnc.v[i - 1], shouldZero.v[i], "Q")
        mux_slots[i][1] = mi
        rEnc.set(mi.o, "L", i)
    c.subs = [tuple(x) if isinstance(x, list) else x for x in c.subs]
    rDec.set(r, "L", 0)
    for i in range(1, M):
        rDec.set(rDec.v[i - 1] * r, "Q", i)
    sumEnc.set(rEnc.v[0] * processed.v[0], "Q", 0)
    for i in range(1, M):
        sumEnc.set(sumEnc.v[i - 1] + rEnc.v[i] * processed.v[i], "Q", i)
    sumDec.set(rDec.v[0] * decoded[0], "Q", 0)
    for i in range(1, M):
        sumDec.set(sumDec.v[i - 1] + rDec.v[i] * decoded[i], "Q", i)
    fin = c.sub("anon_IsEqual_final", cl.IsEqual(sumEnc.v[M - 1], sumDec.v[M - 1]))
    c.o = isValid.set(fin.o, "L")
    return c


# ------------------------------------------------------------ email-verifier.circom


def EmailVerifier(maxHeadersLength, maxBodyLength, n, k, ignoreBodyHashCheck, inputs, body_hash_regex=None,
                  enableHeaderMasking=0, enableBodyMasking=0, removeSoftLineBreaks=0):
    """email-verifier.circom:42-174; main component, `public [ pubkey ]`
    (tests/test-circuits/email-verifier-test.circom:5).

    `inputs`: dict of integer lists/ints keyed by signal name.  `body_hash_regex(msg)` ->
    Comp with .o = (out, reveal0[]) stands for the [EXT] BodyHashRegex template."""
    assert maxHeadersLength % 64 == 0 and maxBodyLength % 64 == 0
    assert n * k > 2048 and n < (255 // 2)
    c = Comp(f"EmailVerifier({maxHeadersLength},{maxBodyLength},{n},{k},{ignoreBodyHashCheck},{enableHeaderMasking},{enableBodyMasking},{removeSoftLineBreaks})", is_main=True)
    c.public = {"pubkey"}
    emailHeader = [int(x) % P for x in inputs["emailHeader"]]
    emailHeaderLength = int(inputs["emailHeaderLength"]) % P
    pubkey = [int(x) % P for x in inputs["pubkey"]]
    signature = [int(x) % P for x in inputs["signature"]]
    # declaration order (email-verifier.circom:49-54, 70-71, 109-112)
    c.inp("emailHeader", maxHeadersLength).setall(emailHeader, "K")
    c.inp("emailHeaderLength").set(emailHeaderLength, "K")
    c.inp("pubkey", k).setall(pubkey, "K")
    c.inp("signature", k).setall(signature, "K")
    pubkeyHash = c.out("pubkeyHash")
    sha_sig = c.mid("sha", 256)
    shaHi = c.out("shaHi")
    shaLo = c.out("shaLo")

    n2bHeaderLength = c.sub("n2bHeaderLength", cl.Num2Bits(log2Ceil(maxHeadersLength), emailHeaderLength))
    c.sub("anon_AssertZeroPadding_header", AssertZeroPadding(maxHeadersLength, emailHeader, emailHeaderLength))
    shab = c.sub("anon_Sha256Bytes", Sha256Bytes(maxHeadersLength, emailHeader, emailHeaderLength))
    sha = shab.o
    sha_sig.setall(sha, "L")
    bitPacker = c.sub("bitPacker", PackBits(256, 128, sha))
    shaHi.set(bitPacker.o[0], "L")
    shaLo.set(bitPacker.o[1], "L")

    rsaMessageSize = (256 + n) // n
    rm_in = [[0] * n for _ in range(rsaMessageSize)]
    for i in range(256):
        rm_in[i // n][i % n] = sha[255 - i]
    rsaMessage = [c.sub(f"rsaMessage[{i}]", cl.Bits2Num(n, rm_in[i])) for i in range(rsaMessageSize)]
    message = [rsaMessage[i].o for i in range(rsaMessageSize)] + [0] * (k - rsaMessageSize)
    c.sub("rsaVerifier", RSAVerifier65537(n, k, message, signature, pubkey))

    if enableHeaderMasking == 1:   # email-verifier.circom:97-105
        headerMask = [int(x) % P for x in inputs["headerMask"]]
        c.inp("headerMask", maxHeadersLength).setall(headerMask, "K")
        maskedHeader = c.out("maskedHeader", maxHeadersLength)
        bm = c.sub("byteMask_header", ByteMask(maxHeadersLength, emailHeader, headerMask))
        maskedHeader.setall(bm.o, "L")

    if ignoreBodyHashCheck != 1:
        bodyHashIndex = int(inputs["bodyHashIndex"]) % P
        precomputedSHA = [int(x) % P for x in inputs["precomputedSHA"]]
        emailBody = [int(x) % P for x in inputs["emailBody"]]
        emailBodyLength = int(inputs["emailBodyLength"]) % P
        c.inp("bodyHashIndex").set(bodyHashIndex, "K")
        c.inp("precomputedSHA", 32).setall(precomputedSHA, "K")
        c.inp("emailBody", maxBodyLength).setall(emailBody, "K")
        c.inp("emailBodyLength").set(emailBodyLength, "K")
        c.sub("n2bBodyLength", cl.Num2Bits(log2Ceil(maxBodyLength), emailBodyLength))
        c.sub("anon_AssertZeroPadding_body", AssertZeroPadding(maxBodyLength, emailBody, emailBodyLength))
        rx = c.sub("anon_BodyHashRegex", body_hash_regex(emailHeader))
        bhRegexMatch = c.mid("bhRegexMatch").set(rx.o[0], "L")
        bhReveal = c.mid("bhReveal", maxHeadersLength)
        bhReveal.setall(rx.o[1], "L")
        c.eq(bhRegexMatch, 1, "bhRegexMatch === 1 (email-verifier.circom:127)")
        shaB64Length = 44
        sel = c.sub("anon_SelectRegexReveal", SelectRegexReveal(maxHeadersLength, shaB64Length, bhReveal.v, bodyHashIndex))
        c.mid("bhBase64", shaB64Length).setall(sel.o, "L")
        b64 = c.sub("anon_Base64Decode", Base64Decode(32, sel.o))
        c.mid("headerBodyHash", 32).setall(b64.o, "L")
        shap = c.sub("anon_Sha256BytesPartial", Sha256BytesPartial(maxBodyLength, emailBody, emailBodyLength, precomputedSHA))
        c.mid("computedBodyHash", 256).setall(shap.o, "L")
        for i in range(32):
            bits = [0] * 8
            for j in range(8):
                bits[7 - j] = shap.o[i * 8 + j]
            b2n = c.sub(f"computedBodyHashInts[{i}]", cl.Bits2Num(8, bits))
            c.eq(b2n.o, b64.o[i], "computedBodyHashInts[i].out === headerBodyHash[i] (email-verifier.circom:145)")
        if removeSoftLineBreaks == 1:   # email-verifier.circom:148-156
            decodedEmailBodyIn = [int(x) % P for x in inputs["decodedEmailBodyIn"]]
            c.inp("decodedEmailBodyIn", maxBodyLength).setall(decodedEmailBodyIn, "K")
            qp = c.sub("qpEncodingChecker", RemoveSoftLineBreaks(maxBodyLength, emailBody, decodedEmailBodyIn))
            c.eq(qp.o, 1, "qpEncodingChecker.isValid === 1 (email-verifier.circom:155)")
        if enableBodyMasking == 1:   # email-verifier.circom:158-166
            bodyMask = [int(x) % P for x in inputs["bodyMask"]]
            c.inp("bodyMask", maxBodyLength).setall(bodyMask, "K")
            maskedBody = c.out("maskedBody", maxBodyLength)
            bmb = c.sub("byteMask_body", ByteMask(maxBodyLength, emailBody, bodyMask))
            maskedBody.setall(bmb.o, "L")

    ph = c.sub("anon_PoseidonLarge", PoseidonLarge(n, k, pubkey))
    pubkeyHash.set(ph.o, "L")
    c.o = (pubkeyHash.v[0], shaHi.v[0], shaLo.v[0])
    return c


# ------------------------------------------------------------ BodyHashRegex [EXT]
import re as _re

# @zk-email/zk-regex-circom 2.3.2, circuits/common/body_hash_regex.circom (source absent,
# yarn.lock:2794-2802).  Regex compiled by zk-regex (email-verifier.circom:126 call site):
#   (\r\n|^)dkim-signature:([a-z]+=[^;]+; )+bh=[a-zA-Z0-9+/=]+;      public part: the bh value
# zk-regex feeds the DFA the byte 255 in front of the message to stand for `^`.
_BH_RE = _re.compile(rb"(?:\r\n|\xff)dkim-signature:(?:[a-z]+=[^;]+; )+bh=([a-zA-Z0-9+/=]+);")


def BodyHashRegex(msg_bytes, msg):
    """Interface-level restatement: `out` = number of matches != 0, `reveal0[i]` = msg[i] inside
    the public (bh value) part of a match, else 0.  The DFA-internal signals of the generated
    circuit are NOT restated (parity unpinned); only reveal0 (quadratic:
    `reveal0[i] <== in[i+1] * is_reveal0[i]`) is part of the kept layout."""
    c = Comp(f"BodyHashRegex({msg_bytes})")
    out = c.out("out")
    reveal0 = c.out("reveal0", msg_bytes)
    c.inp("msg", msg_bytes).setall(msg, "L")
    data = b"\xff" + bytes(int(x) & 0xFF for x in msg)
    rev = [0] * msg_bytes
    n = 0
    pos = 0
    while True:
        m = _BH_RE.search(data, pos)
        if not m:
            break
        n += 1
        for i in range(m.start(1), m.end(1)):
            rev[i - 1] = data[i]
        pos = m.end()
    out.set(1 if n else 0, "L")
    reveal0.setall(rev, "Q")
    c.o = (out.v[0], reveal0.v)
    return c


# ------------------------------------------------------------ BodyHashRegex, DFA circuit (zkwg v1)
import json as _json
import os as _os

_DFA = None


def bh_dfa():
    global _DFA
    if _DFA is None:
        p = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "zk-email-verify_amd", "data", "bh_dfa.json")
        _DFA = _json.load(open(p))
    return _DFA


def MultiOR(n, ins):
    """[EXT] zk-regex regex_helpers.circom MultiOR(n): sums (linear); is_zero = IsZero(sum); out <== 1 - is_zero.out."""
    c = Comp(f"MultiOR({n})")
    out = c.out("out")
    c.inp("in", n).setall(ins, "L")
    sums = c.mid("sums", n)
    acc = 0
    for i in range(n):
        acc = (acc + ins[i]) % P
        sums.set(acc, "L", i)
    isz = c.sub("is_zero", cl.IsZero(acc))
    c.o = out.set(1 - isz.o, "L")
    return c


def MultiNOR(n, ins):
    """[EXT] regex_helpers.circom MultiNOR(n): out <== IsZero(sum).out."""
    c = Comp(f"MultiNOR({n})")
    out = c.out("out")
    c.inp("in", n).setall(ins, "L")
    isz = c.sub("is_zero", cl.IsZero(sum(ins) % P))
    c.o = out.set(isz.o, "L")
    return c


def BodyHashRegexV1(msg_bytes, msg):
    """zkwg's own DFA circuit for the body-hash regex, in the style of zk-regex's generated circom
    (the real generated file is [EXT] and absent -- parity unpinned).  Tables:
    zk-email-verify_amd/data/bh_dfa.json (tools/gen_bh_dfa.py).

    in[0] = 255 (stands for `^`), in[i+1] = msg[i].  State 0 is permanently active; a transition out of
    state 0 fires only when no other state continues (from_zero_enabled).  Component arrays are declared
    [k][num_bytes] (kind-major), like zk-regex's eq/lt/and arrays.
    """
    T = bh_dfa()
    S, ACC = T["n_states"], T["accept"]
    prims, classes, trans, public = T["prims"], T["classes"], T["transitions"], T["public"]
    nb = msg_bytes + 1
    c = Comp(f"BodyHashRegex({msg_bytes})")
    out = c.out("out")
    reveal0 = c.out("reveal0", msg_bytes)
    c.inp("msg", msg_bytes).setall(msg, "L")
    in_sig = c.mid("in", nb)
    in_ = [255] + [int(x) % P for x in msg]
    in_sig.setall(in_, "L")
    st_sig = c.mid("states", (nb + 1) * S)
    tmp_sig = c.mid("states_tmp", (nb + 1) * S)
    fze_sig = c.mid("from_zero_enabled", nb + 1)
    live_c1 = c.mid("live_c1", nb)
    live_t = c.mid("live_t", nb)
    live = c.mid("live", nb + 2)
    prev_sig = c.mid("prev_states0", len(public) * msg_bytes)
    substr_sig = c.mid("is_substr0", msg_bytes)
    isrev_sig = c.mid("is_reveal0", msg_bytes)

    eq_idx = [k for k, p in enumerate(prims) if p[0] == "eq"]
    rg_idx = [k for k, p in enumerate(prims) if p[0] == "range"]
    multi_cls = [k for k, cdef in enumerate(classes) if len(cdef["members"]) > 1]
    incoming = {d: [] for d in range(S)}
    for t_i, (f, to, cid) in enumerate(trans):
        incoming[to].append(t_i)
    tmp_multi = [d for d in range(1, S) if len([t for t in incoming[d] if trans[t][0] != 0]) > 1]
    st_multi = [d for d in range(1, S) if [t for t in incoming[d] if trans[t][0] == 0] and [t for t in incoming[d] if trans[t][0] != 0]]

    eq = {k: [None] * nb for k in eq_idx}
    lt = {k: [None] * nb for k in rg_idx}        # (lo-side, hi-side)
    and_rng = {k: [None] * nb for k in rg_idx}
    cls_or = {k: [None] * nb for k in multi_cls}
    and_t = [[None] * nb for _ in trans]
    tmp_or = {d: [None] * nb for d in tmp_multi}
    fze_c = [None] * nb
    st_or = {d: [None] * nb for d in st_multi}

    states = [[0] * S for _ in range(nb + 1)]
    tmps = [[0] * S for _ in range(nb + 1)]
    fze = [0] * (nb + 1)
    for j in range(nb + 1):
        states[j][0] = 1
    for i in range(nb):
        b = in_[i]
        pv = {}
        for k in eq_idx:
            eq[k][i] = cl.IsEqual(b, prims[k][1])
            pv[k] = eq[k][i].o
        for k in rg_idx:
            lo, hi = prims[k][1], prims[k][2]
            a = cl.LessThan(8, lo - 1, b)
            bb = cl.LessThan(8, b, hi + 1)
            lt[k][i] = (a, bb)
            and_rng[k][i] = cl.AND(a.o, bb.o)
            pv[k] = and_rng[k][i].o
        cv = []
        for k, cdef in enumerate(classes):
            if len(cdef["members"]) > 1:
                cls_or[k][i] = MultiOR(len(cdef["members"]), [pv[m] for m in cdef["members"]])
                v = cls_or[k][i].o
            else:
                v = pv[cdef["members"][0]]
            cv.append((1 - v) % P if cdef["neg"] else v)
        # transitions from non-zero states
        for t_i, (f, to, cid) in enumerate(trans):
            if f != 0:
                and_t[t_i][i] = cl.AND(states[i][f], cv[cid])
        for d in range(1, S):
            inc = [t for t in incoming[d] if trans[t][0] != 0]
            if len(inc) == 0:
                tmps[i + 1][d] = 0
            elif len(inc) == 1:
                tmps[i + 1][d] = and_t[inc[0]][i].o
            else:
                tmp_or[d][i] = MultiOR(len(inc), [and_t[t][i].o for t in inc])
                tmps[i + 1][d] = tmp_or[d][i].o
        fze_c[i] = MultiNOR(S - 1, tmps[i + 1][1:])
        fze[i] = fze_c[i].o
        for t_i, (f, to, cid) in enumerate(trans):
            if f == 0:
                and_t[t_i][i] = cl.AND(fze[i], cv[cid])
        for d in range(1, S):
            z = [t for t in incoming[d] if trans[t][0] == 0]
            if not z:
                states[i + 1][d] = tmps[i + 1][d]
            elif d in st_multi:
                st_or[d][i] = MultiOR(2, [tmps[i + 1][d], and_t[z[0]][i].o])
                states[i + 1][d] = st_or[d][i].o
            else:
                states[i + 1][d] = and_t[z[0]][i].o
    fze[nb] = 0
    is_accepted = MultiOR(nb + 1, [states[j][ACC] for j in range(nb + 1)])
    out.set(is_accepted.o, "L")
    # live[j]: the thread that is in states[j] reaches the accept state without restarting
    lv = [0] * (nb + 2)
    for j in range(nb, 0, -1):
        c1 = (lv[j + 1] * (1 - fze[j])) % P if j < nb else 0
        live_c1.set(c1, "Q", j - 1)
        tt = ((1 - states[j][ACC]) * c1) % P
        live_t.set(tt, "Q", j - 1)
        lv[j] = (states[j][ACC] + tt) % P
    live.setall(lv, "L")
    rev = [0] * msg_bytes
    substr_or = [None] * msg_bytes
    for i in range(msg_bytes):
        pvals = []
        for k, (s_, d_) in enumerate(public):
            v = states[i + 1][s_] * states[i + 2][d_]
            prev_sig.set(v, "Q", k * msg_bytes + i)
            pvals.append(v)
        substr_or[i] = MultiOR(len(public), pvals)
        substr_sig.set(substr_or[i].o, "L", i)
        ir = (substr_or[i].o * lv[i + 2]) % P
        isrev_sig.set(ir, "Q", i)
        rev[i] = (in_[i + 1] * ir) % P
    reveal0.setall(rev, "Q")
    flat = []
    for j in range(nb + 1):
        flat += states[j]
    st_sig.setall(flat, "L")
    flat = []
    for j in range(nb + 1):
        flat += tmps[j]
    tmp_sig.setall(flat, "L")
    fze_sig.setall(fze, "L")
    # sub-components in declaration order, kind-major like zk-regex's `component eq[..][num_bytes]`
    for n_, k in enumerate(eq_idx):
        for i in range(nb):
            c.sub(f"eq[{n_}][{i}]", eq[k][i])
    for n_, k in enumerate(rg_idx):
        for i in range(nb):
            c.sub(f"lt[{2 * n_}][{i}]", lt[k][i][0])
        for i in range(nb):
            c.sub(f"lt[{2 * n_ + 1}][{i}]", lt[k][i][1])
    for n_, k in enumerate(rg_idx):
        for i in range(nb):
            c.sub(f"and_rng[{n_}][{i}]", and_rng[k][i])
    for n_, k in enumerate(multi_cls):
        for i in range(nb):
            c.sub(f"cls_or[{n_}][{i}]", cls_or[k][i])
    for t_i in range(len(trans)):
        for i in range(nb):
            c.sub(f"and[{t_i}][{i}]", and_t[t_i][i])
    for n_, d in enumerate(tmp_multi):
        for i in range(nb):
            c.sub(f"tmp_or[{n_}][{i}]", tmp_or[d][i])
    for i in range(nb):
        c.sub(f"fze[{i}]", fze_c[i])
    for n_, d in enumerate(st_multi):
        for i in range(nb):
            c.sub(f"st_or[{n_}][{i}]", st_or[d][i])
    c.sub("is_accepted", is_accepted)
    for i in range(msg_bytes):
        c.sub(f"substr_or[{i}]", substr_or[i])
    c.o = (out.v[0], reveal0.v)
    return c
