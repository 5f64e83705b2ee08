"""Literal transcription of packages/circuits/lib/bigint-func.circom (test infrastructure).

circom `var` arithmetic is arithmetic in the BN254 scalar field with the
expression semantics of SURVEY.md Appendix A.5:
  * values are representatives in [0, p);
  * relational operators compare the signed lift (z > p/2 -> z - p);
  * `\\` and `%` are integer quotient / remainder on the representatives;
  * `/` multiplies by the modular inverse; `**` is modular power;
  * `x >> k` = x \\ 2^k ; `x << k` = (x * 2^k & (2^254 - 1)) mod p ; `&` acts on
    representatives.
The helpers below implement exactly those operators and the functions are
transcribed statement by statement, with the reference line cited.
"""
from .comp import P, AssertFailed

_MASK = (1 << 254) - 1


def F(x):
    return x % P


def lift(x):
    x %= P
    return x - P if x > P // 2 else x


def shl(x, k):
    return ((x % P) << k & _MASK) % P


def shr(x, k):
    return (x % P) >> k


def idiv(a, b):
    return (a % P) // (b % P)


def imod(a, b):
    return (a % P) % (b % P)


def fdiv(a, b):
    return (a % P) * pow(b % P, P - 2, P) % P


def fpow(a, e):
    return pow(a % P, e % P, P)


def _assert(cond, where):
    if not cond:
        raise AssertFailed("bigint-func.circom " + where)


def div_ceil(m, n):  # bigint-func.circom:4-12
    if m % n == 0:
        return m // n
    return m // n + 1


def log_ceil(n):  # bigint-func.circom:14-23
    n_temp = n
    for i in range(254):
        if n_temp == 0:
            return i
        n_temp = n_temp // 2
    return 254


def getProperRepresentation(m, n, k, in_):  # bigint-func.circom:32-53
    ceilMN = div_ceil(m, n)
    out = [0] * 100
    _assert(k + ceilMN < 100, ":36")
    for i in range(k):
        out[i] = F(in_[i])
    _assert(n <= m, ":43")
    i = 0
    while i + 1 < k + ceilMN:
        _assert(lift(shl(1, m)) >= lift(out[i]) and lift(out[i]) >= lift(F(-shl(1, m))), ":45")
        shifted_val = F(out[i] + shl(1, m))
        _assert(0 <= lift(shifted_val) and lift(shifted_val) <= lift(shl(1, m + 1)), ":47")
        out[i] = shifted_val & (shl(1, n) - 1)
        out[i + 1] = F(out[i + 1] + shr(shifted_val, n) - shl(1, m - n))
        i += 1
    return out


def poly_eval(len_, a, x):  # bigint-func.circom:56-62
    v = 0
    for i in range(len_):
        v = F(v + a[i] * fpow(x, i))
    return v


def poly_interp(len_, v):  # bigint-func.circom:65-103
    _assert(len_ <= 200, ":66")
    out = [0] * 200
    full_poly = [0] * 201
    full_poly[0] = 1
    for i in range(len_):
        full_poly[i + 1] = 0
        for j in range(i, -1, -1):
            full_poly[j + 1] = F(full_poly[j + 1] + full_poly[j])
            full_poly[j] = F(full_poly[j] * F(-i))
    for i in range(len_):
        cur_v = 1
        for j in range(len_):
            if i == j:
                pass
            else:
                cur_v = F(cur_v * F(i - j))
        cur_v = fdiv(v[i], cur_v)
        cur_rem = full_poly[len_]
        for j in range(len_ - 1, -1, -1):
            out[j] = F(out[j] + cur_v * cur_rem)
            cur_rem = F(full_poly[j] + i * cur_rem)
        _assert(cur_rem == 0, ":99")
    return out


def long_gt(n, k, a, b):  # bigint-func.circom:106-116
    for i in range(k - 1, -1, -1):
        if lift(a[i]) > lift(b[i]):
            return 1
        if lift(a[i]) < lift(b[i]):
            return 0
    return 0


def long_sub(n, k, a, b):  # bigint-func.circom:122-145
    diff = [0] * 100
    borrow = [0] * 100
    for i in range(k):
        if i == 0:
            if lift(a[i]) >= lift(b[i]):
                diff[i] = F(a[i] - b[i])
                borrow[i] = 0
            else:
                diff[i] = F(a[i] - b[i] + shl(1, n))
                borrow[i] = 1
        else:
            if lift(a[i]) >= lift(F(b[i] + borrow[i - 1])):
                diff[i] = F(a[i] - b[i] - borrow[i - 1])
                borrow[i] = 0
            else:
                diff[i] = F(shl(1, n) + a[i] - b[i] - borrow[i - 1])
                borrow[i] = 1
    return diff


def long_scalar_mult(n, k, a, b):  # bigint-func.circom:149-160
    out = [0] * 100
    for i in range(k):
        temp = F(out[i] + a * b[i])
        out[i] = imod(temp, shl(1, n))
        out[i + 1] = F(out[i + 1] + idiv(temp, shl(1, n)))
    return out


def short_div_norm(n, k, a, b):  # bigint-func.circom:225-242
    qhat = idiv(F(a[k] * shl(1, n) + a[k - 1]), b[k - 1])
    if lift(qhat) > lift(F(shl(1, n) - 1)):
        qhat = F(shl(1, n) - 1)
    mult = long_scalar_mult(n, k, qhat, b)
    if long_gt(n, k + 1, mult, a) == 1:
        mult = long_sub(n, k + 1, mult, b)
        if long_gt(n, k + 1, mult, a) == 1:
            return F(qhat - 2)
        else:
            return F(qhat - 1)
    else:
        return qhat


def short_div(n, k, a, b):  # bigint-func.circom:249-264
    scale = idiv(shl(1, n), F(1 + b[k - 1]))
    norm_a = long_scalar_mult(n, k + 1, scale, a)
    norm_b = long_scalar_mult(n, k, scale, b)
    if norm_b[k] != 0:
        ret = short_div_norm(n, k + 1, norm_a, norm_b)
    else:
        ret = short_div_norm(n, k, norm_a, norm_b)
    return ret


def long_div(n, k, m, a, b):  # bigint-func.circom:169-218
    out = [[0] * 100, [0] * 100]
    b = list(b) + [0] * (100 - len(b))
    a = list(a) + [0] * (200 - len(a))
    m += k
    while b[k - 1] == 0:
        out[1][k] = 0
        k -= 1
        _assert(k > 0, ":175")
    m -= k

    remainder = [0] * 200
    for i in range(m + k):
        remainder[i] = a[i]

    dividend = [0] * 200
    for i in range(m, -1, -1):
        if i == m:
            dividend[k] = 0
            for j in range(k - 1, -1, -1):
                dividend[j] = remainder[j + m]
        else:
            for j in range(k, -1, -1):
                dividend[j] = remainder[j + i]

        out[0][i] = short_div(n, k, dividend, b)

        mult_shift = long_scalar_mult(n, k, out[0][i], b)
        subtrahend = [0] * 200
        for j in range(k + 1):
            if i + j < m + k:
                subtrahend[i + j] = mult_shift[j]
        remainder = long_sub(n, m + k, remainder, subtrahend) + [0] * 100
    for i in range(k):
        out[1][i] = remainder[i]
    out[1][k] = 0
    return out
