// FpMul(n, k) for small parameters (n k <= 62) -- lib/fp.circom:16-81 with lib/bigint.circom:16-60 (BigLessThan) and
// :69-94 (CheckCarryToZero) -- the generic-parameter main `component main = FpMul(2, 4)` of
// packages/circuits/tests/test-circuits/fp-mul-test.circom:5 (tests/fp-mul.test.ts:34-46).  The RSA path keeps its
// (121, 17) wavefront algorithms (zkwg_rsa_wave.h); this is the plain one-lane restatement for parameters whose numbers
// fit machine words: a, b, p < 2^62, a b < 2^124.  Compiled for the device (zk_fpmul_small) and for the host
// (tests/native/hosttest.cpp).
#pragma once
#include "zkwg_sched.h"
#include "zkwg_fr.h"

ZK_HD u32 zk_bit_length(u64 a) { u32 r = 0; while (a) { ++r; a >>= 1; } return r; }   // log_ceil of lib/bigint-func.circom:14-23

// 128 x 128 -> 256 bits
ZK_HD Fr zk_mul128(unsigned __int128 a, unsigned __int128 b) {
  const u64 a0 = (u64)a, a1 = (u64)(a >> 64), b0 = (u64)b, b1 = (u64)(b >> 64);
  u64 l00, h00, l01, h01, l10, h10, l11, h11;
  zk_mul64(a0, b0, l00, h00); zk_mul64(a0, b1, l01, h01); zk_mul64(a1, b0, l10, h10); zk_mul64(a1, b1, l11, h11);
  Fr r;
  r.l[0] = l00;
  u64 c = 0, c2 = 0;
  r.l[1] = zk_adc(h00, l01, c); r.l[1] = zk_adc(r.l[1], l10, c2);
  u64 c3 = 0, c4 = 0;
  r.l[2] = zk_adc(h01, h10, c3); r.l[2] = zk_adc(r.l[2], l11, c4);
  u64 c5 = 0;
  r.l[2] = zk_adc(r.l[2], c + c2, c5);
  r.l[3] = h11 + c3 + c4 + c5;
  return r;
}
// a signed integer as a field element
ZK_HD Fr zk_fr_of_i128(__int128 v) {
  if (v >= 0) return Fr{{(u64)v, (u64)((unsigned __int128)v >> 64), 0, 0}};
  const unsigned __int128 m = (unsigned __int128)(-v);
  u64 bw;
  return fr_sub_raw(fr_p(), Fr{{(u64)m, (u64)(m >> 64), 0, 0}}, bw);
}

// One email: image values of main = FpMul(n, k).  Returns the status (0, or ZKWG_ERR_ASSERT_FAILED = 4 when a chunk
// does not fit n bits, p = 0 or the quotient does not fit k chunks -- inputs for which the template's own constraints fail).
ZK_HD int zk_fpmul_small_core(const ZkFpgLayout& L, u32 m_one, const u8* rec, u64* bits, u32* small, Fr* frv) {
  const u32 n = L.n, k = L.k;
  typedef unsigned __int128 u128;
  u64 a[17], b[17], p[17], q[17], r[17];
  int st = 0;
  u64 A = 0, B = 0, Pm = 0;
  for (u32 i = 0; i < k; ++i) {
    const u64* la = (const u64*)(rec + L.in_a + 16u * i);
    const u64* lb = (const u64*)(rec + L.in_b + 16u * i);
    const u64* lp = (const u64*)(rec + L.in_p + 16u * i);
    a[i] = la[0]; b[i] = lb[0]; p[i] = lp[0];
    if (la[1] | lb[1] | lp[1] | (a[i] >> n) | (b[i] >> n) | (p[i] >> n)) st = 4;
    A |= (a[i] & ((1ull << n) - 1)) << (n * i); B |= (b[i] & ((1ull << n) - 1)) << (n * i); Pm |= (p[i] & ((1ull << n) - 1)) << (n * i);
  }
  if (Pm == 0) { st = 4; Pm = 1; }
  // a b = q p + r (long_div of lib/bigint-func.circom:93-140 on numbers that fit machine words)
  u64 lo, hi;
  zk_mul64(A, B, lo, hi);
  u128 rem = 0, quo = 0;
  for (int i = 127; i >= 0; --i) {
    rem = (rem << 1) | (u128)(((i >= 64 ? hi >> (i - 64) : lo >> i)) & 1ull);
    quo <<= 1;
    if (rem >= (u128)Pm) { rem -= Pm; quo |= 1; }
  }
  if (quo >> (n * k)) st = 4;
  for (u32 i = 0; i < k; ++i) {
    q[i] = (u64)(quo >> (n * i)) & ((1ull << n) - 1);
    r[i] = (u64)((u64)rem >> (n * i)) & ((1ull << n) - 1);
  }
  small[m_one] = 1;
  for (u32 i = 0; i < k; ++i) small[L.m_out + i] = (u32)r[i];
  // v_ab[x] = a(x) b(x), v_pq_r[x] = p(x) q(x) + r(x) at x = 0 .. 2k-2 (fp.circom:25-30, 60-66): below 2^232, no reduction
  for (u32 x = 0; x < 2 * k - 1; ++x) {
    u128 ea = 0, eb = 0, ep = 0, eq = 0, er = 0, pw = 1;
    for (u32 i = 0; i < k; ++i) { ea += pw * a[i]; eb += pw * b[i]; ep += pw * p[i]; eq += pw * q[i]; er += pw * r[i]; pw *= x; }
    frv[L.f_main + x] = zk_mul128(ea, eb);
    u64 c;
    frv[L.f_main + 4 * k - 1 + x] = fr_add_raw(zk_mul128(ep, eq), Fr{{(u64)er, (u64)(er >> 64), 0, 0}}, c);
  }
  for (u32 i = 0; i < k; ++i) {
    frv[L.f_main + 2 * k - 1 + i] = fr_from_u64(q[i]);
    frv[L.f_main + 3 * k - 1 + i] = fr_from_u64(r[i]);
    bits[L.b_qr + i] = q[i];          // q_range_check[i] = Num2Bits(n): the chunk's bits
    bits[L.b_qr + k + i] = r[i];
  }
  // r_p_lt_check = BigLessThan(n, k)(r, p) (bigint.circom:16-60)
  u32 lt[17], eqv[17];
  for (u32 i = 0; i < k; ++i) {
    bits[L.b_lt + i] = r[i] + (1ull << n) - p[i];      // LessThan(n): Num2Bits(n + 1) of a + 2^n - b
    lt[i] = r[i] < p[i]; eqv[i] = r[i] == p[i];
    const __int128 d = (__int128)p[i] - (__int128)r[i];   // IsEqual: isz.in = in[1] - in[0] (circomlib comparators.circom)
    frv[L.f_eq + 2 * i] = fr_from_u64(eqv[i]);
    Fr inv = fr_zero();
    if (d != 0) {
      const Fr m = fr_from_mont(fr_mont_inv(fr_to_mont(fr_from_u64((u64)(d < 0 ? -d : d)))));
      inv = d < 0 ? fr_neg(m) : m;
    }
    frv[L.f_eq + 2 * i + 1] = inv;
  }
  {
    u32 ors = 0, eq_ands = 0;
    for (int i = (int)k - 2; i >= 0; --i) {
      u32 ands;
      if (i == (int)k - 2) { ands = eqv[k - 1] & lt[k - 2]; eq_ands = eqv[k - 1] & eqv[k - 2]; ors = lt[k - 1] | ands; }
      else { ands = eq_ands & lt[i]; eq_ands = eq_ands & eqv[i]; ors = ors | ands; }
      small[L.m_gates + i] = ors; small[L.m_gates + (k - 1) + i] = ands; small[L.m_gates + 2 * (k - 1) + i] = eq_ands;
    }
  }
  // tCheck = CheckCarryToZero(n, 2n + log_ceil(k) + 2, 2k - 1) on the coefficients of a b - p q - r (fp.circom:68-77)
  const u32 cb = n + zk_bit_length(k) + 5;
  __int128 carry = 0;
  for (u32 i = 0; i + 1 < 2 * k - 1; ++i) {
    __int128 t = 0;
    for (u32 j = 0; j < k; ++j)
      if (i >= j && i - j < k) t += (__int128)((u128)a[j] * b[i - j]) - (__int128)((u128)p[j] * q[i - j]);
    if (i < k) t -= (__int128)r[i];
    carry = (t + carry) >> n;                          // exact: a b = p q + r as integers
    frv[L.f_carry + i] = zk_fr_of_i128(st ? 0 : carry);
    bits[L.b_carry + i] = st ? 0 : (u64)(carry + ((__int128)1 << (cb - 1)));
  }
  frv[L.f_carry + 2 * k - 2] = fr_zero();              // declared, never assigned (bigint.circom:76)
  return st;
}
