// BN254 base field Fq in LAZY 9 x 29-bit limb form (Montgomery constant 2^261) -- the arithmetic of the prover's multi-exponentiation
// kernels since round 6 (zkwg_ec29.h builds the G1 / G2 point formulas on it; zkwg_msm_core.h the kernels).
//
// Why (tools/mulbench.hip): on gfx950 v_mad_u64_u32 issues at nearly the plain VALU rate, so a Montgomery product is
// bound by its 162 multiply-adds only when nothing surrounds them.  With 29-bit limbs a column of the schoolbook product (<= 9 products of
// the operands + 9 of the reduction) fits a 64-bit accumulator: no carries inside a product (250 instructions, 1.7 x the 8 x 32-bit CIOS).
// Round 5 used that product behind the 4 x 64-bit interface (split / pack / conditional subtraction around every product: ~315
// instructions, and every addition / subtraction a carry chain + compare + select).  Here values STAY in limb form across a whole point
// formula: an addition is 9 independent adds, a subtraction adds a multiple of q whose limbs dominate the subtrahend's (9 adds + 9
// subs, no borrow), and only three values of a mixed addition are carry-normalised (24 dependent adds / shifts each).
//
// A value is described by two bounds, written [U, V] in the comments: every limb < U 2^29 (the top limb, l[8], is only ever bounded by the
// value), and the value < V q.  Rules:
//   fq29_mul(a, b)       needs Ua Ub <= 6 (9 Ua Ub 2^58 + 9 2^58 < 2^64) and Va Vb <= 169 (then the result is < 2 q); returns [1, Va Vb / 169 + 1]
//   fq29_dot2(a,b,c,d)   a b + c d with ONE reduction: needs Ua Ub + Uc Ud <= 6; returns [1, (Va Vb + Vc Vd) / 169 + 1]   (169 = 2^261 / q)
//   fq29_add             [Ua + Ub, Va + Vb]
//   fq29_sub<M, U>(a, b) a + negc<M, U> - b for b = [U, <= M - 1]; returns [Ua + U + 1, Va + M]
//   fq29_norm            exact carry propagation: [1, V]
// The host build can count every violated precondition (ZKWG_FQ29_CHECK: tests/native/hosttest.cpp, asserted zero by tests/test_ec29_cpu.py).
#pragma once
#include "zkwg_fq.h"

struct Fq29 {
  u32 l[9];
};
#define ZKQ29_M 0x1fffffffu
#define ZKQ29_N0 0x04866389u     // -q^-1 mod 2^29
#define ZKQ29_PINV 0x1b799c77u   // q^-1 mod 2^29

#if !defined(__HIP_DEVICE_COMPILE__) && defined(ZKWG_FQ29_CHECK)
static unsigned long long zk_fq29_violations = 0;
#define ZKQ29_EXPECT(cond) do { if (!(cond)) ++zk_fq29_violations; } while (0)
#else
#define ZKQ29_EXPECT(cond) do { } while (0)
#endif

// the limbs of q, and a few constants in limb form (Python: oracle/pyref/bn254_g1.Q)
#define ZKQ29_P(i) ((i) == 0 ? 0x187cfd47u : (i) == 1 ? 0x010460b6u : (i) == 2 ? 0x1c72a34fu : (i) == 3 ? 0x02d522d0u : (i) == 4 ? 0x1585d978u : (i) == 5 ? 0x02db40c0u : (i) == 6 ? 0x00a6e141u : (i) == 7 ? 0x0e5c2634u : 0x0030644eu)
ZK_HD Fq29 fq29_zero() { return Fq29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
ZK_HD Fq29 fq29_one() { return Fq29{{0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u}}; }      // 2^261 mod q
ZK_HD Fq29 fq29_r256() { return Fq29{{0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u}}; }     // 2^256 mod q: x 2^261 -> x 2^256
ZK_HD Fq29 fq29_t266() { return Fq29{{0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u}}; }     // 2^266 mod q: x 2^256 -> x 2^261

// 4 x 64-bit words (value < 2^256) <-> limbs
ZK_HD Fq29 fq29_split(const u64 x[4]) {
  Fq29 r;
  zk29_split<0>(x, r.l);
  r.l[8] = (u32)(x[3] >> 40);      // (zk29_split masks the top limb to 29 bits: 24 bits are there)
  return r;
}
ZK_HD Fq29 fq29_from_fq(const Fq& a) { return fq29_split(a.l); }
// exact carry propagation: limbs 0 .. 7 < 2^29, the value unchanged (it must stay below 2^261: V <= 169)
ZK_HD Fq29 fq29_norm(const Fq29& a) {
  Fq29 r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32 t = a.l[i] + c;          // limbs < 2^32 - 2^3: callers keep U <= 7
    ZKQ29_EXPECT(t >= c);
    r.l[i] = t & ZKQ29_M;
    c = t >> 29;
  }
  r.l[8] = a.l[8] + c;
  ZKQ29_EXPECT(r.l[8] >= c);
  return r;
}
// k q in plain limbs (k small), for the exact comparisons of the slow paths
ZK_HD Fq29 fq29_kp(u32 k) {
  Fq29 r;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    c += (u64)k * ZKQ29_P(i);
    r.l[i] = i < 8 ? ((u32)c & ZKQ29_M) : (u32)c;
    c >>= 29;
  }
  return r;
}
// value == 0 mod q for a = [*, <= V]?  The filter costs three instructions: the low 29 bits of limb 0 are exact whatever the other
// limbs carry, and k q = those bits (mod 2^29) for exactly one k < 2^29 -- a value that is no multiple of q passes with probability
// (V + 1) 2^-29, and then the exact comparison decides.
template <int V>
ZK_HD bool fq29_maybe_zero(const Fq29& a) { return ((a.l[0] * ZKQ29_PINV) & ZKQ29_M) <= (u32)V; }
template <int V>
ZK_HD bool fq29_is_zero_mod(const Fq29& a) {
  const u32 k = (a.l[0] * ZKQ29_PINV) & ZKQ29_M;
  if (k > (u32)V) return false;
  const Fq29 n = fq29_norm(a), kp = fq29_kp(k);
  u32 d = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) d |= n.l[i] ^ kp.l[i];
  return d == 0;
}
ZK_HD bool fq29_all_zero(const Fq29& a) {
  u32 d = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) d |= a.l[i];
  return d == 0;
}
// canonical: the value reduced below q, as 4 x 64-bit words (a = [*, <= V], V <= 40)
template <int V>
ZK_HD Fq fq29_to_fq(const Fq29& a) {
  Fq29 n = fq29_norm(a);
  // subtract q while >= q: binary steps 32 q, 16 q, ... q
#pragma unroll
  for (int s = 5; s >= 0; --s) {
    if ((1 << s) > V) continue;
    const Fq29 kp = fq29_kp(1u << s);
    // n >= kp ?
    bool ge = true;
#pragma unroll
    for (int i = 8; i >= 0; --i) {
      if (n.l[i] != kp.l[i]) { ge = n.l[i] > kp.l[i]; break; }
    }
    if (ge) {
      u32 bw = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const u32 t = n.l[i] - kp.l[i] - bw;
        bw = i < 8 ? (t >> 31) : 0;                // limbs < 2^29: a borrow shows in the top bit
        n.l[i] = i < 8 ? (t & ZKQ29_M) : t;
      }
    }
  }
  u64 w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, k = bit >> 6, s = bit & 63;
    w[k] |= (u64)n.l[i] << s;
    if (s > 64 - 29 && k + 1 < 4) w[k + 1] |= (u64)n.l[i] >> (64 - s);
  }
  return Fq{{w[0], w[1], w[2], w[3]}};
}

// ---- products ------------------------------------------------------------------------------------------------------------------------
ZK_HD Fq29 fq29_mul(const Fq29& a, const Fq29& b) {
  u32 q[9];
  Fq29 r;
  u64 acc = 0;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(ZKWG_FQ29_CHECK)
  { unsigned __int128 worst = 0; for (int k = 0; k < 17; ++k) { unsigned __int128 s = (unsigned __int128)9 << 58; for (int i = 0; i < 9; ++i) if (k - i >= 0 && k - i < 9) s += (unsigned __int128)a.l[i] * b.l[k - i]; if (s > worst) worst = s; } ZKQ29_EXPECT(worst < ((unsigned __int128)1 << 64) - ((unsigned __int128)1 << 36)); }
#endif
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (u64)q[i] * ZKQ29_P(k - i);
    q[k] = ((u32)acc * ZKQ29_N0) & ZKQ29_M;
    acc += (u64)q[k] * ZKQ29_P(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)q[i] * ZKQ29_P(k - i);
    r.l[k - 9] = (u32)acc & ZKQ29_M;
    acc >>= 29;
  }
  ZKQ29_EXPECT(acc < (1ull << 32));
  r.l[8] = (u32)acc;
  return r;
}
// a a: the cross products once, against the doubled limbs (45 multiply-adds for the operand part instead of 81); a = [<= 3, *]
ZK_HD Fq29 fq29_sqr(const Fq29& a) {
  u32 q[9], d[9];
  Fq29 r;
  u64 acc = 0;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(ZKWG_FQ29_CHECK)
  { unsigned __int128 worst = 0; for (int k = 0; k < 17; ++k) { unsigned __int128 s = (unsigned __int128)9 << 58; for (int i = 0; i < 9; ++i) if (k - i >= 0 && k - i < 9) s += (unsigned __int128)a.l[i] * a.l[k - i]; if (s > worst) worst = s; } ZKQ29_EXPECT(worst < ((unsigned __int128)1 << 64) - ((unsigned __int128)1 << 36)); for (int i = 0; i < 9; ++i) ZKQ29_EXPECT(a.l[i] < (1u << 31)); }
#endif
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc += (u64)d[i] * a.l[k - i];
    if ((k & 1) == 0) acc += (u64)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (u64)q[i] * ZKQ29_P(k - i);
    q[k] = ((u32)acc * ZKQ29_N0) & ZKQ29_M;
    acc += (u64)q[k] * ZKQ29_P(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; 2 * i < k; ++i) acc += (u64)d[i] * a.l[k - i];
    if ((k & 1) == 0) acc += (u64)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)q[i] * ZKQ29_P(k - i);
    r.l[k - 9] = (u32)acc & ZKQ29_M;
    acc >>= 29;
  }
  ZKQ29_EXPECT(acc < (1ull << 32));
  r.l[8] = (u32)acc;
  return r;
}
// a b + c d, one reduction
ZK_HD Fq29 fq29_dot2(const Fq29& a, const Fq29& b, const Fq29& c, const Fq29& d) {
  u32 q[9];
  Fq29 r;
  u64 acc = 0;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(ZKWG_FQ29_CHECK)
  { unsigned __int128 worst = 0; for (int k = 0; k < 17; ++k) { unsigned __int128 s = (unsigned __int128)9 << 58; for (int i = 0; i < 9; ++i) if (k - i >= 0 && k - i < 9) s += (unsigned __int128)a.l[i] * b.l[k - i] + (unsigned __int128)c.l[i] * d.l[k - i]; if (s > worst) worst = s; } ZKQ29_EXPECT(worst < ((unsigned __int128)1 << 64) - ((unsigned __int128)1 << 36)); }
#endif
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc += (u64)a.l[i] * b.l[k - i]; acc += (u64)c.l[i] * d.l[k - i]; }
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (u64)q[i] * ZKQ29_P(k - i);
    q[k] = ((u32)acc * ZKQ29_N0) & ZKQ29_M;
    acc += (u64)q[k] * ZKQ29_P(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) { acc += (u64)a.l[i] * b.l[k - i]; acc += (u64)c.l[i] * d.l[k - i]; }
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)q[i] * ZKQ29_P(k - i);
    r.l[k - 9] = (u32)acc & ZKQ29_M;
    acc >>= 29;
  }
  ZKQ29_EXPECT(acc < (1ull << 32));
  r.l[8] = (u32)acc;
  return r;
}

// ---- sums and differences ---------------------------------------------------------------------------------------------------------------
ZK_HD Fq29 fq29_add(const Fq29& a, const Fq29& b) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) { r.l[i] = a.l[i] + b.l[i]; ZKQ29_EXPECT(r.l[i] >= a.l[i]); }
  return r;
}
ZK_HD Fq29 fq29_dbl(const Fq29& a) { return fq29_add(a, a); }
// M q written so that it dominates a [U, M - 1] value limb by limb: c_i = U 2^29 + e_i (i < 8), c_8 = e_8 with
// e = M q - sum_{i = 1..8} U 2^(29 i) in plain limbs.  Compile-time constants.
struct Fq29C { u32 l[9]; };
template <int M, int U>
constexpr Fq29C fq29_negc_make() {
  u64 m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, c = 0;
  for (int i = 0; i < 9; ++i) { c += (u64)M * ZKQ29_P(i); m[i] = i < 8 ? (c & ZKQ29_M) : c; c >>= 29; }
  u64 bw = 0;
  for (int i = 1; i < 9; ++i) {
    const u64 sub = (u64)U + bw;
    if (i < 8) { if (m[i] >= sub) { m[i] -= sub; bw = 0; } else { m[i] = m[i] + (1u << 29) - sub; bw = 1; } }
    else m[i] -= sub;          // (M q >> 232 is far above U + 1 for every M >= 1)
  }
  Fq29C r{};
  for (int i = 0; i < 9; ++i) r.l[i] = i < 8 ? (u32)(m[i] + ((u64)U << 29)) : (u32)m[8];
  return r;
}
template <int M, int U>
ZK_HD Fq29 fq29_negc() {
  constexpr Fq29C c = fq29_negc_make<M, U>();
  return Fq29{{c.l[0], c.l[1], c.l[2], c.l[3], c.l[4], c.l[5], c.l[6], c.l[7], c.l[8]}};
}
// a - b + M q for b = [U, <= M - 1]
template <int M, int U>
ZK_HD Fq29 fq29_sub(const Fq29& a, const Fq29& b) {
  const Fq29 c = fq29_negc<M, U>();
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    ZKQ29_EXPECT(c.l[i] >= b.l[i]);
    r.l[i] = a.l[i] + (c.l[i] - b.l[i]);
    ZKQ29_EXPECT(r.l[i] >= a.l[i]);
  }
  return r;
}
template <int M, int U>
ZK_HD Fq29 fq29_neg(const Fq29& b) { return fq29_sub<M, U>(fq29_zero(), b); }
