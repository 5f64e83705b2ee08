// BN254 Fr Montgomery products for the arithmetic-bound kernels (zk_ntt_*, zk_rslb_*, the MSM scalars): 9 limbs of 29 bits, R = 2^261.
//
// tools/mulbench.hip (profiles/r05/r05_a_mulbench.txt) measured what gfx950 issues: v_mad_u64_u32 runs at (nearly) the plain VALU
// rate, NOT at a quarter of it as rounds 2-4 assumed -- the 8 x 32-bit CIOS of zkwg_fr.h is slow because of what surrounds its 128
// multiply-adds: a 32 x 32 product plus two 32-bit addends needs a 64-bit add with zero-extended operands at every step (120
// v_lshl_add_u64 + 286 v_mov_b32 per product as compiled).  With 29-bit limbs a column of the schoolbook product -- up to 9 products
// of the operands and 9 of the reduction, each below 2^58..2^60 -- fits a 64-bit accumulator, so the whole product is 162 v_mad_u64_u32
// chained through their 64-bit addend, one AND + one 64-bit shift per column, and no carry handling at all (Comba / product scanning).
//
// Values are kept in limb form between products.  A product accepts one operand `a` with limbs < 3 * 2^30 (sums and differences of
// products need no carry propagation first) and one operand `b` with limbs < 2^29 (a twiddle / table constant), and returns limbs
// < 2^29 (the top one < 2^32) with value < a b / 2^261 + r.
#pragma once
#include "zkwg_fr.h"

struct Fr29 {
  u32 l[9];
};
#define ZK29_M 0x1fffffffu
#define ZK29_N0 0x0fffffffu      // -r^-1 mod 2^29

ZK_HD Fr29 fr29_from_fr(const Fr& x) {
  Fr29 r;
  const u64 w[5] = {x.l[0], x.l[1], x.l[2], x.l[3], 0};
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, k = bit >> 6, s = bit & 63;
    const u64 v = s ? ((w[k] >> s) | (w[k + 1] << (64 - s))) : w[k];
    r.l[i] = i < 8 ? ((u32)v & ZK29_M) : (u32)v;
  }
  return r;
}
// limb form -> canonical Fr (value < 2^256 required; fully reduced)
ZK_HD Fr fr29_to_fr(const Fr29& x) {
  // carry-propagate, then pack
  u64 t[9];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { c += x.l[i]; t[i] = i < 8 ? (c & ZK29_M) : c; c >>= 29; }
  u64 w[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, k = bit >> 6, s = bit & 63;
    w[k] |= t[i] << s;
    if (s > 64 - 32 && k + 1 < 5) w[k + 1] |= s ? (t[i] >> (64 - s)) : 0;
  }
  Fr r{{w[0], w[1], w[2], w[3]}};
  for (int k = 0; k < 8 && fr_geq(r, fr_p()); ++k) {
    u64 bw;
    r = fr_sub_raw(r, fr_p(), bw);
  }
  return r;
}
ZK_HD Fr29 fr29_mul(const Fr29& a, const Fr29& b) {
  const u32 P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  u32 q[9];
  Fr29 r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (u64)q[i] * P[k - i];
    q[k] = ((u32)acc * ZK29_N0) & ZK29_M;
    acc += (u64)q[k] * P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)q[i] * P[k - i];
    r.l[k - 9] = (u32)acc & ZK29_M;
    acc >>= 29;
  }
  r.l[8] = (u32)acc;
  return r;
}
// lazy sums: limbs add without carries; a difference adds a multiple of r whose limbs dominate a normalised subtrahend's
ZK_HD Fr29 fr29_add(const Fr29& a, const Fr29& b) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}

// ---- what the transforms of zkwg_kernels_ntt.hip need on top (round 6): values stay in limb form across the butterfly stages of a pass.
// Notation of zkwg_fq29.h: [U, V] = limbs 0 .. 7 < U 2^29, value < V r; the top limb l[8] is only bounded by the value (V <= 1,300:
// l[8] < 2^32), a product with a canonical twiddle brings any such value back to < (V / 169 + 1) r.
#define ZKR29_P(i) ((i) == 0 ? 0x10000001u : (i) == 1 ? 0x1f0fac9fu : (i) == 2 ? 0x0e5c2450u : (i) == 3 ? 0x07d090f3u : (i) == 4 ? 0x1585d283u : (i) == 5 ? 0x02db40c0u : (i) == 6 ? 0x00a6e141u : (i) == 7 ? 0x0e5c2634u : 0x0030644eu)
ZK_HD Fr29 fr29_zero() { return Fr29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
// exact carry propagation: limbs 0 .. 7 < 2^29, the value unchanged
ZK_HD Fr29 fr29_norm(const Fr29& a) {
  Fr29 r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const u32 t = a.l[i] + c; r.l[i] = t & ZK29_M; c = t >> 29; }
  r.l[8] = a.l[8] + c;
  return r;
}
// M r written so that it dominates a [U, M - 1] value limb by limb (zkwg_fq29.h fq29_negc_make, with r's limbs)
struct Fr29C { u32 l[9]; };
template <int M, int U>
constexpr Fr29C fr29_negc_make() {
  u64 m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, c = 0;
  for (int i = 0; i < 9; ++i) { c += (u64)M * ZKR29_P(i); m[i] = i < 8 ? (c & ZK29_M) : c; c >>= 29; }
  u64 bw = 0;
  for (int i = 1; i < 9; ++i) {
    const u64 sub = (u64)U + bw;
    if (i < 8) { if (m[i] >= sub) { m[i] -= sub; bw = 0; } else { m[i] = m[i] + (1u << 29) - sub; bw = 1; } }
    else m[i] -= sub;
  }
  Fr29C r{};
  for (int i = 0; i < 9; ++i) r.l[i] = i < 8 ? (u32)(m[i] + ((u64)U << 29)) : (u32)m[8];
  return r;
}
// a - b + M r for b = [U, <= M - 1]: [Ua + U + 1, Va + M]
template <int M, int U>
ZK_HD Fr29 fr29_sub(const Fr29& a, const Fr29& b) {
  constexpr Fr29C c = fr29_negc_make<M, U>();
  Fr29 r;
  r.l[0] = a.l[0] + (c.l[0] - b.l[0]); r.l[1] = a.l[1] + (c.l[1] - b.l[1]); r.l[2] = a.l[2] + (c.l[2] - b.l[2]);
  r.l[3] = a.l[3] + (c.l[3] - b.l[3]); r.l[4] = a.l[4] + (c.l[4] - b.l[4]); r.l[5] = a.l[5] + (c.l[5] - b.l[5]);
  r.l[6] = a.l[6] + (c.l[6] - b.l[6]); r.l[7] = a.l[7] + (c.l[7] - b.l[7]); r.l[8] = a.l[8] + (c.l[8] - b.l[8]);
  return r;
}
// any limb form with value < 2^256 packed into 4 x 64-bit words (NOT reduced: the work buffers of the transforms hold such words)
ZK_HD Fr fr29_pack(const Fr29& x) {
  const Fr29 t = fr29_norm(x);
  u64 w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, k = bit >> 6, s = bit & 63;
    w[k] |= (u64)t.l[i] << s;
    if (s > 64 - 29 && k + 1 < 4) w[k + 1] |= (u64)t.l[i] >> (64 - s);
  }
  return Fr{{w[0], w[1], w[2], w[3]}};
}
// canonical words of a value < V r (V <= 63)
template <int V>
ZK_HD Fr fr29_to_fr_v(const Fr29& a) {
  Fr29 n = fr29_norm(a);
#pragma unroll
  for (int s = 5; s >= 0; --s) {
    if ((1 << s) >= V) continue;
    // k r, k = 2^s, in plain limbs
    Fr29 kp;
    { u64 c = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i) { c += (u64)(1u << s) * ZKR29_P(i); kp.l[i] = i < 8 ? ((u32)c & ZK29_M) : (u32)c; c >>= 29; } }
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
        bw = i < 8 ? (t >> 31) : 0;
        n.l[i] = i < 8 ? (t & ZK29_M) : t;
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
  return Fr{{w[0], w[1], w[2], w[3]}};
}
