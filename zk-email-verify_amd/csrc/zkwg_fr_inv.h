// a^-1 mod r (BN254 scalar field) by Bernstein-Yang "safegcd" division steps, 30-bit signed limbs.
//
// Why this and not Fermat or the binary extended Euclid: the 306 IsEqual inverses of one email
// (lib/bigint.circom:16-60, 18 BigLessThan x 17 limbs) are batch-inverted per LANE, so the wavefront
// pays for the slowest lane of every data-dependent loop.  The division-step iteration has no
// data-dependent control flow at all (fixed 20 x 30 steps, everything is masks and selects), works on
// 32-bit words (gfx950 has no 64-bit multiplier; v_mad_i64_i32 carries the matrix updates) and costs
// about 18 k issue slots against ~200 k for a branch-free binary Euclid and ~50 k multiplier issues
// for a^(r-2).  Algorithm: D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation and
// modular inversion" (2019), in the half-delta form with 590 <= 600 steps for 256-bit inputs.
#pragma once
#include "zkwg_fr.h"

struct ZkS30 { int32_t v[9]; };   // value = sum v[i] 2^(30 i), limbs in (-2^30, 2^30)

// 30 division steps on the low words; t = [u v; q r] with [f'; g'] = t [f; g] / 2^30
ZK_HD int32_t zk_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t* t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 5
  for (int i = 0; i < 30; ++i) {
    uint32_t c1 = (uint32_t)(zeta >> 31);          // zeta < 0
    const uint32_t c2 = (uint32_t)0 - (g & 1u);    // g odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
    g += x & c2; q += y & c2; r += z & c2;
    c1 &= c2;
    zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
    f += g & c1; u += q & c1; v += r & c1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
  return zeta;
}

ZK_HD void zk_update_fg_30(ZkS30& f, ZkS30& g, const int32_t* t) {
  const int32_t M30 = (int32_t)(0xffffffffu >> 2);
  const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
  int64_t cf = u * f.v[0] + v * g.v[0], cg = q * f.v[0] + r * g.v[0];
  cf >>= 30; cg >>= 30;                            // the low 30 bits are zero by construction
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    cf += u * f.v[i] + v * g.v[i];
    cg += q * f.v[i] + r * g.v[i];
    f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
    g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
  }
  f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
}

// d, e in (-2r, r):  [d'; e'] = (t [d; e] + r [md; me]) / 2^30 with md, me chosen so the division is exact
ZK_HD void zk_update_de_30(ZkS30& d, ZkS30& e, const int32_t* t) {
  const int32_t M30 = (int32_t)(0xffffffffu >> 2);
  const int32_t MOD[9] = {805306369, 260560463, 462883092, 217715230, 22568232, 18274822, 436378501, 329037900, 12388};
  const uint32_t MOD_INV30 = 268435457u;           // r^-1 mod 2^30
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
  int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
  int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
  md -= (int32_t)((MOD_INV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((MOD_INV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)MOD[0] * md;
  ce += (int64_t)MOD[0] * me;
  cd >>= 30; ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i] + (int64_t)MOD[i] * md;
    ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i] + (int64_t)MOD[i] * me;
    d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
    e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
  }
  d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
}

// a in [0, r), standard form -> a^-1 in [0, r); 0 -> 0
ZK_HD Fr fr_inv_by(const Fr& a) {
  const int32_t M30 = (int32_t)(0xffffffffu >> 2);
  const int32_t MOD[9] = {805306369, 260560463, 462883092, 217715230, 22568232, 18274822, 436378501, 329037900, 12388};
  ZkS30 f, g, d, e;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    f.v[i] = MOD[i];
    d.v[i] = 0;
    e.v[i] = i == 0 ? 1 : 0;
    // bits [30 i, 30 i + 30) of a
    const int lo = 30 * i, w = lo >> 6, s = lo & 63;
    u64 x = a.l[w] >> s;
    if (s > 34 && w < 3) x |= a.l[w + 1] << (64 - s);
    g.v[i] = (int32_t)(x & (u64)M30);
  }
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 20; ++it) {
    int32_t t[4];
    zeta = zk_divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    zk_update_de_30(d, e, t);
    zk_update_fg_30(f, g, t);
  }
  // f = +-1 (or 0 for a = 0), g = 0;  d = +- a^-1 in (-2r, r)
  // normalise: add r if negative, negate if f is negative, add r if negative again
  const int32_t fneg = f.v[8] >> 31;
  int32_t cadd = d.v[8] >> 31;
  int32_t r9[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int32_t x = d.v[i] + (MOD[i] & cadd);
    x = (x ^ fneg) - fneg;
    r9[i] = x;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { r9[i + 1] += r9[i] >> 30; r9[i] &= M30; }
  cadd = r9[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) r9[i] += MOD[i] & cadd;
#pragma unroll
  for (int i = 0; i < 8; ++i) { r9[i + 1] += r9[i] >> 30; r9[i] &= M30; }
  // the value may still be >= r by one modulus (d in (-2r, r) before the fix-ups): one conditional subtract
  Fr out = fr_zero();
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int lo = 30 * i, w = lo >> 6, s = lo & 63;
    const u64 x = (u64)(uint32_t)r9[i];
    out.l[w] |= x << s;
    if (s > 34 && w < 3) out.l[w + 1] |= x >> (64 - s);
  }
  if (fr_geq(out, fr_p())) { u64 bw; out = fr_sub_raw(out, fr_p(), bw); }
  return out;
}
