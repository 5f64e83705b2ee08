// BN254 G2 (the sextic twist y^2 = x^3 + 3 / (9 + i) over Fq2 = Fq[i] / (i^2 + 1)) for `proof.pi_b`, the one G2
// multi-exponentiation of groth16_prove.js (oracle: oracle/pyref/bn254_g2.py).  The same XYZZ formulas as zkwg_g1.h over the
// quadratic extension: an Fq2 product is 3 Fq products (Karatsuba), a square 2, so a mixed addition costs 8 x 3 + 2 x 2 = 28 Fq
// products against G1's 10.  Layout as in the zkey: x.c0 | x.c1 | y.c0 | y.c1, Montgomery form, 128 bytes, zeros = infinity.
//
// Used by: zk_msm_table / zk_fixed_base (once per key / tools: one lane per point, Fq2 products as function calls -- 400 bytes of scratch there) and
// the host (proof assembly, tests).  The G2 sum itself runs on lane pairs in limb form (zkwg_ec29.h).
#pragma once
#include "zkwg_fq.h"

struct Fq2 { Fq c0, c1; };
ZK_HD Fq2 fq2_zero() { return Fq2{fq_zero(), fq_zero()}; }
ZK_HD Fq2 fq2_one() { return Fq2{fq_R(), fq_zero()}; }
ZK_HD bool fq2_is_zero(const Fq2& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
ZK_HD bool fq2_eq(const Fq2& a, const Fq2& b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
ZK_HD Fq2 fq2_add(const Fq2& a, const Fq2& b) { return Fq2{fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; }
ZK_HD Fq2 fq2_sub(const Fq2& a, const Fq2& b) { return Fq2{fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; }
ZK_HD Fq2 fq2_neg(const Fq2& a) { return Fq2{fq_neg(a.c0), fq_neg(a.c1)}; }
ZK_HD Fq2 fq2_dbl(const Fq2& a) { return Fq2{fq_dbl(a.c0), fq_dbl(a.c1)}; }
// On the device the two products are real functions (arguments by value, in registers): G2 point formulas inline 28 - 42 field
// products each, and with every product expanded in place (3 x ~560 instructions per fq2_mul) the multi-exponentiation kernels
// took the compiler more than half an hour; a call costs a few dozen instructions against ~1,700 of work.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_FQ2_FN __device__ __noinline__
#else
#define ZK_FQ2_FN inline
#endif
// (a0 + a1 i)(b0 + b1 i) = (a0 b0 - a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) i
ZK_FQ2_FN Fq2 fq2_mul(Fq2 a, Fq2 b) {
  const Fq t0 = fq_mont_mul(a.c0, b.c0), t1 = fq_mont_mul(a.c1, b.c1);
  const Fq t2 = fq_mont_mul(fq_add(a.c0, a.c1), fq_add(b.c0, b.c1));
  return Fq2{fq_sub(t0, t1), fq_sub(fq_sub(t2, t0), t1)};
}
// (a0 + a1 i)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 i
ZK_FQ2_FN Fq2 fq2_sqr(Fq2 a) {
  const Fq t = fq_mont_mul(a.c0, a.c1);
  return Fq2{fq_mont_mul(fq_add(a.c0, a.c1), fq_sub(a.c0, a.c1)), fq_dbl(t)};
}
// 1 / (a0 + a1 i) = (a0 - a1 i) / (a0^2 + a1^2)
ZK_HD Fq2 fq2_inv(const Fq2& a) {
  const Fq d = fq_mont_inv(fq_add(fq_mont_sqr(a.c0), fq_mont_sqr(a.c1)));
  return Fq2{fq_mont_mul(a.c0, d), fq_neg(fq_mont_mul(a.c1, d))};
}
// the twist's b = 3 / (9 + i), Montgomery form
ZK_HD Fq2 fq2_twist_b() {
  return Fq2{Fq{{0x3bf938e377b802a8ULL, 0x020b1b273633535dULL, 0x26b7edf049755260ULL, 0x2514c6324384a86dULL}},
             Fq{{0x38e7ecccd1dcff67ULL, 0x65f0b37d93ce0d3eULL, 0xd749d0dd22ac00aaULL, 0x0141b9ce4a688d4dULL}}};
}

struct G2Affine { Fq2 x, y; };            // Montgomery form; all zeros = infinity
struct G2Xyzz { Fq2 x, y, zz, zzz; };     // zz = 0 = infinity

ZK_HD bool g2_is_inf(const G2Affine& p) { return fq2_is_zero(p.x) && fq2_is_zero(p.y); }
ZK_HD bool g2_is_inf(const G2Xyzz& p) { return fq2_is_zero(p.zz); }
ZK_HD G2Xyzz g2_xyzz_inf() { return G2Xyzz{fq2_zero(), fq2_zero(), fq2_zero(), fq2_zero()}; }
ZK_HD G2Affine g2_neg(const G2Affine& p) { return G2Affine{p.x, fq2_neg(p.y)}; }

ZK_HD G2Xyzz g2_dbl_affine(const G2Affine& p) {
  if (g2_is_inf(p) || fq2_is_zero(p.y)) return g2_xyzz_inf();
  const Fq2 U = fq2_dbl(p.y), V = fq2_sqr(U), W = fq2_mul(U, V), S = fq2_mul(p.x, V);
  const Fq2 X2 = fq2_sqr(p.x), M = fq2_add(fq2_dbl(X2), X2);
  G2Xyzz r;
  r.x = fq2_sub(fq2_sqr(M), fq2_dbl(S));
  r.y = fq2_sub(fq2_mul(M, fq2_sub(S, r.x)), fq2_mul(W, p.y));
  r.zz = V; r.zzz = W;
  return r;
}
ZK_HD G2Xyzz g2_dbl(const G2Xyzz& p) {
  if (g2_is_inf(p)) return p;
  const Fq2 U = fq2_dbl(p.y), V = fq2_sqr(U), W = fq2_mul(U, V), S = fq2_mul(p.x, V);
  const Fq2 X2 = fq2_sqr(p.x), M = fq2_add(fq2_dbl(X2), X2);
  G2Xyzz r;
  r.x = fq2_sub(fq2_sqr(M), fq2_dbl(S));
  r.y = fq2_sub(fq2_mul(M, fq2_sub(S, r.x)), fq2_mul(W, p.y));
  r.zz = fq2_mul(V, p.zz); r.zzz = fq2_mul(W, p.zzz);
  return r;
}
ZK_HD G2Xyzz g2_add_mixed(const G2Xyzz& a, const G2Affine& p) {
  if (g2_is_inf(p)) return a;
  if (g2_is_inf(a)) return G2Xyzz{p.x, p.y, fq2_one(), fq2_one()};
  const Fq2 U2 = fq2_mul(p.x, a.zz), S2 = fq2_mul(p.y, a.zzz);
  const Fq2 P = fq2_sub(U2, a.x), Rr = fq2_sub(S2, a.y);
  if (fq2_is_zero(P)) return fq2_is_zero(Rr) ? g2_dbl_affine(p) : g2_xyzz_inf();
  const Fq2 PP = fq2_sqr(P), PPP = fq2_mul(P, PP), Qv = fq2_mul(a.x, PP);
  G2Xyzz r;
  r.x = fq2_sub(fq2_sub(fq2_sqr(Rr), PPP), fq2_dbl(Qv));
  r.y = fq2_sub(fq2_mul(Rr, fq2_sub(Qv, r.x)), fq2_mul(a.y, PPP));
  r.zz = fq2_mul(a.zz, PP);
  r.zzz = fq2_mul(a.zzz, PPP);
  return r;
}
ZK_HD G2Xyzz g2_add(const G2Xyzz& a, const G2Xyzz& b) {
  if (g2_is_inf(a)) return b;
  if (g2_is_inf(b)) return a;
  const Fq2 U1 = fq2_mul(a.x, b.zz), U2 = fq2_mul(b.x, a.zz), S1 = fq2_mul(a.y, b.zzz), S2 = fq2_mul(b.y, a.zzz);
  const Fq2 P = fq2_sub(U2, U1), Rr = fq2_sub(S2, S1);
  if (fq2_is_zero(P)) return fq2_is_zero(Rr) ? g2_dbl(a) : g2_xyzz_inf();
  const Fq2 PP = fq2_sqr(P), PPP = fq2_mul(P, PP), Qv = fq2_mul(U1, PP);
  G2Xyzz r;
  r.x = fq2_sub(fq2_sub(fq2_sqr(Rr), PPP), fq2_dbl(Qv));
  r.y = fq2_sub(fq2_mul(Rr, fq2_sub(Qv, r.x)), fq2_mul(S1, PPP));
  r.zz = fq2_mul(fq2_mul(a.zz, b.zz), PP);
  r.zzz = fq2_mul(fq2_mul(a.zzz, b.zzz), PPP);
  return r;
}
ZK_HD G2Affine g2_to_affine(const G2Xyzz& p) {
  if (g2_is_inf(p)) return G2Affine{fq2_zero(), fq2_zero()};
  const Fq2 iz3 = fq2_inv(p.zzz);
  const Fq2 iz2 = fq2_sqr(fq2_mul(iz3, p.zz));
  return G2Affine{fq2_mul(p.x, iz2), fq2_mul(p.y, iz3)};
}
ZK_HD bool g2_on_curve(const G2Affine& p) {
  if (g2_is_inf(p)) return true;
  return fq2_eq(fq2_sqr(p.y), fq2_add(fq2_mul(fq2_sqr(p.x), p.x), fq2_twist_b()));
}
