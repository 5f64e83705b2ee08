// BN254 G1 (y^2 = x^3 + 3 over Fq) for the prover's multi-exponentiations (the row after the H
// evaluations of zkwg_ntt_api.hip; oracle: oracle/pyref/bn254_g1.py, which restates snarkjs's groth16_prove.js call sites).
//
// Layouts are the ones the zkey and ffjavascript use: a base is an AFFINE point, x | y in Montgomery form, 64 bytes, the point at
// infinity all zeros ((0, 0) is not on the curve).  Accumulators are in XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2;
// infinity: ZZ = 0): a mixed addition accumulator += affine costs 8 M + 2 S = 10 Montgomery products = 1,280 v_mad_u64_u32 on
// gfx950 (zkwg_fq.h), against 11 M + 5 S for Jacobian -- and the bucket method is nothing but mixed additions.
//
// The multi-exponentiation is the bucket method with SIGNED windows of c bits: scalar = sum_w d_w 2^(c w), |d_w| <= 2^(c-1), so a
// window has 2^(c-1) buckets; bucket b of window w collects +P for d_w = b + 1 and -P for d_w = -(b + 1); then
// sum_b (b + 1) bucket[b] by a running sum from the top, and the windows are combined by c doublings each.  zk_msm_host() below
// runs exactly these steps with the functions a kernel will call (digit extraction, accumulate, running sum), so the CPU tests pin
// the arithmetic and the edge cases (equal points, opposite points, infinity, zero and maximal digits) before any kernel exists.
//
// Used by: zk_msm_table (K shifted copies of the bases, once per key), zk_fixed_base, zkwg_groth16_assemble / zkwg_msm_finish_host (host), the CPU
// tests' reference.  The kernels of the sums run the same formulas over the lazy limb form (zkwg_ec29.h).
#pragma once
#include "zkwg_fq.h"
#include <vector>

struct G1Affine { Fq x, y; };              // Montgomery form; (0, 0) = infinity
struct G1Xyzz { Fq x, y, zz, zzz; };       // Montgomery form; zz = 0 = infinity

ZK_HD bool g1_is_inf(const G1Affine& p) { return fq_is_zero(p.x) && fq_is_zero(p.y); }
ZK_HD bool g1_is_inf(const G1Xyzz& p) { return fq_is_zero(p.zz); }
ZK_HD G1Xyzz g1_xyzz_inf() { return G1Xyzz{fq_zero(), fq_zero(), fq_zero(), fq_zero()}; }
ZK_HD G1Xyzz g1_from_affine(const G1Affine& p) { return g1_is_inf(p) ? g1_xyzz_inf() : G1Xyzz{p.x, p.y, fq_R(), fq_R()}; }
ZK_HD G1Affine g1_neg(const G1Affine& p) { return G1Affine{p.x, fq_neg(p.y)}; }

// 2 P for an affine P (EFD mdbl-2008-s-1, a = 0): 3 M + 3 S... here 2 S + 4 M with U = 2 Y
ZK_HD G1Xyzz g1_dbl_affine(const G1Affine& p) {
  if (g1_is_inf(p) || fq_is_zero(p.y)) return g1_xyzz_inf();   // (no point of order 2 on this curve; kept for completeness)
  const Fq U = fq_dbl(p.y), V = fq_mont_sqr(U), W = fq_mont_mul(U, V), S = fq_mont_mul(p.x, V);
  const Fq X2 = fq_mont_sqr(p.x), M = fq_add(fq_dbl(X2), X2);
  G1Xyzz r;
  r.x = fq_sub(fq_mont_sqr(M), fq_dbl(S));
  r.y = fq_sub(fq_mont_mul(M, fq_sub(S, r.x)), fq_mont_mul(W, p.y));
  r.zz = V; r.zzz = W;
  return r;
}
// 2 P (EFD dbl-2008-s-1, a = 0)
ZK_HD G1Xyzz g1_dbl(const G1Xyzz& p) {
  if (g1_is_inf(p)) return p;
  const Fq U = fq_dbl(p.y), V = fq_mont_sqr(U), W = fq_mont_mul(U, V), S = fq_mont_mul(p.x, V);
  const Fq X2 = fq_mont_sqr(p.x), M = fq_add(fq_dbl(X2), X2);
  G1Xyzz r;
  r.x = fq_sub(fq_mont_sqr(M), fq_dbl(S));
  r.y = fq_sub(fq_mont_mul(M, fq_sub(S, r.x)), fq_mont_mul(W, p.y));
  r.zz = fq_mont_mul(V, p.zz); r.zzz = fq_mont_mul(W, p.zzz);
  return r;
}
// acc + P for an affine P (EFD madd-2008-s): 8 M + 2 S; every special case handled (the bucket method meets all of them: a bucket's
// first point, the same base twice, a base and its negative)
ZK_HD G1Xyzz g1_add_mixed(const G1Xyzz& a, const G1Affine& p) {
  if (g1_is_inf(p)) return a;
  if (g1_is_inf(a)) return G1Xyzz{p.x, p.y, fq_R(), fq_R()};
  const Fq U2 = fq_mont_mul(p.x, a.zz), S2 = fq_mont_mul(p.y, a.zzz);
  const Fq P = fq_sub(U2, a.x), Rr = fq_sub(S2, a.y);
  if (fq_is_zero(P)) return fq_is_zero(Rr) ? g1_dbl_affine(p) : g1_xyzz_inf();
  const Fq PP = fq_mont_sqr(P), PPP = fq_mont_mul(P, PP), Qv = fq_mont_mul(a.x, PP);
  G1Xyzz r;
  r.x = fq_sub(fq_sub(fq_mont_sqr(Rr), PPP), fq_dbl(Qv));
  r.y = fq_sub(fq_mont_mul(Rr, fq_sub(Qv, r.x)), fq_mont_mul(a.y, PPP));
  r.zz = fq_mont_mul(a.zz, PP);
  r.zzz = fq_mont_mul(a.zzz, PPP);
  return r;
}
// a + b (EFD add-2008-s): 12 M + 2 S
ZK_HD G1Xyzz g1_add(const G1Xyzz& a, const G1Xyzz& b) {
  if (g1_is_inf(a)) return b;
  if (g1_is_inf(b)) return a;
  const Fq U1 = fq_mont_mul(a.x, b.zz), U2 = fq_mont_mul(b.x, a.zz), S1 = fq_mont_mul(a.y, b.zzz), S2 = fq_mont_mul(b.y, a.zzz);
  const Fq P = fq_sub(U2, U1), Rr = fq_sub(S2, S1);
  if (fq_is_zero(P)) return fq_is_zero(Rr) ? g1_dbl(a) : g1_xyzz_inf();
  const Fq PP = fq_mont_sqr(P), PPP = fq_mont_mul(P, PP), Qv = fq_mont_mul(U1, PP);
  G1Xyzz r;
  r.x = fq_sub(fq_sub(fq_mont_sqr(Rr), PPP), fq_dbl(Qv));
  r.y = fq_sub(fq_mont_mul(Rr, fq_sub(Qv, r.x)), fq_mont_mul(S1, PPP));
  r.zz = fq_mont_mul(fq_mont_mul(a.zz, b.zz), PP);
  r.zzz = fq_mont_mul(fq_mont_mul(a.zzz, b.zzz), PPP);
  return r;
}
// -> affine (one inversion: x = X / ZZ, y = Y / ZZZ with 1 / ZZZ and ZZ / ZZZ^2 ... computed from one inverse of ZZZ)
ZK_HD G1Affine g1_to_affine(const G1Xyzz& p) {
  if (g1_is_inf(p)) return G1Affine{fq_zero(), fq_zero()};
  const Fq iz3 = fq_mont_inv(p.zzz);                               // 1 / ZZZ
  const Fq iz2 = fq_mont_sqr(fq_mont_mul(iz3, p.zz));              // (ZZ / ZZZ)^2 = 1 / ZZ   (ZZ^3 = ZZZ^2)
  return G1Affine{fq_mont_mul(p.x, iz2), fq_mont_mul(p.y, iz3)};
}
// y^2 = x^3 + 3 (Montgomery form)
ZK_HD bool g1_on_curve(const G1Affine& p) {
  if (g1_is_inf(p)) return true;
  return fq_eq(fq_mont_sqr(p.y), fq_add(fq_mont_mul(fq_mont_sqr(p.x), p.x), fq_3R()));
}

// ---- signed windows ----
// number of c-bit signed windows that cover a scalar < 2^254 (the top window absorbs the last carry)
ZK_HD u32 zk_msm_windows(u32 c) { return (254u + c) / c; }
// digit w of the scalar k (standard form, 4 x 64-bit limbs): d in [-2^(c-1), 2^(c-1)]; the caller walks w upwards and passes the
// carry along (0 at w = 0).  c <= 31.
ZK_HD int zk_msm_digit(const u64 k[4], u32 w, u32 c, u32& carry) {
  const u32 bit = w * c;
  u64 v = 0;
  if (bit < 256u) {
    const u32 limb = bit >> 6, off = bit & 63u;
    v = k[limb] >> off;
    if (off + c > 64u && limb + 1u < 4u) v |= k[limb + 1] << (64u - off);
  }
  u32 d = ((u32)v & ((1u << c) - 1u)) + carry;
  carry = d > (1u << (c - 1)) ? 1u : 0u;
  return carry ? (int)d - (int)(1u << c) : (int)d;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// sum_i k_i P_i by the bucket method, the way the kernels will run it: per window, digits -> buckets (mixed additions of +-P),
// the weighted bucket sum by a running sum from the top, c doublings between windows.  Scalars in STANDARD form (32 bytes each).
inline G1Xyzz zk_msm_host(const G1Affine* pts, const u64* scalars, size_t n, u32 c) {
  const u32 K = zk_msm_windows(c), nb = 1u << (c - 1);
  std::vector<int> digits((size_t)n * K);
  for (size_t i = 0; i < n; ++i) {
    u32 carry = 0;
    for (u32 w = 0; w < K; ++w) digits[i * K + w] = zk_msm_digit(scalars + 4 * i, w, c, carry);
    // (the top window absorbs the final carry: a scalar below 2^254 never leaves one)
  }
  G1Xyzz total = g1_xyzz_inf();
  std::vector<G1Xyzz> bucket(nb);
  for (u32 w = K; w-- > 0;) {
    for (u32 s = 0; s < c; ++s) total = g1_dbl(total);
    for (auto& b : bucket) b = g1_xyzz_inf();
    for (size_t i = 0; i < n; ++i) {
      const int d = digits[i * K + w];
      if (d > 0) bucket[d - 1] = g1_add_mixed(bucket[d - 1], pts[i]);
      else if (d < 0) bucket[-d - 1] = g1_add_mixed(bucket[-d - 1], g1_neg(pts[i]));
    }
    G1Xyzz run = g1_xyzz_inf(), acc = g1_xyzz_inf();
    for (u32 b = nb; b-- > 0;) { run = g1_add(run, bucket[b]); acc = g1_add(acc, run); }
    total = g1_add(total, acc);
  }
  return total;
}
#endif
