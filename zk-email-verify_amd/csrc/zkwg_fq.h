// BN254 base-field (Fq) arithmetic, 4 x 64-bit limbs, Montgomery form -- the coordinates of the G1 points the prover's
// multi-exponentiations add (oracle: oracle/pyref/bn254_g1.py).  Same representation and the same two
// Montgomery-product paths as zkwg_fr.h: 4 x 64-bit limbs through __int128 on the host, 8 x 32-bit limbs on gfx950 (no 64 x 64
// multiplier there: every inner step is one v_mad_u64_u32).  The two paths are plain functions of (a, b, modulus), compiled for
// both sides, so the CPU tests run the device's 32-bit path as well (tests/test_g1_cpu.py).
//
// Used by: the key tables and fixed-base multiples (zkwg_kernels_msm.hip zk_msm_table / zk_fixed_base), the proof assembly on the host
// (zkwg_groth16_assemble) and the CPU tests' reference; the sums themselves run the limb form of zkwg_fq29.h since round 6.
#pragma once
#include "zkwg_fr.h"

struct Fq {
  u64 l[4];
};

#define ZK_Q0 0x3c208c16d87cfd47ULL
#define ZK_Q1 0x97816a916871ca8dULL
#define ZK_Q2 0xb85045b68181585dULL
#define ZK_Q3 0x30644e72e131a029ULL
#define ZK_QN0 0x87d20782e4866389ULL  // -q^{-1} mod 2^64

ZK_HD Fq fq_p() { return Fq{{ZK_Q0, ZK_Q1, ZK_Q2, ZK_Q3}}; }
ZK_HD Fq fq_zero() { return Fq{{0, 0, 0, 0}}; }
ZK_HD Fq fq_R() { return Fq{{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}}; }    // 1 in Montgomery form
ZK_HD Fq fq_R2() { return Fq{{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}; }
ZK_HD Fq fq_3R() { return Fq{{0x7a17caa950ad28d7ULL, 0x1f6ac17ae15521b9ULL, 0x334bea4e696bd284ULL, 0x2a1f6744ce179d8eULL}}; }   // the curve's b = 3
ZK_HD bool fq_is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
ZK_HD bool fq_eq(const Fq& a, const Fq& b) { return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0; }
ZK_HD bool fq_geq(const Fq& a, const Fq& b) {
  for (int i = 3; i >= 0; --i) {
    if (a.l[i] > b.l[i]) return true;
    if (a.l[i] < b.l[i]) return false;
  }
  return true;
}
ZK_HD Fq fq_sub_raw(const Fq& a, const Fq& b, u64& borrow) {
  Fq r;
  borrow = 0;
  for (int i = 0; i < 4; ++i) r.l[i] = zk_sbb(a.l[i], b.l[i], borrow);
  return r;
}
ZK_HD Fq fq_add(const Fq& a, const Fq& b) {   // a, b < q < 2^254: no carry out of 256 bits
  Fq r;
  u64 c = 0;
  for (int i = 0; i < 4; ++i) r.l[i] = zk_adc(a.l[i], b.l[i], c);
  if (fq_geq(r, fq_p())) { u64 bw; r = fq_sub_raw(r, fq_p(), bw); }
  return r;
}
ZK_HD Fq fq_sub(const Fq& a, const Fq& b) {
  u64 bw;
  Fq r = fq_sub_raw(a, b, bw);
  if (bw) { u64 c = 0; const Fq p = fq_p(); for (int i = 0; i < 4; ++i) r.l[i] = zk_adc(r.l[i], p.l[i], c); }
  return r;
}
ZK_HD Fq fq_neg(const Fq& a) {
  if (fq_is_zero(a)) return a;
  u64 bw;
  return fq_sub_raw(fq_p(), a, bw);
}
ZK_HD Fq fq_dbl(const Fq& a) { return fq_add(a, a); }

// Montgomery product a * b * 2^-256 mod q, CIOS over 8 x 32-bit limbs (the gfx950 path: 128 v_mad_u64_u32)
ZK_HD Fq fq_mont_mul_32(const Fq& a, const Fq& b) {
  const u32 P32[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  const u32 N0_32 = 0xe4866389u;  // -q^{-1} mod 2^32
  u32 A[8], Bv[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 4; ++i) {
    A[2 * i] = (u32)a.l[i]; A[2 * i + 1] = (u32)(a.l[i] >> 32);
    Bv[2 * i] = (u32)b.l[i]; Bv[2 * i + 1] = (u32)(b.l[i] >> 32);
  }
  u32 t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 8; ++i) {
    u64 c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 8; ++j) {
      c = (u64)A[j] * Bv[i] + t[j] + c;   // <= (2^32-1)^2 + 2(2^32-1) < 2^64
      t[j] = (u32)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (u32)c;
    t[9] = (u32)(c >> 32);
    const u32 m = t[0] * N0_32;
    c = (u64)m * P32[0] + t[0];
    c >>= 32;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 1; j < 8; ++j) {
      c = (u64)m * P32[j] + t[j] + c;
      t[j - 1] = (u32)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (u32)c;
    t[8] = t[9] + (u32)(c >> 32);
  }
  Fq r{{(u64)t[0] | ((u64)t[1] << 32), (u64)t[2] | ((u64)t[3] << 32), (u64)t[4] | ((u64)t[5] << 32), (u64)t[6] | ((u64)t[7] << 32)}};
  if (t[8] || fq_geq(r, fq_p())) { u64 bw; r = fq_sub_raw(r, fq_p(), bw); }
  return r;
}
#if !defined(__HIP_DEVICE_COMPILE__)
// the same over 4 x 64-bit limbs (host)
inline Fq fq_mont_mul_64(const Fq& a, const Fq& b) {
  const u64 p[4] = {ZK_Q0, ZK_Q1, ZK_Q2, ZK_Q3};
  u64 t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (unsigned __int128)a.l[j] * b.l[i] + t[j];
      t[j] = (u64)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (u64)c;
    t[5] = (u64)(c >> 64);
    const u64 m = t[0] * ZK_QN0;
    c = (unsigned __int128)m * p[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (unsigned __int128)m * p[j] + t[j];
      t[j - 1] = (u64)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (u64)c;
    t[4] = t[5] + (u64)(c >> 64);
  }
  Fq r{{t[0], t[1], t[2], t[3]}};
  if (t[4] || fq_geq(r, fq_p())) { u64 bw; r = fq_sub_raw(r, fq_p(), bw); }
  return r;
}
#endif
// the device's product since round 5: 9 x 29-bit product scanning (zkwg_comba29.h), 1.7 x the CIOS's rate
ZK_HD Fq fq_mont_mul_comba(const Fq& a, const Fq& b) {
  const ZkComba29P P{{0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu}, 0x04866389u};
  const u64 p64[4] = {ZK_Q0, ZK_Q1, ZK_Q2, ZK_Q3};
  Fq r;
  zk_comba29_mul(a.l, b.l, P, p64, r.l);
  return r;
}
ZK_HD Fq fq_mont_mul(const Fq& a, const Fq& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKWG_FR_CIOS32)
  return fq_mont_mul_comba(a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
  return fq_mont_mul_32(a, b);
#else
  return fq_mont_mul_64(a, b);
#endif
}
ZK_HD Fq fq_mont_sqr(const Fq& a) { return fq_mont_mul(a, a); }
ZK_HD Fq fq_to_mont(const Fq& a) { return fq_mont_mul(a, fq_R2()); }
ZK_HD Fq fq_from_mont(const Fq& a) { return fq_mont_mul(a, Fq{{1, 0, 0, 0}}); }
// a^(q-2) in Montgomery form (host: conversions to affine; a kernel batches inversions instead)
ZK_HD Fq fq_mont_inv(const Fq& a) {
  const u64 e[4] = {ZK_Q0 - 2, ZK_Q1, ZK_Q2, ZK_Q3};
  Fq r = fq_R(), b = a;
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 64; ++k) {
      if ((e[i] >> k) & 1) r = fq_mont_mul(r, b);
      b = fq_mont_sqr(b);
    }
  return r;
}
