// BN254 scalar-field (Fr) arithmetic, 4 x 64-bit limbs, Montgomery form.
// Shared by the host-side schedule builder (inverse tables, Poseidon constants)
// and the gfx950 kernels.  r = CIRCOM_FIELD_MODULUS
// (reference: packages/helpers/src/constants.ts:1).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

struct Fr {
  u64 l[4];
};

#define ZK_P0 0x43e1f593f0000001ULL
#define ZK_P1 0x2833e84879b97091ULL
#define ZK_P2 0xb85045b68181585dULL
#define ZK_P3 0x30644e72e131a029ULL
#define ZK_N0 0xc2e1f593efffffffULL  // -r^{-1} mod 2^64

ZK_HD void zk_mul64(u64 a, u64 b, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  lo = a * b;
  hi = __umul64hi(a, b);
#else
  unsigned __int128 p = (unsigned __int128)a * b;
  lo = (u64)p;
  hi = (u64)(p >> 64);
#endif
}

// (carry, out) = a + b + carry
ZK_HD u64 zk_adc(u64 a, u64 b, u64& carry) {
  u64 s = a + b;
  u64 c1 = s < a;
  u64 s2 = s + carry;
  u64 c2 = s2 < s;
  carry = c1 + c2;
  return s2;
}
// (borrow, out) = a - b - borrow
ZK_HD u64 zk_sbb(u64 a, u64 b, u64& borrow) {
  u64 d = a - b;
  u64 b1 = a < b;
  u64 d2 = d - borrow;
  u64 b2 = d < borrow;
  borrow = b1 + b2;
  return d2;
}

ZK_HD Fr fr_p() { return Fr{{ZK_P0, ZK_P1, ZK_P2, ZK_P3}}; }
ZK_HD Fr fr_zero() { return Fr{{0, 0, 0, 0}}; }
ZK_HD Fr fr_from_u64(u64 x) { return Fr{{x, 0, 0, 0}}; }  // standard form
// Montgomery constants
ZK_HD Fr fr_R() { return Fr{{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}}; }
ZK_HD Fr fr_R2() { return Fr{{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}; }

ZK_HD bool fr_is_zero(const Fr& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
ZK_HD bool fr_eq(const Fr& a, const Fr& b) {
  return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3];
}
// a >= b as 256-bit integers
ZK_HD bool fr_geq(const Fr& a, const Fr& b) {
  for (int i = 3; i >= 0; --i) {
    if (a.l[i] > b.l[i]) return true;
    if (a.l[i] < b.l[i]) return false;
  }
  return true;
}
ZK_HD Fr fr_sub_raw(const Fr& a, const Fr& b, u64& borrow) {
  Fr r;
  borrow = 0;
  for (int i = 0; i < 4; ++i) r.l[i] = zk_sbb(a.l[i], b.l[i], borrow);
  return r;
}
ZK_HD Fr fr_add_raw(const Fr& a, const Fr& b, u64& carry) {
  Fr r;
  carry = 0;
  for (int i = 0; i < 4; ++i) r.l[i] = zk_adc(a.l[i], b.l[i], carry);
  return r;
}
// modular add/sub/neg (inputs in [0, r))
ZK_HD Fr fr_add(const Fr& a, const Fr& b) {
  u64 c;
  Fr s = fr_add_raw(a, b, c);  // < 2r < 2^255, no carry out
  if (fr_geq(s, fr_p())) {
    u64 bw;
    s = fr_sub_raw(s, fr_p(), bw);
  }
  return s;
}
ZK_HD Fr fr_sub(const Fr& a, const Fr& b) {
  u64 bw;
  Fr d = fr_sub_raw(a, b, bw);
  if (bw) {
    u64 c;
    d = fr_add_raw(d, fr_p(), c);
  }
  return d;
}
ZK_HD Fr fr_neg(const Fr& a) {
  if (fr_is_zero(a)) return a;
  u64 bw;
  return fr_sub_raw(fr_p(), a, bw);
}

#include "zkwg_comba29.h"
// the device's product since round 5 (ZKWG_FR_CIOS32 brings the round 2-4 CIOS back for comparison)
ZK_HD Fr fr_mont_mul_comba(const Fr& a, const Fr& b) {
  const ZkComba29P P{{0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu}, 0x0fffffffu};
  const u64 p64[4] = {ZK_P0, ZK_P1, ZK_P2, ZK_P3};
  Fr r;
  zk_comba29_mul(a.l, b.l, P, p64, r.l);
  return r;
}
// Montgomery product a*b*R^{-1} mod r (CIOS).  Host: 4 x 64-bit limbs via __int128.  gfx950 has no
// 64x64 multiplier: the device path runs the same algorithm on 8 x 32-bit limbs so that every
// inner step is one v_mad_u64_u32 (32x32+64) plus one 64-bit add.
// rounds 2-4: CIOS over 8 x 32-bit limbs (128 v_mad_u64_u32 + the carry handling: 562 instructions as compiled); kept for comparison
// (tools/mulbench.hip) and behind ZKWG_FR_CIOS32
ZK_HD Fr fr_mont_mul_cios32(const Fr& a, const Fr& b) {
  const u32 P32[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  const u32 N0_32 = 0xefffffffu;  // -r^{-1} mod 2^32
  u32 A[8], Bv[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A[2 * i] = (u32)a.l[i]; A[2 * i + 1] = (u32)(a.l[i] >> 32);
    Bv[2 * i] = (u32)b.l[i]; Bv[2 * i + 1] = (u32)(b.l[i] >> 32);
  }
  u32 t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u64 c = 0;
#pragma unroll
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
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      c = (u64)m * P32[j] + t[j] + c;
      t[j - 1] = (u32)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (u32)c;
    t[8] = t[9] + (u32)(c >> 32);
  }
  Fr r{{(u64)t[0] | ((u64)t[1] << 32), (u64)t[2] | ((u64)t[3] << 32), (u64)t[4] | ((u64)t[5] << 32),
        (u64)t[6] | ((u64)t[7] << 32)}};
  if (t[8] || fr_geq(r, fr_p())) {
    u64 bw;
    r = fr_sub_raw(r, fr_p(), bw);
  }
  return r;
}
ZK_HD Fr fr_mont_mul(const Fr& a, const Fr& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKWG_FR_CIOS32)
  return fr_mont_mul_comba(a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
  return fr_mont_mul_cios32(a, b);
#else
  const u64 p[4] = {ZK_P0, ZK_P1, ZK_P2, ZK_P3};
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
    const u64 m = t[0] * ZK_N0;
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
  Fr r{{t[0], t[1], t[2], t[3]}};
  if (t[4] || fr_geq(r, fr_p())) {
    u64 bw;
    r = fr_sub_raw(r, fr_p(), bw);
  }
  return r;
#endif
}
// ---- lazy reduction: sum_j a_j * b_j accumulated unreduced, one Montgomery reduction at the end ----
// Up to 17 products of values < r fit 17 x 32-bit limbs (17 r^2 < 2^513).  A dot product of n terms then
// costs n x 64 + 64 multiplier issues instead of n x 128 (dense Poseidon rounds, zkwg_poseidon_sparse.h).
struct FrWide {
  u32 l[17];
};
ZK_HD void fr_wide_zero(FrWide& w) {
#pragma unroll
  for (int i = 0; i < 17; ++i) w.l[i] = 0;
}
// w += a * b  (a, b < 2^256; the caller keeps the running sum below 2^544)
ZK_HD void fr_wide_mac(FrWide& w, const Fr& a, const Fr& b) {
  u32 A[8], Bv[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A[2 * i] = (u32)a.l[i]; A[2 * i + 1] = (u32)(a.l[i] >> 32);
    Bv[2 * i] = (u32)b.l[i]; Bv[2 * i + 1] = (u32)(b.l[i] >> 32);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u64 c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      c = (u64)A[k] * Bv[i] + w.l[i + k] + c;
      w.l[i + k] = (u32)c;
      c >>= 32;
    }
#pragma unroll
    for (int k = i + 8; k < 17; ++k) {
      c += w.l[k];
      w.l[k] = (u32)c;
      c >>= 32;
    }
  }
}
// w * 2^-256 mod r for w < 17 r^2 (result < 4.3 r before the final subtractions)
ZK_HD Fr fr_wide_redc(FrWide& w) {
  const u32 P32[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  const u32 N0_32 = 0xefffffffu;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32 m = w.l[i] * N0_32;
    u64 c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      c = (u64)m * P32[k] + w.l[i + k] + c;
      w.l[i + k] = (u32)c;
      c >>= 32;
    }
#pragma unroll
    for (int k = i + 8; k < 17; ++k) {
      c += w.l[k];
      w.l[k] = (u32)c;
      c >>= 32;
    }
  }
  // value = l[8..16] (9 limbs), < 4.3 r: subtract 4r, 2r, r where possible
  u32 v[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] = w.l[8 + i];
#pragma unroll
  for (int sh = 2; sh >= 0; --sh) {
    u32 d[9];
    u64 bw = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      // limb i of (r << sh)
      const u32 lo = i < 8 ? P32[i] : 0u, below = i > 0 ? P32[i - 1] : 0u;
      const u32 pl = sh ? (u32)((lo << sh) | (below >> (32 - sh))) : lo;
      const u64 t = (u64)v[i] - pl - bw;
      d[i] = (u32)t;
      bw = (t >> 32) & 1u;
    }
    if (!bw) {
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] = d[i];
    }
  }
  return Fr{{(u64)v[0] | ((u64)v[1] << 32), (u64)v[2] | ((u64)v[3] << 32), (u64)v[4] | ((u64)v[5] << 32),
             (u64)v[6] | ((u64)v[7] << 32)}};
}

ZK_HD Fr fr_to_mont(const Fr& a) { return fr_mont_mul(a, fr_R2()); }
ZK_HD Fr fr_from_mont(const Fr& a) { return fr_mont_mul(a, fr_from_u64(1)); }

// a^(r-2) in Montgomery form (a in Montgomery form); 0 -> 0
ZK_HD Fr fr_mont_inv(const Fr& a) {
  // exponent r - 2
  const u64 e[4] = {ZK_P0 - 2, ZK_P1, ZK_P2, ZK_P3};
  Fr acc = fr_R();  // 1 in Montgomery form
  for (int i = 253; i >= 0; --i) {
    acc = fr_mont_mul(acc, acc);
    if ((e[i >> 6] >> (i & 63)) & 1) acc = fr_mont_mul(acc, a);
  }
  return acc;
}
