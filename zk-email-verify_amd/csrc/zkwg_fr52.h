// BN254 Fr Montgomery products on the FP64 FMA pipe: 5 limbs of 52 bits held as doubles, R = 2^260.
//
// Why: gfx950 has no 64 x 64 multiplier and v_mad_u64_u32 issues at a fraction of the plain VALU rate (tools/mulbench.hip
// measures it), so the 8 x 32-bit CIOS of zkwg_fr.h costs 128 slow issues per product.  v_fma_f64 delivers the exact
// 106-bit product of two 53-bit integers in two issues (Emmart / Zheng / Weems, "Faster modular exponentiation using double
// precision floating point arithmetic on the GPU", ARITH 2018 -- PAPERS.md lists it as technique background): with the FP64 rounding
// mode set to round-toward-zero,
//     hi = fma(a, b, 2^104)            = 2^104 + 2^52 * floor(a b / 2^52)          (one binade: ulp 2^52)
//     lo = fma(a, b, (2^104 + 2^52) - hi) = 2^52 + (a b mod 2^52)
// and the mantissa fields of hi / lo ARE the two halves as integers, so column sums are 64-bit integer adds of the raw bit
// patterns (their exponent fields add up to constants that are subtracted once).
//
// Values stay in limb form between products and are only required to be < 2^257: a product of operands < 8 r is < 2 r
// (R / r > 2^6), so butterflies need no conditional subtraction.  A kernel that calls fr52_enter() runs FP64 in
// round-toward-zero mode to its end (MODE is per-wavefront state, reloaded at every launch); nothing else on the witness
// path uses FP64.  Host build: the same code under fesetround(FE_TOWARDZERO) (tests/native/hosttest.cpp).
#pragma once
#include "zkwg_fr.h"
#if !defined(__HIP_DEVICE_COMPILE__)
#include <cfenv>
#include <cmath>
#include <cstring>
#endif

struct Fr52 {
  double l[5];   // integers < 2^52 (the top limb of a value < 2^257 is < 2^49)
};
struct Fr52Ctx {
  double c1, c2, two52;   // 2^104, 2^104 + 2^52, 2^52 -- produced by fr52_enter so that every FMA depends on the mode switch
};

#define ZK52_B1 0x4670000000000000ull   // bit pattern of 2^104
#define ZK52_B2 0x4330000000000000ull   // bit pattern of 2^52
#define ZK52_M 0x000fffffffffffffull
#define ZK52_N0 0x1f593efffffffull      // -r^-1 mod 2^52

ZK_HD double zk52_as_double(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double((long long)x);
#else
  double d; memcpy(&d, &x, 8); return d;
#endif
}
ZK_HD u64 zk52_bits(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (u64)__double_as_longlong(d);
#else
  u64 x; memcpy(&x, &d, 8); return x;
#endif
}
// FP64 round-toward-zero from here to the end of the kernel
ZK_HD Fr52Ctx fr52_enter() {
  u64 z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  u32 z32;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\tv_mov_b32 %0, 0" : "=v"(z32));
  z = z32;
#else
  fesetround(FE_TOWARDZERO);
#endif
  Fr52Ctx cx;
  cx.c1 = zk52_as_double(ZK52_B1 | z);
  cx.c2 = zk52_as_double((ZK52_B1 + 1u) | z);
  cx.two52 = zk52_as_double(ZK52_B2 | z);
  return cx;
}
ZK_HD double zk52_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  return fma(a, b, c);
#endif
}
ZK_HD double zk52_to_double(u64 x, const Fr52Ctx& cx) { return zk52_as_double(x | ZK52_B2) - cx.two52; }   // x < 2^52, exact
ZK_HD u64 zk52_to_int(double d, const Fr52Ctx& cx) { return zk52_bits(d + cx.two52) & ZK52_M; }               // d an integer < 2^52

ZK_HD Fr52 fr52_from_fr(const Fr& x, const Fr52Ctx& cx) {
  Fr52 r;
  r.l[0] = zk52_to_double(x.l[0] & ZK52_M, cx);
  r.l[1] = zk52_to_double(((x.l[0] >> 52) | (x.l[1] << 12)) & ZK52_M, cx);
  r.l[2] = zk52_to_double(((x.l[1] >> 40) | (x.l[2] << 24)) & ZK52_M, cx);
  r.l[3] = zk52_to_double(((x.l[2] >> 28) | (x.l[3] << 36)) & ZK52_M, cx);
  r.l[4] = zk52_to_double(x.l[3] >> 16, cx);
  return r;
}
// limb form (value < 2^256) -> canonical Fr (fully reduced)
ZK_HD Fr fr52_to_fr(const Fr52& x, const Fr52Ctx& cx) {
  u64 t[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) t[i] = zk52_to_int(x.l[i], cx);
  Fr r{{t[0] | (t[1] << 52), (t[1] >> 12) | (t[2] << 40), (t[2] >> 24) | (t[3] << 28), (t[3] >> 36) | (t[4] << 16)}};
  for (int k = 0; k < 5 && fr_geq(r, fr_p()); ++k) {
    u64 bw;
    r = fr_sub_raw(r, fr_p(), bw);
  }
  return r;
}
// lazy add / sub in limb form: limbs stay below 2^52 only after fr52_norm; a product accepts limbs < 2^52, so sums are
// normalised (carry propagation on the integer pipe) before they are multiplied
ZK_HD Fr52 fr52_norm_u64(const u64 t_in[5], const Fr52Ctx& cx) {
  u64 t[5] = {t_in[0], t_in[1], t_in[2], t_in[3], t_in[4]};
#pragma unroll
  for (int i = 0; i < 4; ++i) { t[i + 1] += t[i] >> 52; t[i] &= ZK52_M; }
  Fr52 r;
#pragma unroll
  for (int i = 0; i < 5; ++i) r.l[i] = zk52_to_double(t[i], cx);
  return r;
}

// a b 2^-260 mod r, result < a b / 2^260 + r  (operands: limbs < 2^52; a b < 64 r^2 gives a result < 2 r)
ZK_HD Fr52 fr52_mul(const Fr52& a, const Fr52& b, const Fr52Ctx& cx) {
  const double P[5] = {(double)0x1f593f0000001ull, (double)0x4879b9709143eull, (double)0x181585d2833e8ull, (double)0xa029b85045b68ull, (double)0x30644e72e131ull};
  // column sums as integers; the exponent fields of all hi / lo patterns that will ever be added (product and reduction:
  // twice the schoolbook pattern) are subtracted up front -- every such constant is a multiple of 2^52, so the low 52 bits
  // of a column are right at any time, and the whole column is right once its last contribution has arrived
  u64 col[10] = {0x79a0000000000000ull, 0x6660000000000000ull, 0x5320000000000000ull, 0x3fe0000000000000ull, 0x2ca0000000000000ull,
                 0x2620000000000000ull, 0x3960000000000000ull, 0x4ca0000000000000ull, 0x5fe0000000000000ull, 0x7320000000000000ull};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const double hi = zk52_fma(a.l[i], b.l[j], cx.c1);
      const double lo = zk52_fma(a.l[i], b.l[j], cx.c2 - hi);
      col[i + j + 1] += zk52_bits(hi);
      col[i + j] += zk52_bits(lo);
    }
  }
  const double n0 = (double)ZK52_N0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double x = zk52_to_double(col[i] & ZK52_M, cx);
    const double qh = zk52_fma(x, n0, cx.c1);
    const double q = zk52_fma(x, n0, cx.c2 - qh) - cx.two52;     // x n0 mod 2^52
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const double hi = zk52_fma(q, P[j], cx.c1);
      const double lo = zk52_fma(q, P[j], cx.c2 - hi);
      col[i + j + 1] += zk52_bits(hi);
      col[i + j] += zk52_bits(lo);
    }
    col[i + 1] += col[i] >> 52;    // low 52 bits of col[i] are zero now
  }
  return fr52_norm_u64(col + 5, cx);
}
