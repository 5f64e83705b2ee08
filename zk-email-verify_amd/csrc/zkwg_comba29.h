// a * b * 2^-256 mod p for a 254-bit prime, as a drop-in for the 8 x 32-bit CIOS products of zkwg_fr.h / zkwg_fq.h on the device.
//
// tools/mulbench.hip (profiles/r05/r05_b_mulbench.json): v_mad_u64_u32 issues at nearly the plain VALU rate on gfx950, so the cost
// of the CIOS is not its 128 multiply-adds but the carry handling around them (120 64-bit adds + 286 moves as compiled: 562
// instructions).  With 29-bit limbs a column of the schoolbook product -- 9 products of the operands, 9 of the reduction, each below
// 2^58 -- fits a 64-bit accumulator: the whole product is 171 v_mad_u64_u32 chained through their 64-bit addend, one AND and one
// 64-bit shift per column, no carries (product scanning): 250 instructions, 1.70 x the CIOS's rate in a pure product loop.
//
// Nine 29-bit limbs reduce by 2^261, the callers' Montgomery form is 2^256: the first operand's limbs are cut from a << 5 (the same
// shifts at other offsets: free), so the result is a b 2^5 / 2^261 = a b / 2^256.  In and out: the callers' 4 x 64-bit words,
// canonical (< p); conversions ~60 instructions.  P = { nine 29-bit limbs of p, -p^-1 mod 2^29 }.
#pragma once
// (included by zkwg_fr.h after its integer helpers)

#define ZK29_MASK 0x1fffffffu
struct ZkComba29P { u32 p[9]; u32 n0; };

// limbs of (x << SH) for SH in {0, 5}: x < 2^254 as 4 x u64
template <int SH>
ZK_HD void zk29_split(const u64 x[4], u32 (&l)[9]) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i - SH;               // first bit of the limb inside x
    u64 v;
    if (bit < 0) v = x[0] << (-bit);
    else {
      const int k = bit >> 6, s = bit & 63;
      v = x[k] >> s;
      if (s > 64 - 29 && k + 1 < 4) v |= x[k + 1] << (64 - s);
    }
    l[i] = (u32)v & ZK29_MASK;
  }
}
// r = a b / 2^256 mod p (canonical); a, b canonical
ZK_HD void zk_comba29_mul(const u64 a[4], const u64 b[4], const ZkComba29P& P, const u64 p64[4], u64 (&out)[4]) {
  u32 A[9], B[9], q[9], r[9];
  zk29_split<5>(a, A);
  zk29_split<0>(b, B);
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (u64)A[i] * B[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (u64)q[i] * P.p[k - i];
    q[k] = ((u32)acc * P.n0) & ZK29_MASK;
    acc += (u64)q[k] * P.p[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)A[i] * B[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)q[i] * P.p[k - i];
    r[k - 9] = (u32)acc & ZK29_MASK;
    acc >>= 29;
  }
  r[8] = (u32)acc;                 // value < a b 2^5 / 2^261 + p < 2 p: the top limb stays below 2^23
  // pack 9 x 29 bits (normalised) into 4 x 64
  u64 w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, k = bit >> 6, s = bit & 63;
    w[k] |= (u64)r[i] << s;
    if (s > 64 - 29 && k + 1 < 4) w[k + 1] |= (u64)r[i] >> (64 - s);
  }
  // one conditional subtraction
  bool ge = true;
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    if (w[i] != p64[i]) { ge = w[i] > p64[i]; break; }
  }
  if (ge) {
    u64 bw = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = zk_sbb(w[i], p64[i], bw);
  }
  out[0] = w[0]; out[1] = w[1]; out[2] = w[2]; out[3] = w[3];
}
