// RSAVerifier65537(121,17) witness computation, wave-collective.
//
// One wavefront evaluates one email: the 17 sequential FpMul's of FpPow65537Mod
// (packages/circuits/lib/rsa.circom:57-92) and, for each, every hint and genuine field
// value of lib/fp.circom:16-81 / lib/bigint.circom:16-94:
//
//   q, r            2048-bit Barrett reduction: 64 x 32-bit limbs staged in LDS, the
//                   O(n^2) products as lane-parallel column sums (lane = product column)
//   v_ab, v_pq_r    lane = evaluation point x (0..32), BN254-Fr Montgomery products
//   t -> carry      lane = coefficient i (0..32), exact signed 256-bit integers
//   BigLessThan     lane = limb; the 306 IsEqual inverses are batch-inverted per lane
//
// The code is written as phases of ZK_PAR_FOR loops (one iteration per lane, no
// cross-iteration dependence) separated by ZK_SYNC(), plus ZK_SEQ blocks executed by
// lane 0.  Compiled for the host the same phases run sequentially, which is how the
// unit test (tests/native) checks this file against the oracle without a GPU.
#pragma once
#include "zkwg_sched.h"

#if defined(__HIPCC__)
#define ZK_LANE() (threadIdx.x & 63u)
#define ZK_PAR_FOR(i, n) for (u32 i = ZK_LANE(); i < (u32)(n); i += 64u)
#define ZK_SEQ if (ZK_LANE() == 0u)
#define ZK_SYNC() __syncthreads()
#define ZK_DEV __device__
#elif defined(ZKWG_WAVESIM)   // tests/native/wavesim.h: the 64 lanes as fibers of one host thread
#define ZK_LANE() zk_wavesim_lane()
#define ZK_PAR_FOR(i, n) for (u32 i = ZK_LANE(); i < (u32)(n); i += 64u)
#define ZK_SEQ if (ZK_LANE() == 0u)
#define ZK_SYNC() zk_wavesim_sync()
#define ZK_DEV
#else
#define ZK_PAR_FOR(i, n) for (u32 i = 0; i < (u32)(n); ++i)
#define ZK_SEQ
#define ZK_SYNC()
#define ZK_DEV
#endif

#define ZK_RSA_K 17
#define ZK_RSA_N 121
#define ZK_BIG 68    // u32 limbs per big number buffer (>= 66)
#define ZK_BIG2 136  // double-width buffers

struct u256s {  // signed/unsigned 256-bit integer, two's complement, little-endian u64 limbs
  u64 l[4];
};

struct ZkRsaLds {           // per-wave working set (LDS on the GPU)
  u32 a[ZK_BIG], b[ZK_BIG], p[ZK_BIG], mu[ZK_BIG], base[ZK_BIG];
  u32 q1[ZK_BIG], q3[ZK_BIG], r[ZK_BIG];
  u32 x[ZK_BIG2], q2[ZK_BIG2], t[ZK_BIG2];
  u32 cs[ZK_BIG2][3];       // column sums (96-bit)
  u64 a121[ZK_RSA_K][2], b121[ZK_RSA_K][2], p121[ZK_RSA_K][2], q121[ZK_RSA_K][2], r121[ZK_RSA_K][2];
  u64 s121[ZK_RSA_K][2], m121[ZK_RSA_K][2];  // signature, message
  u256s tt[33];
  u256s tq[33];             // device path: column sums of p*q (tt holds those of a*b)
  u32 L;                    // bit length of the modulus
  u32 ok;                   // assertion flag (0 = some constraint failed)
};

// ------------------------------------------------------------------ small helpers
ZK_HD u32 zk_minu(u32 a, u32 b) { return a < b ? a : b; }
ZK_HD u64 zk_mask64(u32 n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }

// bits [pos, pos+32) of the integer sum_i limb[i] * 2^(121 i), limbs < 2^121
ZK_HD u32 zk_bits_from_121(const u64 (*l)[2], u32 pos) {
  u32 out = 0, got = 0;
  while (got < 32) {
    u32 i = (pos + got) / ZK_RSA_N, off = (pos + got) % ZK_RSA_N;
    if (i >= ZK_RSA_K) break;
    u32 take = zk_minu(32u - got, ZK_RSA_N - off);
    u64 w = off < 64 ? (l[i][0] >> off) | (off ? (l[i][1] << (64 - off)) : 0) : (l[i][1] >> (off - 64));
    out |= (u32)(w & zk_mask64(take)) << got;
    got += take;
  }
  return out;
}
// 121-bit limb i of the integer held in 32-bit limbs x[0..n)
ZK_HD void zk_limb121_from_32(const u32* x, u32 n, u32 i, u64* out2) {
  u64 lo = 0, hi = 0;
  u32 pos = i * ZK_RSA_N;
  for (u32 got = 0; got < ZK_RSA_N;) {
    u32 w = (pos + got) >> 5, off = (pos + got) & 31;
    u32 take = zk_minu(32u - off, ZK_RSA_N - got);
    u64 v = w < n ? ((u64)(x[w] >> off) & zk_mask64(take)) : 0;
    if (got < 64) {
      lo |= v << got;
      if (got + take > 64) hi |= v >> (64 - got);
    } else {
      hi |= v << (got - 64);
    }
    got += take;
  }
  out2[0] = lo; out2[1] = hi;
}

// ------------------------------------------------------------------ 256-bit integers
ZK_HD u256s u256_zero() { return u256s{{0, 0, 0, 0}}; }
ZK_HD u256s u256_add(const u256s& a, const u256s& b) {
  u256s r; u64 c = 0;
  for (int i = 0; i < 4; ++i) r.l[i] = zk_adc(a.l[i], b.l[i], c);
  return r;
}
ZK_HD u256s u256_sub(const u256s& a, const u256s& b) {
  u256s r; u64 c = 0;
  for (int i = 0; i < 4; ++i) r.l[i] = zk_sbb(a.l[i], b.l[i], c);
  return r;
}
ZK_HD bool u256_is_neg(const u256s& a) { return (a.l[3] >> 63) != 0; }
ZK_HD bool u256_is_zero(const u256s& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
// a * x + y, x small (x < 2^32); y = 128-bit (two u64)
ZK_HD u256s u256_mul_small_add(const u256s& a, u32 x, const u64* y2) {
  // 8 x (32x32+64) multiply-adds: one v_mad_u64_u32 each on gfx950
  u32 al[8];
  for (int i = 0; i < 4; ++i) { al[2 * i] = (u32)a.l[i]; al[2 * i + 1] = (u32)(a.l[i] >> 32); }
  const u32 yl[4] = {(u32)y2[0], (u32)(y2[0] >> 32), (u32)y2[1], (u32)(y2[1] >> 32)};
  u32 rl[8];
  u64 c = 0;
  for (int i = 0; i < 8; ++i) {
    c = (u64)al[i] * x + (i < 4 ? yl[i] : 0u) + c;
    rl[i] = (u32)c;
    c >>= 32;
  }
  u256s r;
  for (int i = 0; i < 4; ++i) r.l[i] = (u64)rl[2 * i] | ((u64)rl[2 * i + 1] << 32);
  return r;
}
// (a1:a0) * (b1:b0) -> 256-bit
ZK_HD u256s u256_mul128(const u64* a, const u64* b) {
  u64 p00l, p00h, p01l, p01h, p10l, p10h, p11l, p11h;
  zk_mul64(a[0], b[0], p00l, p00h);
  zk_mul64(a[0], b[1], p01l, p01h);
  zk_mul64(a[1], b[0], p10l, p10h);
  zk_mul64(a[1], b[1], p11l, p11h);
  u256s r;
  r.l[0] = p00l;
  u64 c = 0;
  u64 m = zk_adc(p00h, p01l, c);
  u64 c2 = 0;
  m = zk_adc(m, p10l, c2);
  r.l[1] = m;
  u64 carry1 = c + c2;  // into limb 2
  c = 0;
  u64 h = zk_adc(p01h, p10h, c);
  u64 c3 = 0;
  h = zk_adc(h, p11l, c3);
  u64 c4 = 0;
  h = zk_adc(h, carry1, c4);
  r.l[2] = h;
  r.l[3] = p11h + c + c3 + c4;
  return r;
}
// arithmetic shift right by 121
ZK_HD u256s u256_sar121(const u256s& a) {
  u256s r;
  u64 sign = u256_is_neg(a) ? ~0ull : 0ull;
  // 121 = 64 + 57
  r.l[0] = (a.l[1] >> 57) | (a.l[2] << 7);
  r.l[1] = (a.l[2] >> 57) | (a.l[3] << 7);
  r.l[2] = (a.l[3] >> 57) | (sign << 7);
  r.l[3] = sign;
  return r;
}
// signed integer -> field element (negative -> r - |v|); |v| < r
ZK_HD Fr fr_from_signed(const u256s& v) {
  Fr f{{v.l[0], v.l[1], v.l[2], v.l[3]}};
  if (u256_is_neg(v)) {
    u64 c;
    f = fr_add_raw(f, fr_p(), c);  // two's complement wrap: v + r (mod 2^256)
  }
  return f;
}
ZK_HD Fr fr_mul_std(const Fr& a, const Fr& b) {  // standard-form product
  return fr_mont_mul(fr_mont_mul(a, b), fr_R2());
}

// a^{-1} mod r (standard form), binary extended Euclid; a != 0, a < r
ZK_HD Fr fr_inv_std(const Fr& a) {
  const Fr P = fr_p();
  Fr u = a, v = P, x1 = fr_from_u64(1), x2 = fr_zero();
  const Fr one = fr_from_u64(1);
  auto halve_mod = [&](Fr& x) {
    u64 top = 0;
    if (x.l[0] & 1) { x = fr_add_raw(x, P, top); }
    x.l[0] = (x.l[0] >> 1) | (x.l[1] << 63);
    x.l[1] = (x.l[1] >> 1) | (x.l[2] << 63);
    x.l[2] = (x.l[2] >> 1) | (x.l[3] << 63);
    x.l[3] = (x.l[3] >> 1) | (top << 63);
  };
  auto shr1 = [&](Fr& x) {
    x.l[0] = (x.l[0] >> 1) | (x.l[1] << 63);
    x.l[1] = (x.l[1] >> 1) | (x.l[2] << 63);
    x.l[2] = (x.l[2] >> 1) | (x.l[3] << 63);
    x.l[3] >>= 1;
  };
  for (int guard = 0; guard < 1200 && !fr_eq(u, one) && !fr_eq(v, one); ++guard) {
    while (!(u.l[0] & 1)) { shr1(u); halve_mod(x1); }
    while (!(v.l[0] & 1)) { shr1(v); halve_mod(x2); }
    if (fr_geq(u, v)) {
      u64 bw; u = fr_sub_raw(u, v, bw);
      x1 = fr_sub(x1, x2);
    } else {
      u64 bw; v = fr_sub_raw(v, u, bw);
      x2 = fr_sub(x2, x1);
    }
  }
  return fr_eq(u, one) ? x1 : x2;
}

// ------------------------------------------------------------------ big-number primitives (wave-collective)
// out[0..na+nb) = a[0..na) * b[0..nb): lane-parallel column sums, then one carry pass.
ZK_DEV inline void zk_wave_mul(ZkRsaLds& S, u32* out, const u32* a, u32 na, const u32* b, u32 nb) {
  const u32 nc = na + nb;
  ZK_PAR_FOR(c, nc) {
    u64 lo = 0; u32 hi = 0;  // 96-bit accumulator
    u32 i0 = c >= nb ? c - nb + 1 : 0, i1 = zk_minu(c, na - 1);
    for (u32 i = i0; i <= i1 && i1 < na; ++i) {
      u64 pr = (u64)a[i] * (u64)b[c - i];
      u64 s = lo + pr;
      hi += (s < lo);
      lo = s;
    }
    S.cs[c][0] = (u32)lo; S.cs[c][1] = (u32)(lo >> 32); S.cs[c][2] = hi;
  }
  ZK_SYNC();
  ZK_SEQ {
    u64 carry_lo = 0; u32 carry_hi = 0;  // 96-bit running carry
    for (u32 c = 0; c < nc; ++c) {
      u64 col = (u64)S.cs[c][0] | ((u64)S.cs[c][1] << 32);
      u64 s = carry_lo + col;
      u32 h = carry_hi + S.cs[c][2] + (s < carry_lo);
      out[c] = (u32)s;
      carry_lo = (s >> 32) | ((u64)h << 32);
      carry_hi = 0;
    }
  }
  ZK_SYNC();
}
// out[0..nout) = x[0..nx) >> sh
ZK_DEV inline void zk_wave_shr(u32* out, u32 nout, const u32* x, u32 nx, u32 sh) {
  const u32 ws = sh >> 5, bs = sh & 31;
  ZK_PAR_FOR(w, nout) {
    u32 lo = (w + ws) < nx ? x[w + ws] : 0;
    u32 hi = (w + ws + 1) < nx ? x[w + ws + 1] : 0;
    out[w] = bs ? (lo >> bs) | (hi << (32 - bs)) : lo;
  }
  ZK_SYNC();
}
// lane-0 helpers
ZK_HD int zk_seq_cmp(const u32* a, const u32* b, u32 n) {
  for (u32 i = n; i-- > 0;) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return -1;
  }
  return 0;
}
ZK_HD void zk_seq_sub(u32* out, const u32* a, const u32* b, u32 n) {
  u32 borrow = 0;
  for (u32 i = 0; i < n; ++i) {
    u64 d = (u64)a[i] - b[i] - borrow;
    out[i] = (u32)d;
    borrow = (u32)(d >> 63);
  }
}
ZK_HD u32 zk_seq_bitlen(const u32* a, u32 n) {
  for (u32 i = n; i-- > 0;)
    if (a[i]) return 32 * i + (32 - __builtin_clz(a[i]));
  return 0;
}

// mu = floor(2^(2L) / p), schoolbook long division (Knuth D) by lane 0.  p has L bits.
ZK_DEV inline void zk_seq_barrett_mu(ZkRsaLds& S) {
  ZK_SEQ {
    const u32 L = S.L;
    const u32 nd = (L + 31) / 32;          // divisor limbs
    const u32 sh = (32 - (L & 31)) & 31;   // normalisation shift
    // normalised divisor in S.t[0..nd), numerator 2^(2L) << sh in S.x[0..nn]
    u32* d = S.t;
    for (u32 i = 0; i < nd; ++i) d[i] = sh ? (S.p[i] << sh) | (i ? S.p[i - 1] >> (32 - sh) : 0) : S.p[i];
    const u32 nbit = 2 * L + sh;
    const u32 nn = nbit / 32 + 1;
    u32* num = S.x;
    for (u32 i = 0; i <= nn; ++i) num[i] = 0;
    num[nbit >> 5] = 1u << (nbit & 31);
    for (u32 i = 0; i < ZK_BIG; ++i) S.mu[i] = 0;
    const u64 dtop = d[nd - 1], dsec = nd > 1 ? d[nd - 2] : 0;
    for (u32 j = nn - nd + 1; j-- > 0;) {   // quotient digit j
      u64 top = ((u64)num[j + nd] << 32) | num[j + nd - 1];
      u64 qhat = top / dtop, rhat = top % dtop;
      while (qhat >= (1ull << 32) || (nd > 1 && qhat * dsec > ((rhat << 32) | num[j + nd - 2]))) {
        --qhat; rhat += dtop;
        if (rhat >= (1ull << 32)) break;
      }
      // num[j..j+nd] -= qhat * d
      u64 borrow = 0, carry = 0;
      for (u32 i = 0; i < nd; ++i) {
        u64 pr = qhat * d[i] + carry;
        carry = pr >> 32;
        u64 sub = (u64)num[j + i] - (u32)pr - borrow;
        num[j + i] = (u32)sub;
        borrow = (sub >> 63) & 1;
      }
      u64 sub = (u64)num[j + nd] - carry - borrow;
      num[j + nd] = (u32)sub;
      if ((sub >> 63) & 1) {  // add back
        --qhat;
        u64 c = 0;
        for (u32 i = 0; i < nd; ++i) {
          u64 s = (u64)num[j + i] + d[i] + c;
          num[j + i] = (u32)s;
          c = s >> 32;
        }
        num[j + nd] += (u32)c;
      }
      if (j < ZK_BIG) S.mu[j] = (u32)qhat;
    }
  }
  ZK_SYNC();
}

// (q3, r) = divmod(a * b, p) via Barrett (HAC 14.42, base 2).  Requires a*b < 2^(2L).
// Results: S.q3[0..66) quotient, S.r[0..66) remainder.  Sets S.ok = 0 if the reduction
// cannot be completed (operands out of range -- the email fails its range checks anyway).
ZK_DEV inline void zk_wave_mulmod(ZkRsaLds& S) {
  const u32 L = S.L;
  zk_wave_mul(S, S.x, S.a, 65, S.b, 65);                 // x = a*b (130 limbs)
  zk_wave_shr(S.q1, 66, S.x, 130, L - 1);                // q1 = x >> (L-1)
  zk_wave_mul(S, S.q2, S.q1, 66, S.mu, 66);              // q2 = q1 * mu
  zk_wave_shr(S.q3, 66, S.q2, 132, L + 1);               // q3 = q2 >> (L+1)
  zk_wave_mul(S, S.t, S.q3, 66, S.p, 65);                // t = q3 * p
  ZK_SEQ {
    zk_seq_sub(S.r, S.x, S.t, 67);                       // r = x - t (low 67 limbs)
    int it = 0;
    S.p[65] = S.p[66] = 0;
    while (zk_seq_cmp(S.r, S.p, 67) >= 0) {
      if (++it > 3) { S.ok = 0; break; }
      zk_seq_sub(S.r, S.r, S.p, 67);
      for (u32 i = 0; i < 66; ++i) { if (++S.q3[i]) break; }
    }
  }
  ZK_SYNC();
}

// ------------------------------------------------------------------ witness pieces
// BigLessThan(121,17)(a, b) (lib/bigint.circom:16-60).  Writes the LessThan Num2Bits(122)
// inputs, the IsEqual (out, diff) pairs -- diff = b_i - a_i is inverted later by
// zk_rsa_invert_all -- and the ors/ands/eq_ands gate outputs.  Returns `out` (a < b).
ZK_DEV inline u32 zk_blt_emit(const u64 (*a)[2], const u64 (*b)[2], const ZkBltLayout& Lb,
                              u64* bits, u32* small, Fr* frv, u32* lt_eq /* LDS scratch 34 */) {
  ZK_PAR_FOR(i, ZK_RSA_K) {
    // n2b.in = a + 2^121 - b   (LessThan(121): in[0] + (1<<n) - in[1])
    u64 c = 0, bw = 0;
    u64 lo = zk_adc(a[i][0], 0, c);
    u64 hi = a[i][1] + (1ull << 57) + c;
    u64 dlo = zk_sbb(lo, b[i][0], bw);
    u64 dhi = hi - b[i][1] - bw;
    bits[Lb.b_lt + 2 * i] = dlo;
    bits[Lb.b_lt + 2 * i + 1] = dhi;
    lt_eq[i] = 1u - (u32)((dhi >> 57) & 1);  // out = 1 - n2b.out[121]
    // isz.in = in[1] - in[0] = b - a (signed 122-bit) -> field element
    u256s d;
    bw = 0;
    d.l[0] = zk_sbb(b[i][0], a[i][0], bw);
    d.l[1] = zk_sbb(b[i][1], a[i][1], bw);
    d.l[2] = zk_sbb(0, 0, bw);
    d.l[3] = zk_sbb(0, 0, bw);
    const bool zero = u256_is_zero(d);
    lt_eq[17 + i] = zero ? 1u : 0u;
    frv[Lb.f_eq + 2 * i] = fr_from_u64(zero ? 1 : 0);
    frv[Lb.f_eq + 2 * i + 1] = fr_from_signed(d);
  }
  ZK_SYNC();
  u32 out = 0;
  ZK_SEQ {
    const u32* lt = lt_eq;
    const u32* eq = lt_eq + 17;
    u32 ors_n = 0, eqa_n = 0;
    for (int i = ZK_RSA_K - 2; i >= 0; --i) {
      u32 ands_i, eqa_i, ors_i;
      if (i == ZK_RSA_K - 2) {
        ands_i = eq[ZK_RSA_K - 1] & lt[ZK_RSA_K - 2];
        eqa_i = eq[ZK_RSA_K - 1] & eq[ZK_RSA_K - 2];
        ors_i = lt[ZK_RSA_K - 1] | ands_i;
      } else {
        ands_i = eqa_n & lt[i];
        eqa_i = eqa_n & eq[i];
        ors_i = ors_n | ands_i;
      }
      small[Lb.m_gates + i] = ors_i;
      small[Lb.m_gates + 16 + i] = ands_i;
      small[Lb.m_gates + 32 + i] = eqa_i;
      ors_n = ors_i; eqa_n = eqa_i;
    }
    lt_eq[34] = ors_n;
  }
  ZK_SYNC();
  out = lt_eq[34];
  ZK_SYNC();
  return out;
}

// All signals of one FpMul(121,17) given a, b, p, q, r as 121-bit limbs in S
// (lib/fp.circom:25-76, lib/bigint.circom:69-94).
ZK_DEV inline void zk_fpmul_emit(ZkRsaLds& S, const ZkFpMulLayout& F, u64* bits, u32* small, Fr* frv,
                                 u32* lt_eq) {
  // v_ab[x] = A(x) * B(x), v_pq_r[x] = P(x) Q(x) + R(x) for x = 0..32.  The Horner values are
  // < 17 * 2^121 * 32^16 < 2^206, i.e. already reduced.
  ZK_PAR_FOR(x, 2 * ZK_RSA_K - 1) {
    u256s va = u256_zero(), vb = u256_zero(), vp = u256_zero(), vq = u256_zero(), vr = u256_zero();
    for (int i = ZK_RSA_K - 1; i >= 0; --i) {
      va = u256_mul_small_add(va, x, S.a121[i]);
      vb = u256_mul_small_add(vb, x, S.b121[i]);
      vp = u256_mul_small_add(vp, x, S.p121[i]);
      vq = u256_mul_small_add(vq, x, S.q121[i]);
      vr = u256_mul_small_add(vr, x, S.r121[i]);
    }
    Fr fa{{va.l[0], va.l[1], va.l[2], va.l[3]}}, fb{{vb.l[0], vb.l[1], vb.l[2], vb.l[3]}};
    Fr fp_{{vp.l[0], vp.l[1], vp.l[2], vp.l[3]}}, fq{{vq.l[0], vq.l[1], vq.l[2], vq.l[3]}};
    Fr fr_{{vr.l[0], vr.l[1], vr.l[2], vr.l[3]}};
    frv[F.f_main + x] = fr_mul_std(fa, fb);
    frv[F.f_main + 67 + x] = fr_add(fr_mul_std(fp_, fq), fr_);
  }
  ZK_PAR_FOR(i, ZK_RSA_K) {
    frv[F.f_main + 33 + i] = Fr{{S.q121[i][0], S.q121[i][1], 0, 0}};
    frv[F.f_main + 50 + i] = Fr{{S.r121[i][0], S.r121[i][1], 0, 0}};
    bits[F.b_qr + 2 * i] = S.q121[i][0];
    bits[F.b_qr + 2 * i + 1] = S.q121[i][1];
    bits[F.b_qr + 34 + 2 * i] = S.r121[i][0];
    bits[F.b_qr + 34 + 2 * i + 1] = S.r121[i][1];
  }
  // t[i] = sum_j a[j] b[i-j] - p[j] q[i-j]  - r[i]   (exact signed integers)
  ZK_PAR_FOR(i, 2 * ZK_RSA_K - 1) {
    u256s acc = u256_zero();
    int j0 = (int)i - (ZK_RSA_K - 1);
    if (j0 < 0) j0 = 0;
    int j1 = (int)i < ZK_RSA_K - 1 ? (int)i : ZK_RSA_K - 1;
    for (int j = j0; j <= j1; ++j) {
      acc = u256_add(acc, u256_mul128(S.a121[j], S.b121[i - j]));
      acc = u256_sub(acc, u256_mul128(S.p121[j], S.q121[i - j]));
    }
    if (i < ZK_RSA_K) acc = u256_sub(acc, u256s{{S.r121[i][0], S.r121[i][1], 0, 0}});
    S.tt[i] = acc;
  }
  ZK_SYNC();
  ZK_SEQ {  // CheckCarryToZero(121, 249, 33): carry[i] = (t[i] + carry[i-1]) / 2^121 (exact)
    u256s c = u256_zero();
    const u64 low_mask = (1ull << 57) - 1;
    for (u32 i = 0; i < 2 * ZK_RSA_K - 2; ++i) {
      u256s sum = u256_add(S.tt[i], c);
      if (sum.l[0] != 0 || (sum.l[1] & low_mask) != 0) S.ok = 0;  // in + carry === carry * 2^121
      c = u256_sar121(sum);
      frv[F.f_carry + i] = fr_from_signed(c);
      // carryRangeChecks[i].in = carry + 2^130, must fit Num2Bits(131)
      u256s rc = u256_add(c, u256s{{0, 0, 1ull << 2, 0}});
      if (u256_is_neg(rc) || rc.l[3] != 0 || (rc.l[2] >> 3) != 0) S.ok = 0;
      bits[F.b_carry + 3 * i] = rc.l[0];
      bits[F.b_carry + 3 * i + 1] = rc.l[1];
      bits[F.b_carry + 3 * i + 2] = rc.l[2];
    }
    if (!u256_is_zero(u256_add(S.tt[2 * ZK_RSA_K - 2], c))) S.ok = 0;  // in[k-1] + carry[k-2] === 0
    frv[F.f_carry + 2 * ZK_RSA_K - 2] = fr_zero();                      // carry[k-1] is never assigned
  }
  ZK_SYNC();
  u32 lt = zk_blt_emit(S.r121, S.p121, F.blt, bits, small, frv, lt_eq);  // r_p_lt_check.out === 1
  ZK_SEQ { if (!lt) S.ok = 0; }
  ZK_SYNC();
}

// Invert the 18 x 17 IsEqual differences of one email in place (Montgomery's trick per
// lane, one binary-Euclid inversion per lane).  Zero differences keep inv = 0.
ZK_DEV inline void zk_rsa_invert_all(const ZkRsaLayout& R, Fr* frv) {
  const u32 total = 18 * ZK_RSA_K;
  auto slot_of = [&](u32 e) -> u32 {
    u32 blk = e / ZK_RSA_K, i = e - blk * ZK_RSA_K;
    const ZkBltLayout& Lb = blk == 0 ? R.blt : R.mul[blk - 1].blt;
    return Lb.f_eq + 2 * i + 1;
  };
  ZK_PAR_FOR(lane, 64) {
    Fr pre[5];
    Fr acc = fr_from_u64(1);
    u32 cnt = 0;
    for (u32 e = lane; e < total; e += 64, ++cnt) {
      pre[cnt] = acc;
      Fr v = frv[slot_of(e)];
      if (!fr_is_zero(v)) acc = fr_mul_std(acc, v);
    }
    Fr inv = fr_inv_std(acc);
    for (u32 k = cnt; k-- > 0;) {
      u32 sl = slot_of(lane + 64 * k);
      Fr v = frv[sl];
      if (!fr_is_zero(v)) {
        frv[sl] = fr_mul_std(inv, pre[k]);
        inv = fr_mul_std(inv, v);
      }
    }
  }
  ZK_SYNC();
}

// The whole RSAVerifier65537(121,17) of one email.  `rec` = packed input record;
// `digest` = header SHA-256 state words (EmailVerifier) or nullptr (message from record).
ZK_DEV inline void zk_rsa_email(ZkRsaLds& S, const ZkRsaLayout& R, const u8* rec, const u32* digest,
                                u64* bits, u32* small, Fr* frv, u32* lt_eq) {
  const u64 top_mask = (1ull << 57) - 1;
  ZK_SEQ { S.ok = 1; }
  ZK_SYNC();
  ZK_PAR_FOR(i, ZK_RSA_K) {
    const u64* pm = (const u64*)(rec + R.in_mod + 16 * i);
    const u64* ps = (const u64*)(rec + R.in_sig + 16 * i);
    S.p121[i][0] = pm[0]; S.p121[i][1] = pm[1] & top_mask;
    S.s121[i][0] = ps[0]; S.s121[i][1] = ps[1] & top_mask;
    bool bad = (pm[1] >> 57) != 0 || (ps[1] >> 57) != 0;   // Num2Bits(121) range checks
    u64 m0 = 0, m1 = 0;
    if (digest) {  // rsaMessage[i]: the digest as a 256-bit big-endian integer, 121-bit limbs
      u32 w32[8];
      for (int j = 0; j < 8; ++j) w32[j] = digest[7 - j];  // little-endian 32-bit limbs
      u64 tmp[2];
      zk_limb121_from_32(w32, 8, i, tmp);
      m0 = tmp[0]; m1 = tmp[1];
    } else {
      const u64* pmsg = (const u64*)(rec + R.in_msg + 16 * i);
      m0 = pmsg[0]; m1 = pmsg[1] & top_mask;
      bad = bad || (pmsg[1] >> 57) != 0;
    }
    S.m121[i][0] = m0; S.m121[i][1] = m1;
    if (bad) S.ok = 0;
    // Num2Bits(121) images
    bits[R.b_modbits + 2 * i] = S.p121[i][0]; bits[R.b_modbits + 2 * i + 1] = S.p121[i][1];
    bits[R.b_msgbits + 2 * i] = m0;           bits[R.b_msgbits + 2 * i + 1] = m1;
    bits[R.b_sigbits + 2 * i] = S.s121[i][0]; bits[R.b_sigbits + 2 * i + 1] = S.s121[i][1];
  }
  ZK_SYNC();
  ZK_PAR_FOR(w, ZK_BIG) {
    S.p[w] = w < 65 ? zk_bits_from_121(S.p121, 32 * w) : 0;
    S.base[w] = w < 65 ? zk_bits_from_121(S.s121, 32 * w) : 0;
    S.a[w] = S.base[w];
    S.b[w] = S.base[w];
  }
  ZK_SYNC();
  ZK_SEQ {
    S.L = zk_seq_bitlen(S.p, 65);
    // RSAPad (lib/rsa.circom:101-181): messageBits[256..] === 0
    if ((S.m121[2][0] >> 14) != 0 || S.m121[2][1] != 0) S.ok = 0;
    for (int i = 3; i < ZK_RSA_K; ++i) if (S.m121[i][0] | S.m121[i][1]) S.ok = 0;
    // paddedMessageBits[416..480] === 1  <=>  modulus has a bit set at position >= 488
    if (S.L < 489) S.ok = 0;
  }
  ZK_SYNC();
  // modulusZero[idx].in = popcount of modulus bits at positions >= 424 + 8 idx   (idx 0..204)
  ZK_PAR_FOR(idx, 205) {
    u32 b0 = 424 + 8 * idx, cnt = 0;
    u32 w0 = b0 >> 5;
    cnt += __builtin_popcount(S.p[w0] >> (b0 & 31));
    for (u32 w = w0 + 1; w < 65; ++w) cnt += __builtin_popcount(S.p[w]);
    small[R.m_modzero + idx] = cnt;
  }
  u32 sig_lt = zk_blt_emit(S.s121, S.p121, R.blt, bits, small, frv, lt_eq);  // bigLessThan.out === 1
  ZK_SEQ { if (!sig_lt) S.ok = 0; }
  ZK_SYNC();
  const bool chain = S.L >= 2;
  if (chain) zk_seq_barrett_mu(S);
  for (u32 m = 0; m < 17; ++m) {
    // doublers[m]: a = b = previous result; adder (m == 16): a = base, b = doublers[15].out
    ZK_PAR_FOR(i, ZK_RSA_K) {
      if (m == 0) {
        S.a121[i][0] = S.s121[i][0]; S.a121[i][1] = S.s121[i][1];
        S.b121[i][0] = S.s121[i][0]; S.b121[i][1] = S.s121[i][1];
      } else if (m < 16) {
        S.a121[i][0] = S.r121[i][0]; S.a121[i][1] = S.r121[i][1];
        S.b121[i][0] = S.r121[i][0]; S.b121[i][1] = S.r121[i][1];
      } else {
        S.a121[i][0] = S.s121[i][0]; S.a121[i][1] = S.s121[i][1];
        S.b121[i][0] = S.r121[i][0]; S.b121[i][1] = S.r121[i][1];
      }
    }
    ZK_PAR_FOR(w, ZK_BIG) {
      if (m > 0) {
        u32 rv = w < 66 ? S.r[w] : 0;
        S.b[w] = rv;
        S.a[w] = (m < 16) ? rv : S.base[w];
      }
    }
    ZK_SYNC();
    if (chain) {
      zk_wave_mulmod(S);
    } else {
      ZK_PAR_FOR(w, ZK_BIG) { S.q3[w] = 0; S.r[w] = 0; }
      ZK_SYNC();
    }
    ZK_SEQ {  // q must fit 17 x 121 bits
      if (zk_seq_bitlen(S.q3, 66) > ZK_RSA_K * ZK_RSA_N) S.ok = 0;
    }
    ZK_PAR_FOR(i, ZK_RSA_K) {
      zk_limb121_from_32(S.q3, 66, i, S.q121[i]);
      zk_limb121_from_32(S.r, 66, i, S.r121[i]);
    }
    ZK_SYNC();
    zk_fpmul_emit(S, R.mul[m], bits, small, frv, lt_eq);
  }
  zk_rsa_invert_all(R, frv);
  // bigPow.out[i] === padder.out[i] (lib/rsa.circom:43-45): expected EMSA-PKCS1-v1_5 value
  ZK_SEQ {
    // ones run: bit i (>= 416) is 1 iff m8(i) + 8 <= highest set modulus bit, m8 = i rounded up to 8
    const u32 hb = S.L ? S.L - 1 : 0;
    u32* P = S.t;  // expected padded message, 32-bit limbs
    for (u32 w = 0; w < 66; ++w) P[w] = 0;
    for (u32 w = 0; w < 8; ++w) P[w] = zk_bits_from_121(S.m121, 32 * w);
    // DigestInfo 0x3031300d060960864801650304020105000420 at bits [256, 408)
    const u32 di[5] = {0x05000420u, 0x03040201u, 0x86480165u, 0x0d060960u, 0x00303130u};
    for (u32 w = 0; w < 5; ++w) P[8 + w] |= di[w];
    for (u32 i = 416; i < ZK_RSA_K * ZK_RSA_N; ++i) {
      u32 m8 = (i & 7) ? ((i >> 3) + 1) << 3 : i;
      if (S.L && m8 + 8 <= hb) P[i >> 5] |= 1u << (i & 31);
    }
    for (u32 w = 0; w < 65; ++w) if (P[w] != S.r[w]) S.ok = 0;
    if (S.r[65] != 0) S.ok = 0;
  }
  ZK_SYNC();
}
