// RSAVerifier65537(121,17) witness computation for gfx950 -- the wavefront-parallel device path.
//
// Same values, same image layout and same assertions as zkwg_rsa_core.h (whose phase-sequential code
// stays the host-testable restatement, tests/native), but no lane-0 serial passes:
//
//   carries / borrows   every big-number add, subtract and column-sum normalisation resolves its ripple
//                       with a generate/propagate look-ahead over the wavefront: two ballots and one
//                       64-bit add ((G << 1 | cin) + P) ^ P gives the carry into every lane at once;
//                       neighbours' partial sums travel by DPP/ds_bpermute shuffles, not through LDS
//   Barrett mu          Knuth D with the multiply-subtract of each quotient digit spread over the lanes
//   compare / correct   r >= p is the borrow-out of the parallel subtraction (no limb-by-limb compare)
//   CheckCarryToZero    carry[i] = th[i] + u[i]: th[i] = t[i] >> 121 is lane-local, u[i] is a +-1 style
//                       correction from a 33-step chain on 64-bit scalars fed by v_readlane
//   BigLessThan gates   prefix AND / OR over ballot masks
//   Horner values       P(x) once per email, A(x), B(x) re-used from the previous FpMul's R(x)
//   IsEqual inverses    one local Montgomery-trick batch per lane without domain conversions; the one
//                       inversion per lane is Bernstein-Yang safegcd (no data-dependent control flow)
//
// Reference gate sequence: packages/circuits/lib/rsa.circom:13-181, lib/fp.circom:16-81,
// lib/bigint.circom:16-94, lib/bigint-func.circom (long_div / poly_interp hints).
#pragma once
#include "zkwg_rsa_core.h"
#include "zkwg_fr_inv.h"

#if defined(__HIPCC__) || defined(ZKWG_WAVESIM)
#if defined(ZKWG_WAVESIM)   // host build on a simulated wavefront (tests/native/wavesim.h supplies __ballot, __shfl ...)
#define __device__
#define __forceinline__ inline
#endif

__device__ __forceinline__ u64 zkw_ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ u32 zkw_rl(u32 v, int lane) { return (u32)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ u64 zkw_rl64(u64 v, int lane) {
  return (u64)zkw_rl((u32)v, lane) | ((u64)zkw_rl((u32)(v >> 32), lane) << 32);
}
__device__ __forceinline__ u32 zkw_up(u32 v, u32 d) { return (u32)__shfl_up((int)v, d); }

// carry (or borrow) into every lane of one 64-column round.  G: lanes that generate, P: lanes that
// propagate (G & P = 0), cin: carry into lane 0.  Returns the arrival mask; cout = carry out of lane 63.
__device__ __forceinline__ u64 zkw_lookahead(u64 G, u64 P, u32 cin, u32& cout) {
  const u64 X = (G << 1) | (u64)cin;
  const u64 sum = P + X;
  cout = (u32)(G >> 63) | (sum < P ? 1u : 0u);
  return sum ^ P;
}

// out[0..na+nb) = a[0..na) * b[0..nb): lane = product column (96-bit column sums in registers), carries
// resolved per 64-column round; nothing but a, b and out touches LDS.
__device__ inline void zkw_mul(u32* out, const u32* a, u32 na, const u32* b, u32 nb) {
  const u32 nc = na + nb;
  const u32 lane = ZK_LANE();
  u32 cin = 0, g63 = 0, e63 = 0, f62 = 0, f63 = 0;
  for (u32 base = 0; base < nc; base += 64) {
    const u32 c = base + lane;
    u64 lo = 0; u32 hi = 0;
    if (c < nc) {
      const u32 i0 = c >= nb ? c - nb + 1 : 0, i1 = zk_minu(c, na - 1);
      for (u32 i = i0; i <= i1; ++i) {
        const u64 pr = (u64)a[i] * (u64)b[c - i];
        const u64 s = lo + pr;
        hi += (s < lo);
        lo = s;
      }
    }
    const u32 d = (u32)lo, e = (u32)(lo >> 32), f = hi;
    u32 e1 = zkw_up(e, 1), f2 = zkw_up(f, 2);
    if (lane == 0) { e1 = e63; f2 = f62; }
    if (lane == 1) f2 = f63;
    const u64 s0 = (u64)d + e1 + f2;            // < 3 * 2^32
    const u32 d0 = (u32)s0, g = (u32)(s0 >> 32);
    u32 gp = zkw_up(g, 1);
    if (lane == 0) gp = g63;
    const u64 s1 = (u64)d0 + gp;
    const u32 d1 = (u32)s1;
    const bool gen = (s1 >> 32) != 0;
    const u64 G = zkw_ballot(gen), P = zkw_ballot(!gen && d1 == 0xffffffffu);
    u32 cout;
    const u64 arr = zkw_lookahead(G, P, cin, cout);
    if (c < nc) out[c] = d1 + (u32)((arr >> lane) & 1);
    cin = cout;
    g63 = zkw_rl(g, 63); e63 = zkw_rl(e, 63); f62 = zkw_rl(f, 62); f63 = zkw_rl(f, 63);
  }
  ZK_SYNC();
}

// out = a - b over n limbs (n <= 128), returns the borrow out.  `commit_if_no_borrow`: write out only when
// the result is non-negative (used as "if (a >= b) a -= b").  out may alias a.
__device__ inline u32 zkw_sub(u32* out, const u32* a, const u32* b, u32 n, bool commit_if_no_borrow) {
  const u32 lane = ZK_LANE();
  u32 dig[2] = {0, 0};
  u32 bin = 0;
  for (u32 r = 0; r < 2; ++r) {
    const u32 c = r * 64 + lane;
    if (r * 64 >= n) break;
    const u32 av = c < n ? a[c] : 0u, bv = c < n ? b[c] : 0u;
    const u32 d = av - bv;
    const u64 G = zkw_ballot(av < bv), P = zkw_ballot(av == bv);
    u32 bout;
    const u64 arr = zkw_lookahead(G, P, bin, bout);
    dig[r] = d - (u32)((arr >> lane) & 1);
    bin = bout;
  }
  if (!(commit_if_no_borrow && bin)) {
    if (lane < n) out[lane] = dig[0];
    if (64 + lane < n) out[64 + lane] = dig[1];
  }
  ZK_SYNC();
  return bin;
}
// x += v (v = 0/1 uniform) over n limbs (n <= 128)
__device__ inline void zkw_inc(u32* x, u32 n, u32 v) {
  const u32 lane = ZK_LANE();
  u32 cin = v;
  for (u32 r = 0; r < 2; ++r) {
    const u32 c = r * 64 + lane;
    if (r * 64 >= n) break;
    const u32 xv = c < n ? x[c] : 0u;
    const u64 P = zkw_ballot(c < n && xv == 0xffffffffu);
    u32 cout;
    const u64 arr = zkw_lookahead(0, P, cin, cout);
    if (c < n) x[c] = xv + (u32)((arr >> lane) & 1);
    cin = cout;
  }
  ZK_SYNC();
}
// bit length of x[0..n) (n <= 128)
__device__ inline u32 zkw_bitlen(const u32* x, u32 n) {
  const u32 lane = ZK_LANE();
  for (int r = 1; r >= 0; --r) {
    if ((u32)r * 64 >= n) continue;
    const u32 c = r * 64 + lane;
    const u32 v = c < n ? x[c] : 0u;
    const u64 m = zkw_ballot(v != 0);
    if (m) {
      const u32 top = 63u - (u32)__builtin_clzll(m);
      const u32 w = x[r * 64 + top];
      return 32u * (r * 64 + top) + (32u - (u32)__builtin_clz(w));
    }
  }
  return 0;
}

// mu = floor(2^(2L) / p) (Knuth D).  The normalised divisor lives in registers (lane = limb), the
// running numerator in S.x; per quotient digit one multiply-subtract round with borrow look-ahead.
__device__ inline void zkw_barrett_mu(ZkRsaLds& S) {
  const u32 lane = ZK_LANE();
  const u32 L = S.L;
  const u32 nd = (L + 31) / 32;
  const u32 sh = (32 - (L & 31)) & 31;
  // normalised divisor limbs i = lane and i = 64 + lane (nd <= 65), 0 beyond nd
  u32 dv[2];
  for (u32 r = 0; r < 2; ++r) {
    const u32 i = r * 64 + lane;
    u32 v = 0;
    if (i < nd) v = sh ? (S.p[i] << sh) | (i ? S.p[i - 1] >> (32 - sh) : 0u) : S.p[i];
    dv[r] = v;
  }
  const u32 nbit = 2 * L + sh;
  const u32 nn = nbit / 32 + 1;
  u32* num = S.x;
  for (u32 i = lane; i <= nn + 1 && i < ZK_BIG2; i += 64) num[i] = 0;
  for (u32 i = lane; i < ZK_BIG; i += 64) S.mu[i] = 0;
  ZK_SYNC();
  if (lane == 0) num[nbit >> 5] = 1u << (nbit & 31);
  ZK_SYNC();
  const u32 top_r = (nd - 1) >> 6, top_l = (nd - 1) & 63;
  const u64 dtop = top_r ? (u64)zkw_rl(dv[1], 0) : (u64)(u32)__shfl((int)dv[0], (int)top_l);
  u64 dsec = 0;
  if (nd > 1) {
    const u32 sr = (nd - 2) >> 6, sl = (nd - 2) & 63;
    dsec = sr ? zkw_rl(dv[1], 0) : (u64)(u32)__shfl((int)dv[0], (int)sl);
  }
  const u32 ncol = nd + 1;                       // columns of the multiply-subtract (limb nd: the running top)
  for (u32 j = nn - nd + 1; j-- > 0;) {
    const u64 n2 = num[j + nd], n1 = num[j + nd - 1], n0 = nd > 1 ? num[j + nd - 2] : 0;
    const u64 top = (n2 << 32) | n1;
    u64 qhat = top / dtop, rhat = top % dtop;
    while (qhat >= (1ull << 32) || (nd > 1 && qhat * dsec > ((rhat << 32) | n0))) {
      --qhat; rhat += dtop;
      if (rhat >= (1ull << 32)) break;
    }
    const u32 q32 = (u32)qhat;
    // num[j .. j+nd] -= qhat * d
    u32 bin = 0, hi63 = 0, c63 = 0;
    u32 dig[2] = {0, 0};
    for (u32 r = 0; r < 2; ++r) {
      if (r * 64 >= ncol) break;
      const u32 i = r * 64 + lane;
      const u64 pr = (u64)q32 * dv[r];
      const u32 lo = (u32)pr, hi = (u32)(pr >> 32);
      u32 hp = zkw_up(hi, 1);
      if (lane == 0) hp = hi63;
      const u64 u = (u64)lo + hp;
      const u32 sum32 = (u32)u, c1 = (u32)(u >> 32);
      u32 cp = zkw_up(c1, 1);
      if (lane == 0) cp = c63;
      const u32 nv = i < ncol ? num[j + i] : 0u;
      const u32 s32 = i < ncol ? sum32 : 0u, cpp = i < ncol ? cp : 0u;
      const u32 d1 = nv - s32;
      const bool b1 = nv < s32;
      const u32 d2 = d1 - cpp;
      const bool b2 = d1 < cpp;
      const u64 G = zkw_ballot(b1 || b2), P = zkw_ballot(!(b1 || b2) && d2 == 0);
      u32 bout;
      const u64 arr = zkw_lookahead(G, P, bin, bout);
      dig[r] = d2 - (u32)((arr >> lane) & 1);
      bin = bout;
      hi63 = zkw_rl(hi, 63); c63 = zkw_rl(c1, 63);
    }
    u32 qd = q32;
    // negative: add the divisor back (at most once with the refined estimate; the guard bounds it)
    for (u32 guard = 0; bin && guard < 3; ++guard) {
      --qd;
      u32 cin = 0, carried = 0;
      for (u32 r = 0; r < 2; ++r) {
        if (r * 64 >= ncol) break;
        const u32 i = r * 64 + lane;
        const u32 add = i < nd ? dv[r] : 0u;
        const u64 s = (u64)dig[r] + add;
        const u32 d1 = (u32)s;
        const bool gen = (s >> 32) != 0;
        const u64 G = zkw_ballot(gen), P = zkw_ballot(!gen && d1 == 0xffffffffu && i < ncol);
        u32 cout;
        const u64 arr = zkw_lookahead(G, P, cin, cout);
        dig[r] = d1 + (u32)((arr >> lane) & 1);
        cin = cout;
        if (((ncol - 1) >> 6) == r) {       // carry out of the top column: the value crossed zero
          const u32 last = (ncol - 1) & 63;
          carried = last == 63 ? cout : (u32)((arr >> (last + 1)) & 1);
        }
      }
      if (carried) bin = 0;
    }
    if (lane < ncol) num[j + lane] = dig[0];
    if (64 + lane < ncol) num[j + 64 + lane] = dig[1];
    if (lane == 0 && j < ZK_BIG) S.mu[j] = qd;
    ZK_SYNC();
  }
}

// (q3, r) = divmod(a * b, p) via Barrett, see zk_wave_mulmod (zkwg_rsa_core.h)
__device__ inline void zkw_mulmod(ZkRsaLds& S) {
  const u32 L = S.L;
  const u32 lane = ZK_LANE();
  zkw_mul(S.x, S.a, 65, S.b, 65);                        // x = a*b (130 limbs)
  zk_wave_shr(S.q1, 66, S.x, 130, L - 1);                // q1 = x >> (L-1)
  zkw_mul(S.q2, S.q1, 66, S.mu, 66);                     // q2 = q1 * mu
  zk_wave_shr(S.q3, 66, S.q2, 132, L + 1);               // q3 = q2 >> (L+1)
  zkw_mul(S.t, S.q3, 66, S.p, 65);                       // t = q3 * p
  zkw_sub(S.r, S.x, S.t, 67, false);                     // r = x - t (low 67 limbs)
  if (lane == 0) { S.p[65] = 0; S.p[66] = 0; }
  ZK_SYNC();
  u32 it = 0;
  for (;;) {
    const u32 bw = zkw_sub(S.r, S.r, S.p, 67, true);     // if (r >= p) r -= p
    if (bw) break;
    if (++it > 3) { if (lane == 0) S.ok = 0; break; }
    zkw_inc(S.q3, 66, 1);
  }
  ZK_SYNC();
}

// BigLessThan(121,17)(a, b): same image values as zk_blt_emit (zkwg_rsa_core.h); gates from ballots
__device__ inline u32 zkw_blt_emit(const u64 (*a)[2], const u64 (*b)[2], const ZkBltLayout& Lb,
                                   u64* bits, u32* small, Fr* frv) {
  const u32 i = ZK_LANE();
  bool lt = false, eq = false;
  if (i < ZK_RSA_K) {
    u64 c = 0, bw = 0;
    const u64 lo = zk_adc(a[i][0], 0, c);
    const u64 hi = a[i][1] + (1ull << 57) + c;
    const u64 dlo = zk_sbb(lo, b[i][0], bw);
    const u64 dhi = hi - b[i][1] - bw;
    bits[Lb.b_lt + 2 * i] = dlo;
    bits[Lb.b_lt + 2 * i + 1] = dhi;
    lt = ((dhi >> 57) & 1) == 0;
    u256s d;
    bw = 0;
    d.l[0] = zk_sbb(b[i][0], a[i][0], bw);
    d.l[1] = zk_sbb(b[i][1], a[i][1], bw);
    d.l[2] = zk_sbb(0, 0, bw);
    d.l[3] = zk_sbb(0, 0, bw);
    eq = u256_is_zero(d);
    frv[Lb.f_eq + 2 * i] = fr_from_u64(eq ? 1 : 0);
    frv[Lb.f_eq + 2 * i + 1] = fr_from_signed(d);
  }
  const u32 LT = (u32)zkw_ballot(lt) & 0x1ffffu, EQ = (u32)zkw_ballot(eq) & 0x1ffffu;
  // eq_ands[i] = eq[16] & .. & eq[i];  ands[i] = eq_ands[i+1] & lt[i] (eq_ands[16] := eq[16]);
  // ors[i] = lt[16] | ands[15] | .. | ands[i]
  const u32 NE = ~EQ & 0x1ffffu;
  const bool eqa = i < 16 && (NE >> i) == 0;
  const bool eqa_above = i < 16 && (NE >> (i + 1)) == 0;
  const bool ands = eqa_above && ((LT >> i) & 1);
  const u32 ANDS = (u32)zkw_ballot(ands) & 0xffffu;
  const bool ors = ((LT >> 16) & 1) || (i < 16 && (ANDS >> i) != 0);
  if (i < 16) {
    small[Lb.m_gates + i] = ors ? 1u : 0u;
    small[Lb.m_gates + 16 + i] = ands ? 1u : 0u;
    small[Lb.m_gates + 32 + i] = eqa ? 1u : 0u;
  }
  return (((LT >> 16) & 1) || ANDS != 0) ? 1u : 0u;   // ors[0]
}

// Horner value sum_i limb[i] x^i (x < 33, limbs < 2^121): < 2^206
__device__ __forceinline__ u256s zkw_horner(const u64 (*l)[2], u32 x) {
  u256s v = u256_zero();
#pragma unroll 1
  for (int i = ZK_RSA_K - 1; i >= 0; --i) v = u256_mul_small_add(v, x, l[i]);
  return v;
}
__device__ __forceinline__ Fr zkw_fr(const u256s& v) { return Fr{{v.l[0], v.l[1], v.l[2], v.l[3]}}; }

// All signals of one FpMul(121,17); vA, vB, vP = A(x), B(x), P(x) of this lane's evaluation point,
// returns R(x) in vR (the next FpMul's operand).  q, r as 121-bit limbs in S.q121 / S.r121.
__device__ inline void zkw_fpmul_emit(ZkRsaLds& S, const ZkFpMulLayout& F, u64* bits, u32* small, Fr* frv,
                                      const u256s& vA, const u256s& vB, const u256s& vP, u256s& vR, bool& bad) {
  const u32 lane = ZK_LANE();
  if (lane < 2 * ZK_RSA_K - 1) {
    const u256s vQ = zkw_horner(S.q121, lane);
    vR = zkw_horner(S.r121, lane);
    frv[F.f_main + lane] = fr_mul_std(zkw_fr(vA), zkw_fr(vB));
    frv[F.f_main + 67 + lane] = fr_add(fr_mul_std(zkw_fr(vP), zkw_fr(vQ)), zkw_fr(vR));
  }
  if (lane < ZK_RSA_K) {
    const u32 i = lane;
    frv[F.f_main + 33 + i] = Fr{{S.q121[i][0], S.q121[i][1], 0, 0}};
    frv[F.f_main + 50 + i] = Fr{{S.r121[i][0], S.r121[i][1], 0, 0}};
    bits[F.b_qr + 2 * i] = S.q121[i][0];
    bits[F.b_qr + 2 * i + 1] = S.q121[i][1];
    bits[F.b_qr + 34 + 2 * i] = S.r121[i][0];
    bits[F.b_qr + 34 + 2 * i + 1] = S.r121[i][1];
  }
  // column sums: work item w < 33: sum_j a[j] b[w-j]; w >= 33: sum_j p[j] q[w-33-j]   (66 items)
  for (u32 w = lane; w < 2 * (2 * ZK_RSA_K - 1); w += 64) {
    const bool pq = w >= 2 * ZK_RSA_K - 1;
    const int i = (int)(pq ? w - (2 * ZK_RSA_K - 1) : w);
    const u64 (*xa)[2] = pq ? S.p121 : S.a121;
    const u64 (*xb)[2] = pq ? S.q121 : S.b121;
    int j0 = i - (ZK_RSA_K - 1);
    if (j0 < 0) j0 = 0;
    const int j1 = i < ZK_RSA_K - 1 ? i : ZK_RSA_K - 1;
    u256s acc = u256_zero();
    for (int j = j0; j <= j1; ++j) acc = u256_add(acc, u256_mul128(xa[j], xb[i - j]));
    if (pq) S.tq[i] = acc; else S.tt[i] = acc;
  }
  ZK_SYNC();
  // t[i] = ab[i] - pq[i] - r[i];  th = t >> 121 (arithmetic), tl = t mod 2^121
  const u64 m57 = (1ull << 57) - 1;
  u256s th = u256_zero();
  u64 tl0 = 0, tl1 = 0;
  u256s t = u256_zero();
  if (lane < 2 * ZK_RSA_K - 1) {
    t = u256_sub(S.tt[lane], S.tq[lane]);
    if (lane < ZK_RSA_K) t = u256_sub(t, u256s{{S.r121[lane][0], S.r121[lane][1], 0, 0}});
    th = u256_sar121(t);
    tl0 = t.l[0]; tl1 = t.l[1] & m57;
  }
  // w[i] = tl[i] + th[i-1]  ->  wl = w mod 2^121, wh = w >> 121 (fits 64 bits: |th| < 2^136)
  u256s thp;
  for (int k = 0; k < 4; ++k) {
    const u32 lo = zkw_up((u32)th.l[k], 1), hi = zkw_up((u32)(th.l[k] >> 32), 1);
    thp.l[k] = (u64)lo | ((u64)hi << 32);
  }
  if (lane == 0) thp = u256_zero();
  const u256s wv = u256_add(u256s{{tl0, tl1, 0, 0}}, thp);
  const u64 wl0 = wv.l[0], wl1 = wv.l[1] & m57;
  const long long wh = (long long)u256_sar121(wv).l[0];
  // chain on scalars: u[i] = wh[i] + floor((wl[i] + u[i-1]) / 2^121); every step must divide exactly
  long long u = 0, my_u = 0, my_up = 0;
  bool inexact = false;
#pragma unroll
  for (int i = 0; i < 2 * ZK_RSA_K - 1; ++i) {
    const u64 l0 = zkw_rl64(wl0, i), l1 = zkw_rl64(wl1, i);
    const long long whi = (long long)zkw_rl64((u64)wh, i);
    if ((int)lane == i) my_up = u;
    // v = wl + u (signed 128-bit)
    u64 c = 0;
    const u64 v0 = zk_adc(l0, (u64)u, c);
    const u64 v1 = l1 + (u < 0 ? ~0ull : 0ull) + c;
    if (i < 2 * ZK_RSA_K - 2 && (v0 != 0 || (v1 & m57) != 0)) inexact = true;   // in + carry === carry * 2^121
    u = whi + ((long long)v1 >> 57);
    if ((int)lane == i) my_u = u;
  }
  if (inexact) bad = true;
  if (lane < 2 * ZK_RSA_K - 2) {
    // carry[i] = th[i] + u[i]
    const u256s uu{{(u64)my_u, my_u < 0 ? ~0ull : 0ull, my_u < 0 ? ~0ull : 0ull, my_u < 0 ? ~0ull : 0ull}};
    const u256s c = u256_add(th, uu);
    frv[F.f_carry + lane] = fr_from_signed(c);
    const u256s rc = u256_add(c, u256s{{0, 0, 1ull << 2, 0}});     // carryRangeChecks[i].in = carry + 2^130
    if (u256_is_neg(rc) || rc.l[3] != 0 || (rc.l[2] >> 3) != 0) bad = true;
    bits[F.b_carry + 3 * lane] = rc.l[0];
    bits[F.b_carry + 3 * lane + 1] = rc.l[1];
    bits[F.b_carry + 3 * lane + 2] = rc.l[2];
  }
  if (lane == 2 * ZK_RSA_K - 2) {
    // in[k-1] + carry[k-2] === 0:  t[32] + th[31] + u[31]
    const u256s up{{(u64)my_up, my_up < 0 ? ~0ull : 0ull, my_up < 0 ? ~0ull : 0ull, my_up < 0 ? ~0ull : 0ull}};
    if (!u256_is_zero(u256_add(u256_add(t, thp), up))) bad = true;
    frv[F.f_carry + 2 * ZK_RSA_K - 2] = fr_zero();                 // carry[k-1] is never assigned
  }
  const u32 lt = zkw_blt_emit(S.r121, S.p121, F.blt, bits, small, frv);   // r_p_lt_check.out === 1
  if (!lt) bad = true;
}

// The 18 x 17 IsEqual differences of one email, inverted in place: each lane batches its 4..5 values
// with Montgomery's trick.  Products are taken with fr_mont_mul on STANDARD-form values; the stray
// R^-1 factors cancel between the prefix products and the back-substitution, so no
// conversion to or from Montgomery form is needed.  Zero differences keep inv = 0.
__device__ inline void zkw_invert_all(const ZkRsaLayout& R, Fr* frv) {
  const u32 total = 18 * ZK_RSA_K;
  const u32 lane = ZK_LANE();
  auto slot_of = [&](u32 e) -> u32 {
    const u32 blk = e / ZK_RSA_K, i = e - blk * ZK_RSA_K;
    const ZkBltLayout& Lb = blk == 0 ? R.blt : R.mul[blk - 1].blt;
    return Lb.f_eq + 2 * i + 1;
  };
  Fr val[5], pre[5];
  u32 cnt = 0;
  // A_1 = v_1, A_j = mont(A_{j-1}, v_j) = v_1 .. v_j R^-(j-1)   (zeros are skipped)
  Fr acc = fr_zero();
  bool have = false;
#pragma unroll
  for (u32 k = 0; k < 5; ++k) {
    const u32 e = lane + 64 * k;
    val[k] = fr_zero();
    pre[k] = fr_zero();
    if (e < total) {
      const Fr v = frv[slot_of(e)];
      val[k] = v;
      if (!fr_is_zero(v)) {
        pre[k] = acc;                 // product of the earlier non-zero values (meaningless if !have)
        acc = have ? fr_mont_mul(acc, v) : v;
        // remember whether this value was the first of the chain
        if (!have) pre[k] = fr_zero();
        have = true;
      }
      cnt = k + 1;
    }
  }
  // y = A_n = prod v * R^-(n-1);  y^-1 = prod v^-1 * R^(n-1)
  Fr inv = have ? fr_inv_by(acc) : fr_zero();   // safegcd: no data-dependent control flow (zkwg_fr_inv.h)
  bool last_done = false;   // walking back: the first non-zero value met from the top is the chain's last
#pragma unroll
  for (int k = 4; k >= 0; --k) {
    const u32 e = lane + 64 * (u32)k;
    if ((u32)k < cnt && e < total && !fr_is_zero(val[k])) {
      const bool first = fr_is_zero(pre[k]);
      // out_j = mont(I_j, A_{j-1}) = v_j^-1 (standard form);  I_{j-1} = mont(I_j, v_j)
      frv[slot_of(e)] = first ? inv : fr_mont_mul(inv, pre[k]);
      if (!first) inv = fr_mont_mul(inv, val[k]);
      last_done = true;
    }
  }
  (void)last_done;
  ZK_SYNC();
}

// The whole RSAVerifier65537(121,17) of one email (device path).
__device__ inline void zkw_rsa_email(ZkRsaLds& S, const ZkRsaLayout& R, const u8* rec, const u32* digest,
                                     u64* bits, u32* small, Fr* frv) {
  const u64 top_mask = (1ull << 57) - 1;
  const u32 lane = ZK_LANE();
  bool bad = false;
  if (lane == 0) S.ok = 1;
  if (lane < ZK_RSA_K) {
    const u32 i = lane;
    const u64* pm = (const u64*)(rec + R.in_mod + 16 * i);
    const u64* ps = (const u64*)(rec + R.in_sig + 16 * i);
    S.p121[i][0] = pm[0]; S.p121[i][1] = pm[1] & top_mask;
    S.s121[i][0] = ps[0]; S.s121[i][1] = ps[1] & top_mask;
    bad = (pm[1] >> 57) != 0 || (ps[1] >> 57) != 0;   // Num2Bits(121) range checks
    u64 m0 = 0, m1 = 0;
    if (digest) {
      u32 w32[8];
      for (int j = 0; j < 8; ++j) w32[j] = digest[7 - j];
      u64 tmp[2];
      zk_limb121_from_32(w32, 8, i, tmp);
      m0 = tmp[0]; m1 = tmp[1];
    } else {
      const u64* pmsg = (const u64*)(rec + R.in_msg + 16 * i);
      m0 = pmsg[0]; m1 = pmsg[1] & top_mask;
      bad = bad || (pmsg[1] >> 57) != 0;
    }
    S.m121[i][0] = m0; S.m121[i][1] = m1;
    // RSAPad (lib/rsa.circom:101-181): messageBits[256..] === 0
    if (i == 2 && ((m0 >> 14) != 0 || m1 != 0)) bad = true;
    if (i > 2 && (m0 | m1) != 0) bad = true;
    bits[R.b_modbits + 2 * i] = S.p121[i][0]; bits[R.b_modbits + 2 * i + 1] = S.p121[i][1];
    bits[R.b_msgbits + 2 * i] = m0;           bits[R.b_msgbits + 2 * i + 1] = m1;
    bits[R.b_sigbits + 2 * i] = S.s121[i][0]; bits[R.b_sigbits + 2 * i + 1] = S.s121[i][1];
  }
  ZK_SYNC();
  for (u32 w = lane; w < ZK_BIG; w += 64) {
    S.p[w] = w < 65 ? zk_bits_from_121(S.p121, 32 * w) : 0;
    S.base[w] = w < 65 ? zk_bits_from_121(S.s121, 32 * w) : 0;
    S.a[w] = S.base[w];
    S.b[w] = S.base[w];
  }
  ZK_SYNC();
  const u32 L = zkw_bitlen(S.p, 65);
  if (lane == 0) S.L = L;
  // paddedMessageBits[416..480] === 1  <=>  modulus has a bit set at position >= 488
  if (L < 489) bad = true;
  // modulusZero[idx].in = popcount of modulus bits at positions >= 424 + 8 idx (idx 0..204):
  // suffix popcounts of the limbs (S.q1 as scratch), then one partial limb per idx
  {
    for (u32 w = lane; w < ZK_BIG; w += 64) S.q1[w] = w < 65 ? (u32)__builtin_popcount(S.p[w]) : 0u;
    ZK_SYNC();
    // inclusive suffix sum over 68 entries: lane handles w = lane (and lane 0..3 also 64 + lane)
    u32 hi4 = 0;
    if (lane < 4) hi4 = S.q1[64 + lane];
    // suffix over the 4 top entries first
    const u32 h0 = zkw_rl(hi4, 0), h1 = zkw_rl(hi4, 1), h2 = zkw_rl(hi4, 2), h3 = zkw_rl(hi4, 3);
    u32 v = S.q1[lane];
    for (u32 d = 1; d < 64; d <<= 1) {
      const u32 o = (u32)__shfl_down((int)v, d);
      if (lane + d < 64) v += o;
    }
    ZK_SYNC();
    S.q1[lane] = v + h0 + h1 + h2 + h3;
    if (lane == 0) { S.q1[64] = h0 + h1 + h2 + h3; S.q1[65] = h1 + h2 + h3; S.q1[66] = h2 + h3; S.q1[67] = h3; }
    ZK_SYNC();
    for (u32 idx = lane; idx < 205; idx += 64) {
      const u32 b0 = 424 + 8 * idx, w0 = b0 >> 5;
      small[R.m_modzero + idx] = (u32)__builtin_popcount(S.p[w0] >> (b0 & 31)) + (w0 + 1 < ZK_BIG ? S.q1[w0 + 1] : 0u);
    }
    ZK_SYNC();
  }
  const u32 sig_lt = zkw_blt_emit(S.s121, S.p121, R.blt, bits, small, frv);   // bigLessThan.out === 1
  if (!sig_lt) bad = true;
  const bool chain = L >= 2;
  if (chain) zkw_barrett_mu(S);
  // Horner values of the modulus and the signature at this lane's evaluation point
  u256s vP = u256_zero(), vS = u256_zero(), vR = u256_zero();
  if (lane < 2 * ZK_RSA_K - 1) { vP = zkw_horner(S.p121, lane); vS = zkw_horner(S.s121, lane); }
  for (u32 m = 0; m < 17; ++m) {
    // doublers[m]: a = b = previous result; adder (m == 16): a = base, b = doublers[15].out
    if (lane < ZK_RSA_K) {
      const u32 i = lane;
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
    if (m > 0) {
      for (u32 w = lane; w < ZK_BIG; w += 64) {
        const u32 rv = w < 66 ? S.r[w] : 0;
        S.b[w] = rv;
        S.a[w] = (m < 16) ? rv : S.base[w];
      }
    }
    ZK_SYNC();
    if (chain) {
      zkw_mulmod(S);
    } else {
      for (u32 w = lane; w < ZK_BIG; w += 64) { S.q3[w] = 0; S.r[w] = 0; }
      ZK_SYNC();
    }
    // q must fit 17 x 121 = 2057 bits
    if (lane == 0 && ((S.q3[64] >> 9) != 0 || S.q3[65] != 0)) bad = true;
    if (lane < ZK_RSA_K) {
      zk_limb121_from_32(S.q3, 66, lane, S.q121[lane]);
      zk_limb121_from_32(S.r, 66, lane, S.r121[lane]);
    }
    ZK_SYNC();
    const u256s vA = (m == 0 || m == 16) ? vS : vR;
    const u256s vB = m == 0 ? vS : vR;
    u256s vRn = vR;
    zkw_fpmul_emit(S, R.mul[m], bits, small, frv, vA, vB, vP, vRn, bad);
    vR = vRn;
    ZK_SYNC();
  }
  if (R.present != 2) zkw_invert_all(R, frv);   // present == 2: profiling knob ZKWG_DEBUG_SKIP_INV (witness then wrong)
  // bigPow.out[i] === padder.out[i] (lib/rsa.circom:43-45): expected EMSA-PKCS1-v1_5 value, limb per lane.
  // ones run: bit i (>= 416) is 1 iff m8(i) + 8 <= hb (m8 = i rounded up to a multiple of 8, hb = highest
  // set modulus bit)  <=>  i <= 8 * floor((hb - 8) / 8)
  {
    const u32 hb = L ? L - 1 : 0;
    const long long last = (L && hb >= 8) ? (long long)(8 * ((hb - 8) / 8)) : -1;   // inclusive
    for (u32 w = lane; w < 66; w += 64) {
      u32 pv = w < 8 ? zk_bits_from_121(S.m121, 32 * w) : 0u;
      const u32 di[5] = {0x05000420u, 0x03040201u, 0x86480165u, 0x0d060960u, 0x00303130u};
      if (w >= 8 && w < 13) pv |= di[w - 8];
      const long long lastc = last < ZK_RSA_K * ZK_RSA_N - 1 ? last : ZK_RSA_K * ZK_RSA_N - 1;
      const long long lo = 32ll * w > 416 ? 32ll * w : 416, hi = 32ll * w + 31 < lastc ? 32ll * w + 31 : lastc;
      if (hi >= lo) {
        const u32 a0 = (u32)(lo - 32ll * w), a1 = (u32)(hi - 32ll * w);
        const u32 mask = (a1 == 31 ? 0xffffffffu : ((1u << (a1 + 1)) - 1u)) & ~((1u << a0) - 1u);
        pv |= mask;
      }
      if (w < 65 ? pv != S.r[w] : S.r[65] != 0) bad = true;
    }
  }
  if (zkw_ballot(bad) != 0 && lane == 0) S.ok = 0;
  ZK_SYNC();
}

#endif  // __HIPCC__ || ZKWG_WAVESIM
