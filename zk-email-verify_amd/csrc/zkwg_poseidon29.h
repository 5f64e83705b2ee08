// Poseidon(t-1) with sparse partial rounds (zkwg_poseidon_sparse.h: same rounds, same kept signals) in 9 x 29-bit LIMB FORM from the
// first addition to the last product -- the permutation zk_rslb_chunks runs 786 k times per batch of 4,096 emails
// (packages/circuits/utils/hash.circom:49-82, helpers/remove-soft-line-breaks.circom:14-126).
//
// Why a second evaluator: tools/mulbench.hip measured that v_mad_u64_u32 issues at nearly the plain VALU rate on gfx950, so what a
// field product costs is the carry handling around its multiply-adds.  60 % of this permutation's work is the 3,757 lazily reduced
// multiply-accumulates of its dense mixes and first-row dot products: with 32-bit limbs each costs ~264 instructions (64 multiply-adds,
// a 64-bit add and a carry per step); with 29-bit limbs the 81 partial products of a multiply-accumulate are 81 v_mad_u64_u32 chained
// through 17 64-bit COLUMN sums and nothing else -- columns take 63 products of 58 bits before they overflow, so carries are
// propagated once per 7 multiply-accumulates, and the Montgomery reduction (by 2^261) runs once per dot product.  The S-box and the
// column updates are the Comba product of zkwg_comba29.h without its split / pack / conditional subtraction.
//
// Ranges.  A product returns  a b / 2^261 + r  at most (limbs normalised: < 2^29, the top one holds the rest), whatever multiples of r
// its operands carry: (2^261 / r = 169).  Nothing is reduced below that on the way: the partial rounds' u_j grow by < 2.01 r per round
// (68 rounds: < 138 r < 2^261), their dot products stay below 2,214 r^2 -> u_0 < 15 r, S-box inputs < 16.2 r -> every EMITTED signal
// (x^2, x^4, x^5) and the digest are product outputs below 1.11 r: one conditional subtraction when they are packed into the image.
//
// Table (u32, built by zk_build_poseidon29 from the Fr table of zk_build_poseidon_sparse): additive constants as they are,
// multiplicative ones times 2^261 (the Fr table holds them times 2^256):
//   cF[4][T][9] | M[T (row i)][T (j)][9] | P[rp + 1][T][27] = per round k and element j: c'_k[j], (n00 | v_j), (0 | w^_j); row rp is
//   all zeros (round k adds round k + 1's constants) | B[T (i)][T (j)][9] | cL[4][T][9]
// The dense matrices are stored by OUTPUT row, so the 153 limbs one output needs are one contiguous scalar load stream; all table
// indices are wavefront-uniform (scalar loads, the multiply-adds take the limb from an SGPR).
#pragma once
#include "zkwg_poseidon_sparse.h"

#define ZK_P29_M 0x1fffffffu
#define ZK_P29_N0 0x0fffffffu      // -r^-1 mod 2^29
ZK_HD u32 zk_p29_tab_size(u32 t, u32 rp) { return 9u * (4 * t + t * t + 3 * (rp + 1) * t + t * t + 4 * t); }

// table words are read at wavefront-uniform addresses and never written while a kernel runs: on the device they are addressed in the
// CONSTANT address space, which is what lets the compiler fetch them with scalar loads (a limb is then an SGPR operand of
// v_mad_u64_u32) -- through a plain pointer it must assume the kernel's own stores may alias and uses vector loads: 153 VGPRs of
// constants per dense-mix output
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) u32* ZkTab29;
#define ZK_TAB29(p) ((ZkTab29)(unsigned long long)(p))
#else
typedef const u32* ZkTab29;
#define ZK_TAB29(p) (p)
#endif

// Left alone the compiler hoists every scalar load of a loop body to its top (153 SGPRs for one dense-mix output: spilled to VGPR
// lanes, then VGPRs to scratch).  ZK_P29_AFTER(p, x) emits nothing, but makes the pointer p look computed from x, so loads through p
// cannot move above the instructions that produce x.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_P29_AFTER(p, x) asm volatile("" : "+s"(p) : "v"((u32)(x)))
// ZK_P29_ARRIVED(a, b) emits nothing either, but uses the 9 + 9 limbs: the compiler puts the wait for their loads HERE; with the
// scheduling barrier behind it the next operands' loads are issued after that wait and before the 81 multiply-adds that hide them
#define ZK_P29_ARRIVED(a, b)                                                                                                          \
  asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "s"(b[0]), \
               "s"(b[1]), "s"(b[2]), "s"(b[3]), "s"(b[4]), "s"(b[5]), "s"(b[6]), "s"(b[7]), "s"(b[8]))
#define ZK_P29_BAR() __builtin_amdgcn_sched_barrier(0)
#else
#define ZK_P29_AFTER(p, x) ((void)0)
#define ZK_P29_ARRIVED(a, b) ((void)0)
#define ZK_P29_BAR() ((void)0)
#endif

// ZKWG_P29_CHECK (host builds of the tests only): every column addition is checked for wrap-around and every operand limb for its
// declared width; zk_p29_violations counts what the range argument above says cannot happen (tests/test_ev_cpu.py asserts 0)
#if defined(ZKWG_P29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
static unsigned long long zk_p29_violations = 0;
#define ZK_P29_EXPECT(cond) do { if (!(cond)) ++zk_p29_violations; } while (0)
#else
#define ZK_P29_EXPECT(cond) ((void)0)
#endif

struct ZkW29 { u64 c[17]; };
ZK_HD void zk_w29_zero(ZkW29& w) {
#pragma unroll
  for (int i = 0; i < 17; ++i) w.c[i] = 0;
}
// w += a * b, column-wise (a, b: limbs < 2^29); at most 7 calls between two zk_w29_carry
template <class BP>
ZK_HD void zk_w29_mac(ZkW29& w, const u32 (&a)[9], BP b) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    ZK_P29_EXPECT(a[i] <= ZK_P29_M);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      ZK_P29_EXPECT(b[k] <= ZK_P29_M && w.c[i + k] + (u64)a[i] * b[k] >= w.c[i + k]);
      w.c[i + k] += (u64)a[i] * b[k];
    }
  }
}
ZK_HD void zk_w29_carry(ZkW29& w) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    w.c[k + 1] += w.c[k] >> 29;
    w.c[k] &= ZK_P29_M;
  }
}
// out = w / 2^261 mod r (+ at most r); w normalised by zk_w29_carry
ZK_HD void zk_w29_redc(const ZkW29& w, u32 (&out)[9]) {
  const u32 P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  u32 q[9];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    acc += w.c[k];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (u64)q[i] * P[k - i];
    q[k] = ((u32)acc * ZK_P29_N0) & ZK_P29_M;
    acc += (u64)q[k] * P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
    acc += w.c[k];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)q[i] * P[k - i];
    out[k - 9] = (u32)acc & ZK_P29_M;
    acc >>= 29;
  }
  out[8] = (u32)acc;
  ZK_P29_EXPECT((acc >> 29) == 0);           // the reduced dot product stays below 2^261
}
// r = a b / 2^261 mod r (+ at most r); a: limbs < 2^29 (top < 2^32), b: limbs < 2^29
template <class BP>
ZK_HD void zk_l29_mul(u32 (&r)[9], const u32 (&a)[9], BP b) {
  const u32 P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  u32 q[9], o[9];
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { ZK_P29_EXPECT(i == 8 || a[i] <= ZK_P29_M); ZK_P29_EXPECT(b[i] <= ZK_P29_M); }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (u64)a[i] * b[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (u64)q[i] * P[k - i];
    q[k] = ((u32)acc * ZK_P29_N0) & ZK_P29_M;
    acc += (u64)q[k] * P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)a[i] * b[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (u64)q[i] * P[k - i];
    o[k - 9] = (u32)acc & ZK_P29_M;
    acc >>= 29;
  }
  o[8] = (u32)acc;
  ZK_P29_EXPECT((acc >> 29) == 0);           // the result stays below 2^261
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = o[i];
}
// the same product with the multiply-adds of a row independent of each other (operand scanning into 17 column sums, then a row-wise
// reduction): ~290 instructions instead of ~215, but 9 multiply-adds in flight instead of one dependent chain -- for a kernel that
// runs one wavefront per SIMD
template <class BP>
ZK_HD void zk_l29_mul_rows(u32 (&r)[9], const u32 (&a)[9], BP b) {
  const u32 P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  u64 c[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) c[i] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
#pragma unroll
    for (int k = 0; k < 9; ++k) c[i + k] += (u64)a[i] * b[k];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const u32 m = ((u32)c[i] * ZK_P29_N0) & ZK_P29_M;
#pragma unroll
    for (int k = 0; k < 9; ++k) c[i + k] += (u64)m * P[k];
    c[i + 1] += c[i] >> 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
    r[k - 9] = (u32)c[k] & ZK_P29_M;
    c[k + 1] += c[k] >> 29;
  }
  r[8] = (u32)c[17];
}
template <int V, class BP>
ZK_HD void zk_l29_mulv(u32 (&r)[9], const u32 (&a)[9], BP b) {
  if (V & 1) zk_l29_mul_rows(r, a, b);
  else zk_l29_mul(r, a, b);
}
// r = a + b, normalised (the top limb takes what is left)
template <class BP>
ZK_HD void zk_l29_add(u32 (&r)[9], const u32 (&a)[9], BP b) {
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const u32 t = a[i] + b[i] + c;
    r[i] = i < 8 ? (t & ZK_P29_M) : t;
    c = t >> 29;
  }
}
// r = a + b + c, normalised
template <class BP, class CP>
ZK_HD void zk_l29_add3(u32 (&r)[9], const u32 (&a)[9], BP b, CP c) {
  u32 cy = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const u32 t = a[i] + b[i] + c[i] + cy;
    r[i] = i < 8 ? (t & ZK_P29_M) : t;
    cy = t >> 29;
  }
}
// value < 2 r -> the canonical 4 x 64-bit words
ZK_HD Fr zk_l29_to_fr(const u32 (&x)[9]) {
  u64 w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, k = bit >> 6, s = bit & 63;
    w[k] |= (u64)x[i] << s;
    if (s > 64 - 29 && k + 1 < 4) w[k + 1] |= (u64)x[i] >> (64 - s);
  }
  Fr v{{w[0], w[1], w[2], w[3]}};
  ZK_P29_EXPECT(x[8] < (1u << 24));          // below 2^256: the packing loses nothing
  if (fr_geq(v, fr_p())) {
    u64 bw;
    v = fr_sub_raw(v, fr_p(), bw);
  }
  ZK_P29_EXPECT(!fr_geq(v, fr_p()));         // one subtraction was enough
  return v;
}
ZK_HD void zk_l29_from_fr(const Fr& x, u32 (&l)[9]) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, k = bit >> 6, s = bit & 63;
    u64 v = x.l[k] >> s;
    if (s > 64 - 29 && k + 1 < 4) v |= x.l[k + 1] << (64 - s);
    l[i] = (u32)v & ZK_P29_M;
  }
}

// x -> x^5; emits (x^5, x^2, x^4) canonical.  2^522 mod r = the factor that puts a value into 2^261-Montgomery form.
template <int V>
ZK_HD void zk_p29_sbox(u32 (&x)[9], Fr* emit) {
  const u32 RR[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
  u32 xm[9], in2[9], t[9], in4[9];
  zk_l29_mulv<V>(xm, x, RR);
  zk_l29_mulv<V>(in2, xm, x);
  zk_l29_mulv<V>(t, in2, RR);
  zk_l29_mulv<V>(in4, t, in2);
  zk_l29_mulv<V>(x, xm, in4);
  emit[0] = zk_l29_to_fr(x);
  emit[1] = zk_l29_to_fr(in2);
  emit[2] = zk_l29_to_fr(in4);
}

// st[i] = sum_j y_j mat[i][j] for all i.  State element j, limb l at st[l * ls + j * js].  The old state is read into registers first
// (static indices: the j loop is unrolled, the i loop is not), so the result overwrites the state in place.
// the first n_out outputs only (the permutation's last mix feeds nothing but element 0)
template <int T>
ZK_HD void zk_p29_dense(u32* st, const u32 js, const u32 ls, ZkTab29 mat, const u32 n_out) {
  u32 y[T][9];
#pragma unroll T
  for (int j = 0; j < T; ++j) {
#pragma unroll
    for (int l = 0; l < 9; ++l) y[j][l] = st[l * ls + j * js];
  }
#pragma nounroll
  for (u32 i = 0; i < n_out; ++i) {
    ZkTab29 row = mat + (size_t)i * T * 9;
    ZkW29 w;
    zk_w29_zero(w);
#pragma unroll T     // an explicit count: a bare `unroll` falls back to a partial unroll above 16 k instructions, and y[] lands in scratch
    for (int j = 0; j < T; ++j) {
      ZkTab29 bj = row + j * 9;
      ZK_P29_AFTER(bj, w.c[0]);          // the row's 153 limbs are fetched 9 at a time, one multiply-accumulate ahead -- not all up front
      zk_w29_mac(w, y[j], bj);
      if (j % 7 == 6) zk_w29_carry(w);
    }
    zk_w29_carry(w);
    u32 o[9];
    zk_w29_redc(w, o);
#pragma unroll
    for (int l = 0; l < 9; ++l) st[l * ls + i * js] = o[l];
  }
}

// The same mix with NOTHING held in registers across outputs: 446 of a SIMD's 512 registers per lane is what the version above costs
// (153 of them the old state), and zk_expand -- 64 registers -- then shares the SIMD with one wavefront instead of six while this
// kernel runs beside it.  Here every multiply-accumulate reads its state element from LDS and its row limbs by
// scalar loads, one step ahead (two operand sets, A and B); the outputs cannot overwrite the state they are computed from, so they
// wait in `stage` (global memory, word k of this lane at stage[k * ss]: a wavefront's lanes are adjacent) and are copied back at the end.
template <int T>
ZK_HD void zk_p29_dense_lean(u32* st, const u32 js, const u32 ls, ZkTab29 mat, const u32 n_out, u32* stage, const size_t ss) {
  u32 i = 0;
#pragma nounroll
  for (; i < n_out; ++i) {
    ZkTab29 row = mat + (size_t)i * T * 9;
    ZkW29 w;
    zk_w29_zero(w);
    u32 aA[9], bA[9], aB[9], bB[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) { aA[l] = st[l * ls]; bA[l] = row[l]; }
    u32 j = 0;
#pragma nounroll
    for (; j + 1 < (u32)T; j += 2) {
      ZK_P29_ARRIVED(aA, bA);
      ZK_P29_BAR();
#pragma unroll
      for (int l = 0; l < 9; ++l) { aB[l] = st[l * ls + (j + 1) * js]; bB[l] = row[(j + 1) * 9 + l]; }
      ZK_P29_BAR();
      zk_w29_mac(w, aA, bA);
      ZK_P29_ARRIVED(aB, bB);
      ZK_P29_BAR();
      const u32 jn = j + 2 < (u32)T ? j + 2 : j;
#pragma unroll
      for (int l = 0; l < 9; ++l) { aA[l] = st[l * ls + jn * js]; bA[l] = row[jn * 9 + l]; }
      ZK_P29_BAR();
      zk_w29_mac(w, aB, bB);
      if (j % 6u == 4u) zk_w29_carry(w);      // after 6 multiply-accumulates
    }
    if (T & 1) zk_w29_mac(w, aA, bA);          // element T - 1, fetched by the last pair
    zk_w29_carry(w);
    u32 o[9];
    zk_w29_redc(w, o);
#pragma unroll
    for (int l = 0; l < 9; ++l) stage[(size_t)(i * 9 + l) * ss] = o[l];
  }
#pragma nounroll
  for (i = 0; i < n_out; ++i) {
    u32 o[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) o[l] = stage[(size_t)(i * 9 + l) * ss];
#pragma unroll
    for (int l = 0; l < 9; ++l) st[l * ls + i * js] = o[l];
  }
}
template <int T, int V>
ZK_HD void zk_p29_densev(u32* st, const u32 js, const u32 ls, ZkTab29 mat, const u32 n_out, u32* stage, const size_t ss) {
  if (V & 4) zk_p29_dense_lean<T>(st, js, ls, mat, n_out, stage, ss);
  else zk_p29_dense<T>(st, js, ls, mat, n_out);
}

// One permutation.  st: T state elements in limb form (element j, limb l at st[l * ls + j * js]; values < 2^29 on entry, element 0 =
// capacity); emit: 3 * (8T + rp) Fr, Sigma signals in component order (sigmaF[8][T], sigmaP[rp]).  Returns out[0] (canonical).
// V: bit 0 = row-wise products (zk_l29_mul_rows), bit 1 = two state elements per loop iteration (two independent chains of products
// for the scheduler to interleave), bit 2 = dense mixes through `stage` (zk_p29_dense_lean; stage = 153 words with stride ss, unused
// otherwise).  Same values either way; which is faster is a measurement (zk_rslb_chunks).
template <int T, int V = 0>
ZK_HD Fr zk_poseidon29(u32* st, const u32 js, const u32 ls, const u32* tab_, const u32 rp, Fr* emit, u32* stage = nullptr, const size_t ss = 0) {
  ZkTab29 cF = ZK_TAB29(tab_);
  ZkTab29 M = cF + 9 * 4 * T;
  ZkTab29 P = M + 9 * T * T;
  ZkTab29 B = P + 27 * (rp + 1) * T;
  ZkTab29 cL = B + 9 * T * T;
  for (u32 half = 0; half < 2; ++half) {
    if (half == 1) {
      u32 u0[9];
#pragma unroll
      for (int l = 0; l < 9; ++l) u0[l] = st[l * ls];
      // the state elements 1 .. T-1 are kept WITH the coming round's constant added (round k adds round k + 1's: one three-way
      // addition per element and round instead of two additions)
#pragma nounroll
      for (u32 j = 1; j < (u32)T; ++j) {
        u32 uj[9];
#pragma unroll
        for (int l = 0; l < 9; ++l) uj[l] = st[l * ls + j * js];
        zk_l29_add(uj, uj, P + j * 27);
#pragma unroll
        for (int l = 0; l < 9; ++l) st[l * ls + j * js] = uj[l];
      }
#pragma nounroll
      for (u32 k = 0; k < rp; ++k) {
        ZkTab29 pk = P + (size_t)k * T * 27;
        u32 y0[9];
        zk_l29_add(y0, u0, pk);
        zk_p29_sbox<V>(y0, emit + 3 * (8 * T + k));
        ZkW29 w;
        zk_w29_zero(w);
        zk_w29_mac(w, y0, pk + 9);
        u32 j = 1;
        if (V & 2) {
#pragma nounroll
          for (; j + 1 < (u32)T; j += 2) {
            ZkTab29 pj = pk + j * 27;
            u32 ua[9], ub[9], pa[9], pb[9];
#pragma unroll
            for (int l = 0; l < 9; ++l) { ua[l] = st[l * ls + j * js]; ub[l] = st[l * ls + (j + 1) * js]; }
            zk_w29_mac(w, ua, pj + 9);
            zk_w29_mac(w, ub, pj + 36);
            if ((j & 3u) == 3u) zk_w29_carry(w);       // j = 1, 3, 5 ...: every second pair -> at most 5 multiply-accumulates apart
            zk_l29_mulv<V>(pa, y0, pj + 18);
            zk_l29_mulv<V>(pb, y0, pj + 45);
            zk_l29_add3(ua, ua, pa, pj + T * 27);
            zk_l29_add3(ub, ub, pb, pj + T * 27 + 27);
#pragma unroll
            for (int l = 0; l < 9; ++l) { st[l * ls + j * js] = ua[l]; st[l * ls + (j + 1) * js] = ub[l]; }
          }
        }
#pragma nounroll
        for (; j < (u32)T; ++j) {
          ZkTab29 pj = pk + j * 27;
          u32 uj[9], p[9];
#pragma unroll
          for (int l = 0; l < 9; ++l) uj[l] = st[l * ls + j * js];
          zk_w29_mac(w, uj, pj + 9);
          if ((j & 3u) == 3u) zk_w29_carry(w);
          zk_l29_mulv<V>(p, y0, pj + 18);
          zk_l29_add3(uj, uj, p, pj + T * 27);
#pragma unroll
          for (int l = 0; l < 9; ++l) st[l * ls + j * js] = uj[l];
        }
        zk_w29_carry(w);
        zk_w29_redc(w, u0);
      }
#pragma unroll
      for (int l = 0; l < 9; ++l) st[l * ls] = u0[l];
      zk_p29_densev<T, V>(st, js, ls, B, T, stage, ss);
    }
    for (u32 r = 0; r < 4; ++r) {
      ZkTab29 c = (half ? cL : cF) + 9 * r * T;
      u32 j = 0;
      if (V & 2) {
#pragma nounroll
        for (; j + 1 < (u32)T; j += 2) {
          u32 xa[9], xb[9];
#pragma unroll
          for (int l = 0; l < 9; ++l) { xa[l] = st[l * ls + j * js]; xb[l] = st[l * ls + (j + 1) * js]; }
          zk_l29_add(xa, xa, c + 9 * j);
          zk_l29_add(xb, xb, c + 9 * j + 9);
          zk_p29_sbox<V>(xa, emit + 3 * ((half * 4 + r) * T + j));
          zk_p29_sbox<V>(xb, emit + 3 * ((half * 4 + r) * T + j + 1));
#pragma unroll
          for (int l = 0; l < 9; ++l) { st[l * ls + j * js] = xa[l]; st[l * ls + (j + 1) * js] = xb[l]; }
        }
      }
#pragma nounroll
      for (; j < (u32)T; ++j) {
        u32 x[9];
#pragma unroll
        for (int l = 0; l < 9; ++l) x[l] = st[l * ls + j * js];
        zk_l29_add(x, x, c + 9 * j);
        zk_p29_sbox<V>(x, emit + 3 * ((half * 4 + r) * T + j));
#pragma unroll
        for (int l = 0; l < 9; ++l) st[l * ls + j * js] = x[l];
      }
      zk_p29_densev<T, V>(st, js, ls, M, (half == 1 && r == 3) ? 1u : (u32)T, stage, ss);
    }
  }
  u32 h[9];
#pragma unroll
  for (int l = 0; l < 9; ++l) h[l] = st[l * ls];
  return zk_l29_to_fr(h);
}

// Host: the limb table from the Fr table of zk_build_poseidon_sparse(t, rp, ...)
static inline void zk_build_poseidon29(u32 t, u32 rp, const std::vector<Fr>& tab, std::vector<u32>& out) {
  const Fr* c_first = tab.data();
  const Fr* mt = c_first + 4 * t;
  const Fr* c_part = mt + t * t;
  const Fr* s_part = c_part + rp * t;
  const Fr* bt = s_part + rp * (2 * t - 1);
  const Fr* c_last = bt + t * t;
  out.assign(zk_p29_tab_size(t, rp), 0u);
  u32* o = out.data();
  auto put = [&](const Fr& v, bool mult) {
    Fr x = v;
    if (mult) for (int i = 0; i < 5; ++i) x = fr_add(x, x);      // times 2^256 -> times 2^261
    u32 l[9];
    zk_l29_from_fr(x, l);
    for (int i = 0; i < 9; ++i) *o++ = l[i];
  };
  for (u32 i = 0; i < 4 * t; ++i) put(c_first[i], false);
  for (u32 i = 0; i < t; ++i) for (u32 j = 0; j < t; ++j) put(mt[j * t + i], true);
  for (u32 k = 0; k < rp; ++k) {
    const Fr* sk = s_part + k * (2 * t - 1);
    for (u32 j = 0; j < t; ++j) {
      put(c_part[k * t + j], false);
      put(sk[j], true);
      put(j ? sk[t - 1 + j] : fr_zero(), true);
    }
  }
  for (u32 j = 0; j < 3 * t; ++j) put(fr_zero(), false);
  for (u32 i = 0; i < t; ++i) for (u32 j = 0; j < t; ++j) put(bt[j * t + i], true);
  for (u32 i = 0; i < 4 * t; ++i) put(c_last[i], false);
}
