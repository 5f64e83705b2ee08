// BN254 G1 and G2 point arithmetic in lazy 29-bit limb form (zkwg_fq29.h) -- what the multi-exponentiation kernels run since round 6.
// One set of XYZZ formulas (EFD madd-2008-s / add-2008-s / dbl-2008-s-1, the ones zkwg_g1.h / zkwg_g2.h state over canonical words),
// written over a FIELD TRAIT:
//
//   ZkF1   Fq: one lane owns one element (G1).
//   ZkF2   Fq2 = Fq[i] / (i^2 + 1) (G2).  On the device the two halves of an element live on a LANE PAIR (even lane c0, odd lane c1):
//          (a0 + a1 i)(b0 + b1 i) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) i is one two-product dot product with ONE reduction per lane
//          (fq29_dot2: 243 multiply-adds, exactly half of Karatsuba's three products + no Karatsuba additions), the partner's limbs
//          arrive by DPP (quad_perm [1,0,3,2]: 18 moves per product), and a lane holds 36 registers of an accumulator instead of 64 + 64
//          words: the round-5 G2 kernels (one lane per point, Fq2 products as real function calls to keep the compile time down) sat at
//          256 VGPRs + 1,168 bytes of scratch, occupancy 1.  On the host (CPU mirror of the kernels) an element is
//          the pair itself and the same dot products run for both halves, so the CPU tests execute the device's arithmetic and its bounds.
//
// Bounds (notation of zkwg_fq29.h: [U, V] = limbs < U 2^29, value < V q).  Stored accumulators: X = [1, 11], Y = [1, 7] (G1: [1, 2] --
// Y3 = R (Q - X3) - Y1 PPP is ONE two-product dot product there, its result normalised), ZZ, ZZZ = [1, 2]; ZZ = 0 (all limbs) = infinity.
// Bases: canonical words in 2^261-Montgomery form, [1, 1].  The bound of every intermediate is written beside it; `mul<VB>` names the
// value bound of its RIGHT operand (G2's even lane needs a multiple of q above it to negate the partner's half); the host build counts
// violated preconditions (ZKWG_FQ29_CHECK).
#pragma once
#include "zkwg_fq29.h"
#include "zkwg_g1.h"
#include "zkwg_g2.h"

// ---- Fq: one lane per element ----------------------------------------------------------------------------------------------------------
struct ZkF1 {
  typedef Fq29 E;
  template <int VB> static ZK_HD E mul(const E& a, const E& b) { return fq29_mul(a, b); }
  template <int VA> static ZK_HD E sqr(const E& a) { return fq29_sqr(a); }
  // a b - c d with ONE reduction: d = [1, <= VD], Ua Ub + 2 Uc <= 6; the result is normalised
  template <int VB, int VD> static ZK_HD E msub(const E& a, const E& b, const E& c, const E& d) { return fq29_dot2(a, b, c, fq29_neg<VD + 1, 1>(d)); }
  static ZK_HD E scale(const E& a, const Fq29& k) { return fq29_mul(a, k); }
  static ZK_HD E add(const E& a, const E& b) { return fq29_add(a, b); }
  static ZK_HD E dbl(const E& a) { return fq29_dbl(a); }
  template <int M, int U> static ZK_HD E sub(const E& a, const E& b) { return fq29_sub<M, U>(a, b); }
  static ZK_HD E norm(const E& a) { return fq29_norm(a); }
  static ZK_HD E zero() { return fq29_zero(); }
  static ZK_HD E one() { return fq29_one(); }
  static ZK_HD bool all_zero(const E& a) { return fq29_all_zero(a); }
  template <int V> static ZK_HD bool maybe_zero(const E& a) { return fq29_maybe_zero<V>(a); }
  template <int V> static ZK_HD bool is_zero_mod(const E& a) { return fq29_is_zero_mod<V>(a); }
};

// ---- Fq2 ---------------------------------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u32 zk_pair_xchg(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ Fq29 zk_pair_xchg(const Fq29& a) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = zk_pair_xchg(a.l[i]);
  return r;
}
struct ZkF2 {
  typedef Fq29 E;                       // this lane's half: c0 on even lanes, c1 on odd lanes
  static __device__ __forceinline__ bool odd() { return (threadIdx.x & 1u) != 0; }
  template <int VB> static __device__ __forceinline__ E mul(const E& a, const E& b) {
    const Fq29 oa = zk_pair_xchg(a), ob = zk_pair_xchg(b);
    const Fq29 nb = fq29_neg<VB + 1, 1>(ob);
    const bool o = odd();
    Fq29 S, T;
#pragma unroll
    for (int i = 0; i < 9; ++i) { S.l[i] = o ? ob.l[i] : b.l[i]; T.l[i] = o ? b.l[i] : nb.l[i]; }
    return fq29_dot2(a, S, oa, T);      // even: a0 b0 + a1 (-b1); odd: a1 b0 + a0 b1
  }
  template <int VA> static __device__ __forceinline__ E sqr(const E& a) { return mul<VA>(a, a); }
  template <int VB, int VD> static __device__ __forceinline__ E msub(const E& a, const E& b, const E& c, const E& d) {
    return fq29_norm(fq29_sub<3, 1>(mul<VB>(a, fq29_norm(b)), mul<VD>(c, d)));       // (c d < 2 q at every call site; two reductions: a four-product dot product per lane would save one)
  }
  static __device__ __forceinline__ E scale(const E& a, const Fq29& k) { return fq29_mul(a, k); }
  static __device__ __forceinline__ E add(const E& a, const E& b) { return fq29_add(a, b); }
  static __device__ __forceinline__ E dbl(const E& a) { return fq29_dbl(a); }
  template <int M, int U> static __device__ __forceinline__ E sub(const E& a, const E& b) { return fq29_sub<M, U>(a, b); }
  static __device__ __forceinline__ E norm(const E& a) { return fq29_norm(a); }
  static __device__ __forceinline__ E zero() { return fq29_zero(); }
  static __device__ __forceinline__ E one() { return odd() ? fq29_zero() : fq29_one(); }
  static __device__ __forceinline__ bool both(bool mine) { const u32 other = zk_pair_xchg(mine ? 1u : 0u); return mine & (other != 0); }
  static __device__ __forceinline__ bool all_zero(const E& a) { return both(fq29_all_zero(a)); }
  template <int V> static __device__ __forceinline__ bool maybe_zero(const E& a) { return both(fq29_maybe_zero<V>(a)); }
  template <int V> static __device__ __forceinline__ bool is_zero_mod(const E& a) { return both(fq29_is_zero_mod<V>(a)); }
};
#else
struct Fq29x2 { Fq29 c[2]; };
struct ZkF2 {
  typedef Fq29x2 E;
  template <int VB> static inline E mul(const E& a, const E& b) {
    return E{{fq29_dot2(a.c[0], b.c[0], a.c[1], fq29_neg<VB + 1, 1>(b.c[1])), fq29_dot2(a.c[1], b.c[0], a.c[0], b.c[1])}};
  }
  template <int VA> static inline E sqr(const E& a) { return mul<VA>(a, a); }
  template <int VB, int VD> static inline E msub(const E& a, const E& b, const E& c, const E& d) { return norm(sub<3, 1>(mul<VB>(a, norm(b)), mul<VD>(c, d))); }
  static inline E scale(const E& a, const Fq29& k) { return E{{fq29_mul(a.c[0], k), fq29_mul(a.c[1], k)}}; }
  static inline E add(const E& a, const E& b) { return E{{fq29_add(a.c[0], b.c[0]), fq29_add(a.c[1], b.c[1])}}; }
  static inline E dbl(const E& a) { return add(a, a); }
  template <int M, int U> static inline E sub(const E& a, const E& b) { return E{{fq29_sub<M, U>(a.c[0], b.c[0]), fq29_sub<M, U>(a.c[1], b.c[1])}}; }
  static inline E norm(const E& a) { return E{{fq29_norm(a.c[0]), fq29_norm(a.c[1])}}; }
  static inline E zero() { return E{{fq29_zero(), fq29_zero()}}; }
  static inline E one() { return E{{fq29_one(), fq29_zero()}}; }
  static inline bool all_zero(const E& a) { return fq29_all_zero(a.c[0]) && fq29_all_zero(a.c[1]); }
  template <int V> static inline bool maybe_zero(const E& a) { return fq29_maybe_zero<V>(a.c[0]) && fq29_maybe_zero<V>(a.c[1]); }
  template <int V> static inline bool is_zero_mod(const E& a) { return fq29_is_zero_mod<V>(a.c[0]) && fq29_is_zero_mod<V>(a.c[1]); }
};
#endif

// ---- points ------------------------------------------------------------------------------------------------------------------------------
template <class F> struct Aff29 { typename F::E x, y; bool inf; };          // x [1, 1], y [2, 2] (the negated y of a negative digit is 2 q - y, unnormalised)
template <class F> struct alignas(16) Xyzz29 { typename F::E x, y, zz, zzz; };          // X [1, 11], Y [1, 7], ZZ, ZZZ [1, 2]; ZZ all zero = infinity

template <class F> ZK_HD Xyzz29<F> ec29_inf() { return Xyzz29<F>{F::zero(), F::zero(), F::zero(), F::zero()}; }
template <class F> ZK_HD bool ec29_is_inf(const Xyzz29<F>& p) { return F::all_zero(p.zz); }
template <class F> ZK_HD Xyzz29<F> ec29_from_affine(const Aff29<F>& p) { return p.inf ? ec29_inf<F>() : Xyzz29<F>{p.x, F::norm(p.y), F::one(), F::one()}; }

// 2 P for an affine P (the equal-points case of a mixed addition: rare, not tuned)
template <class F>
ZK_HD Xyzz29<F> ec29_dbl_affine(const Aff29<F>& p) {
  typedef typename F::E E;
  if (p.inf) return ec29_inf<F>();
  const E py = F::norm(p.y);                                      // [1, 2]
  if (F::template is_zero_mod<2>(py)) return ec29_inf<F>();       // (no point of order 2 on these curves; kept for completeness)
  const E U = F::norm(F::dbl(py));                                // [1, 4]
  const E V = F::template sqr<4>(U);                              // [1, 2]
  const E W = F::template mul<2>(U, V);                           // [1, 2]
  const E S = F::template mul<2>(p.x, V);                         // [1, 2]
  const E X2 = F::template sqr<1>(p.x);                           // [1, 2]
  const E M = F::norm(F::add(F::dbl(X2), X2));                    // [1, 6]
  const E MM = F::template sqr<6>(M);                             // [1, 2]
  Xyzz29<F> r;
  r.x = F::norm(F::template sub<5, 2>(MM, F::dbl(S)));            // [1, 7]
  const E T = F::template sub<12, 1>(S, r.x);                     // [3, 14]
  r.y = F::template msub<14, 2>(M, T, W, py);                     // [1, 7]
  r.zz = V; r.zzz = W;
  return r;
}
// 2 P
template <class F>
ZK_HD Xyzz29<F> ec29_dbl(const Xyzz29<F>& p) {
  typedef typename F::E E;
  if (ec29_is_inf(p)) return p;
  const E U = F::norm(F::dbl(p.y));                               // [1, 14]
  const E V = F::template sqr<14>(U);                             // [1, 4]
  const E W = F::template mul<4>(U, V);                           // [1, 2]
  const E S = F::template mul<4>(p.x, V);                         // [1, 2]
  const E X2 = F::template sqr<11>(p.x);                          // [1, 3]
  const E M = F::norm(F::add(F::dbl(X2), X2));                    // [1, 9]
  const E MM = F::template sqr<9>(M);                             // [1, 3]
  Xyzz29<F> r;
  r.x = F::norm(F::template sub<5, 2>(MM, F::dbl(S)));            // [1, 8]
  const E T = F::template sub<12, 1>(S, r.x);                     // [3, 14]
  r.y = F::template msub<14, 7>(M, T, W, p.y);                    // [1, 7]
  r.zz = F::template mul<2>(V, p.zz);
  r.zzz = F::template mul<2>(W, p.zzz);
  return r;
}
// acc + P for an affine P: 7 M + 2 S + one two-product dot product, three carry normalisations
template <class F>
ZK_HD Xyzz29<F> ec29_add_mixed(const Xyzz29<F>& a, const Aff29<F>& p) {
  typedef typename F::E E;
  if (p.inf) return a;
  if (ec29_is_inf(a)) return Xyzz29<F>{p.x, F::norm(p.y), F::one(), F::one()};
  // (ordered so that an input dies as early as possible: x2, y2, then ZZ1, X1, ZZZ1, Y1 -- the kernels' register budget)
  const E P = F::norm(F::template sub<12, 1>(F::template mul<2>(p.x, a.zz), a.x));         // U2 [1, 2] - X1 -> [1, 14]
  const E Rr = F::norm(F::template sub<8, 1>(F::template mul<2>(p.y, a.zzz), a.y));        // S2 [1, 2] - Y1 -> [1, 10]
  if (F::template maybe_zero<14>(P)) {
    if (F::template is_zero_mod<14>(P)) return F::template is_zero_mod<10>(Rr) ? ec29_dbl_affine<F>(p) : ec29_inf<F>();
  }
  Xyzz29<F> r;
  const E PP = F::template sqr<14>(P);                            // [1, 4]
  r.zz = F::template mul<4>(a.zz, PP);
  const E Q = F::template mul<4>(a.x, PP);                        // [1, 2]
  const E PPP = F::template mul<4>(P, PP);                        // [1, 2]
  r.zzz = F::template mul<2>(a.zzz, PPP);
  const E RR = F::template sqr<10>(Rr);                           // [1, 3]
  r.x = F::norm(F::template sub<5, 2>(F::template sub<3, 1>(RR, PPP), F::dbl(Q)));     // [6, 11] -> [1, 11]
  const E T = F::template sub<12, 1>(Q, r.x);                     // [3, 14]
  r.y = F::template msub<14, 2>(Rr, T, a.y, PPP);                 // R (Q - X3) - Y1 PPP: [1, 7] (G1: [1, 2])
  return r;
}
// a + b: 11 M + 2 S + one two-product dot product
template <class F>
ZK_HD Xyzz29<F> ec29_add(const Xyzz29<F>& a, const Xyzz29<F>& b) {
  typedef typename F::E E;
  if (ec29_is_inf(a)) return b;
  if (ec29_is_inf(b)) return a;
  const E U1 = F::template mul<2>(a.x, b.zz);                                              // [1, 2]
  const E P = F::norm(F::template sub<3, 1>(F::template mul<2>(b.x, a.zz), U1));           // U2 - U1 -> [1, 5]
  const E S1 = F::template mul<2>(a.y, b.zzz);                                             // [1, 2]
  const E Rr = F::norm(F::template sub<3, 1>(F::template mul<2>(b.y, a.zzz), S1));         // S2 - S1 -> [1, 5]
  if (F::template maybe_zero<5>(P)) {
    if (F::template is_zero_mod<5>(P)) return F::template is_zero_mod<5>(Rr) ? ec29_dbl<F>(a) : ec29_inf<F>();
  }
  Xyzz29<F> r;
  const E PP = F::template sqr<5>(P);                             // [1, 2]
  r.zz = F::template mul<2>(F::template mul<2>(a.zz, b.zz), PP);
  const E Q = F::template mul<2>(U1, PP);                         // [1, 2]
  const E PPP = F::template mul<2>(P, PP);                        // [1, 2]
  r.zzz = F::template mul<2>(F::template mul<2>(a.zzz, b.zzz), PPP);
  const E RR = F::template sqr<5>(Rr);                            // [1, 2]
  r.x = F::norm(F::template sub<5, 2>(F::template sub<3, 1>(RR, PPP), F::dbl(Q)));     // [1, 10]
  const E T = F::template sub<12, 1>(Q, r.x);                     // [3, 14]
  r.y = F::template msub<14, 2>(Rr, T, S1, PPP);                  // [1, 7]
  return r;
}

// ---- the two groups as the kernels see them ------------------------------------------------------------------------------------------------
// Affine: the memory form of a base (the zkey's layout; tables hold the 2^261 form), Out: the accumulator handed back to the host
// (zkwg_g1.h / zkwg_g2.h XYZZ, 2^256 form), LANES: lanes per point, load / store per lane half h.
// -y of a base for a negative digit, in limb form: 2 q - y without borrows (9 subtractions + 9 selects; the canonical-word negation was a
// 256-bit compare-and-subtract in front of the split: 45 instructions and 12 wait states per addition).  y canonical -> [2, 2].
ZK_HD Fq29 zk_q29_neg_if(const Fq29& y, bool neg) {
  const Fq29 n = fq29_neg<2, 1>(y);
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = neg ? n.l[i] : y.l[i];
  return r;
}
// one coordinate (32 bytes, 16-byte aligned: tables come from hipMalloc) as two 16-byte loads
ZK_HD Fq zk_ld_fq(const Fq* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4 a = ((const uint4*)p)[0], b = ((const uint4*)p)[1];
  return Fq{{(u64)a.x | ((u64)a.y << 32), (u64)a.z | ((u64)a.w << 32), (u64)b.x | ((u64)b.y << 32), (u64)b.z | ((u64)b.w << 32)}};
#else
  return *p;
#endif
}
struct ZkEcG1 {
  typedef ZkF1 F;
  typedef G1Affine Affine;
  typedef G1Xyzz Out;
  static constexpr int LANES = 1;
  static constexpr int DEV_LANES = 1;      // lanes per point in the kernels (what the host sizes its launches with)
  static ZK_HD Aff29<F> load(const Affine* p, u32 h, bool neg) {
    const Fq x = zk_ld_fq(&p->x), y = zk_ld_fq(&p->y);
    return Aff29<F>{fq29_from_fq(x), zk_q29_neg_if(fq29_from_fq(y), neg), fq_is_zero(x) && fq_is_zero(y)};
  }
  static ZK_HD void store_out(Out* o, const Xyzz29<F>& p, u32 h) {
    const Fq29 k = fq29_r256();
    o->x = fq29_to_fq<2>(fq29_mul(p.x, k)); o->y = fq29_to_fq<2>(fq29_mul(p.y, k));
    o->zz = fq29_to_fq<2>(fq29_mul(p.zz, k)); o->zzz = fq29_to_fq<2>(fq29_mul(p.zzz, k));
  }
};
struct ZkEcG2 {
  typedef ZkF2 F;
  typedef G2Affine Affine;
  typedef G2Xyzz Out;
  static constexpr int DEV_LANES = 2;
#if defined(__HIP_DEVICE_COMPILE__)
  static constexpr int LANES = 2;
  static __device__ __forceinline__ Aff29<F> load(const Affine* p, u32 h, bool neg) {
    const Fq* w = (const Fq*)p;                      // x.c0 | x.c1 | y.c0 | y.c1
    const Fq x = zk_ld_fq(w + h), y = zk_ld_fq(w + 2 + h);
    const bool z = F::both(fq_is_zero(x) && fq_is_zero(y));
    return Aff29<F>{fq29_from_fq(x), zk_q29_neg_if(fq29_from_fq(y), neg), z};
  }
  static __device__ __forceinline__ void store_out(Out* o, const Xyzz29<F>& p, u32 h) {
    const Fq29 k = fq29_r256();
    Fq* w = (Fq*)o;                                  // x.c0 | x.c1 | y.c0 | y.c1 | zz.c0 | zz.c1 | zzz.c0 | zzz.c1
    w[h] = fq29_to_fq<2>(fq29_mul(p.x, k)); w[2 + h] = fq29_to_fq<2>(fq29_mul(p.y, k));
    w[4 + h] = fq29_to_fq<2>(fq29_mul(p.zz, k)); w[6 + h] = fq29_to_fq<2>(fq29_mul(p.zzz, k));
  }
#else
  static constexpr int LANES = 1;
  static inline Aff29<F> load(const Affine* p, u32 h, bool neg) {
    const Affine a = *p;
    return Aff29<F>{Fq29x2{{fq29_from_fq(a.x.c0), fq29_from_fq(a.x.c1)}}, Fq29x2{{zk_q29_neg_if(fq29_from_fq(a.y.c0), neg), zk_q29_neg_if(fq29_from_fq(a.y.c1), neg)}}, g2_is_inf(a)};
  }
  static inline void store_out(Out* o, const Xyzz29<F>& p, u32 h) {
    const Fq29 k = fq29_r256();
    auto cv = [&](const Fq29x2& e) { return Fq2{fq29_to_fq<2>(fq29_mul(e.c[0], k)), fq29_to_fq<2>(fq29_mul(e.c[1], k))}; };
    o->x = cv(p.x); o->y = cv(p.y); o->zz = cv(p.zz); o->zzz = cv(p.zzz);
  }
#endif
};

// canonical words of the zkey (x 2^256 mod q) -> canonical words of the tables (x 2^261 mod q); the point at infinity (zeros) stays zeros
ZK_HD Fq zk_fq_r256_to_r261(const Fq& a) { return fq29_to_fq<2>(fq29_mul(fq29_from_fq(a), fq29_t266())); }
ZK_HD G1Affine zk_g1_to_table_form(const G1Affine& p) { return G1Affine{zk_fq_r256_to_r261(p.x), zk_fq_r256_to_r261(p.y)}; }
ZK_HD G2Affine zk_g2_to_table_form(const G2Affine& p) {
  return G2Affine{Fq2{zk_fq_r256_to_r261(p.x.c0), zk_fq_r256_to_r261(p.x.c1)}, Fq2{zk_fq_r256_to_r261(p.y.c0), zk_fq_r256_to_r261(p.y.c1)}};
}
