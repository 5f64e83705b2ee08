// Multi-exponentiations over BN254 G1 / G2 for the prover (second half of groth16.fullProve, reference call site
// packages/helpers/src/chunked-zkey.ts:80-84): the per-thread bodies of the kernels of zkwg_kernels_msm.hip, shared with the host mirror
// of the CPU tests (tests/native/hosttest.cpp runs exactly these functions, thread by thread, in launch order).
//
// Round 6: EMAILS ARE THE PARALLEL AXIS here as everywhere else in zkwg.  One launch series sums E scalar vectors against ONE set of bases
// (every kernel below takes the email as its second grid dimension and its own slice of one work buffer), so the thin tails of a sum --
// a few lanes folding bit planes, one lane combining windows -- are E lanes wide and the number of launches per proof falls by E; round 5
// ran one launch chain per proof on its own stream and depended on GPU_MAX_HW_QUEUES to overlap them.  Point arithmetic is the lazy
// 29-bit limb form of zkwg_ec29.h (G2 on lane pairs).  Steps of one series:
//
//   zk_msm_classify  (when the scalars are a witness: > 95 % zeros and ones) one pass over the scalars that writes, per email, the index
//                    list of the scalars that are 1 and the list of the others (not 0, not 1), skipping bases at infinity (bit map made at
//                    plan creation).  The prover runs it ONCE for its four witness-shaped sums (three base sets: A, B1 = B2, C).
//   zk_msm_ones      the bases of the `ones` list, ZK_MSM_ONES per lane, then ZK_MSM_JOIN-way joins (log depth)
//   zk_msm_sort      counting sort of the other scalars' signed c-bit digits by bucket: count -> scan -> scatter, with workgroup-local
//                    histograms in LDS when the bucket counters fit (one global atomic per bucket and workgroup, not per digit)
//   zk_msm_slice_*   a bucket's run of entries is summed in SLICES by as many lanes as it needs (level 0: s0 mixed additions of bases per
//                    lane, levels 1 and 2: 8 partial sums per lane), then one lane per bucket joins what is left
//   zk_msm_plane*    the weighted bucket sum  sum_b (b + 1) bucket[b]  by bit planes: T_j = plain sum of the buckets whose weight has bit j
//                    set (8-way joins, c rows at once), one lane folds sum_j 2^j T_j with c - 1 doublings
//   zk_msm_combine   windows (Horner, classic layout only) + the ones' sum -> the accumulator in zkwg_g1.h / zkwg_g2.h XYZZ words
//
// Precomputed windows (default): the table holds K shifted copies of the bases, copy w = 2^(c w) base, so digit d of window w selects
// copy w and ALL windows share one set of 2^(c-1) buckets (KS = 1); the classic layout (KS = K bucket sets) is what is left when the
// copies do not fit.  Tables are in 2^261-Montgomery form (zk_msm_table_thread converts the zkey's 2^256 form once).
#pragma once
#include "zkwg_ec29.h"

#define ZK_MSM_JOIN 8u     // fan-in of the joins of the ones' partial sums
#define ZK_MSM_ONES 16u    // list entries per lane of zk_msm_ones
#define ZK_MSM_PFAN 8u     // fan-in of the bit-plane joins
#define ZK_MSM_S1 8u       // partial sums per lane at slice levels 1 and 2
#define ZK_MSM_LDS_BUCKETS 32768u

// offsets of one email's arrays inside its slice of the work buffer
struct ZkMsmOff {
  u64 count, cursor, entry, bucket, node_s, node_a, window, ones, soff[3], part[3], total;
  u32 cap[3];
};

template <class C>
struct ZkMsmArgsT {
  typedef typename C::Affine Affine;
  typedef Xyzz29<typename C::F> X;
  const Affine* table;        // KS == 1: K copies of the n bases (copy w at table + w n); KS == K: the n bases.  2^261 form, zeros = infinity
  const u32* inf;             // bit i set: base i is the point at infinity
  const Fr* scalars;          // email e's scalars at scalars + e * scalar_stride
  u64 scalar_stride;
  u32 n, c, K, nb, KS, stride;   // points, window bits, windows, buckets per set = 2^(c-1), bucket sets, stride = n (KS == 1) or 0
  u32 E;                      // emails of this launch series
  u32 scalars_mont;           // 1: the scalars are in Montgomery form (the H evaluations of zkwg_ntt_api.hip)
  u32 lds_sort;               // 1: workgroup-local histograms (KS nb <= ZK_MSM_LDS_BUCKETS)
  u32 s0;                     // entries per lane at slice level 0 (16 for a witness' few full-size scalars, 64 for 2^21 of them)
  // index lists (zk_msm_classify).  sel == nullptr: every scalar enters the buckets (the H sum); ones == nullptr: no ones' sum
  const u32* sel; const u32* n_sel;       // email e: sel + e * list_stride, n_sel[e] entries
  const u32* ones; const u32* n_ones;
  u64 list_stride;
  u8* work; u64 work_stride;  // email e's arrays: work + e * work_stride + off.*
  ZkMsmOff off;
  typename C::Out* out;       // [E]
  ZK_HD u8* w(u32 e) const { return work + (u64)e * work_stride; }
  ZK_HD u32* count(u32 e) const { return (u32*)(w(e) + off.count); }
  ZK_HD u32* cursor(u32 e) const { return (u32*)(w(e) + off.cursor); }
  ZK_HD u32* entry(u32 e) const { return (u32*)(w(e) + off.entry); }
  ZK_HD X* bucket(u32 e) const { return (X*)(w(e) + off.bucket); }
  ZK_HD X* node_s(u32 e) const { return (X*)(w(e) + off.node_s); }
  ZK_HD X* node_a(u32 e) const { return (X*)(w(e) + off.node_a); }
  ZK_HD X* window(u32 e) const { return (X*)(w(e) + off.window); }
  ZK_HD X* ones_acc(u32 e) const { return (X*)(w(e) + off.ones); }
  ZK_HD u32* soff(u32 e, int l) const { return (u32*)(w(e) + off.soff[l]); }
  ZK_HD X* part(u32 e, int l) const { return (X*)(w(e) + off.part[l]); }
  ZK_HD u32 sel_count(u32 e) const { return sel ? n_sel[e] : n; }
  ZK_HD u32 sel_at(u32 e, u32 j) const { return sel ? sel[(u64)e * list_stride + j] : j; }
  ZK_HD bool base_inf(u32 i) const { return (inf[i >> 5] >> (i & 31u)) & 1u; }
};

#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_MSM_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#else
static inline u32 zk_msm_host_fetch_add(u32* p, u32 v) { const u32 o = *p; *p = o + v; return o; }
#define ZK_MSM_ATOMIC_ADD(p, v) zk_msm_host_fetch_add((p), (v))
#endif

ZK_HD bool zk_msm_is_one(const Fr& k) { return k.l[0] == 1 && (k.l[1] | k.l[2] | k.l[3]) == 0; }
ZK_HD Fr zk_msm_std(const Fr& k, u32 mont) { return mont ? fr_from_mont(k) : k; }
template <class C>
ZK_HD Fr zk_msm_scalar(const ZkMsmArgsT<C>& A, u32 e, u32 i) { return zk_msm_std(A.scalars[(u64)e * A.scalar_stride + i], A.scalars_mont); }

// ---- classification: one pass over E scalar vectors for up to three base sets ---------------------------------------------------------
struct ZkClassifyTarget {
  const u32* inf;             // the base set's infinity bits
  u32 first, n;               // the set covers the scalars [first, first + n); list entries are relative to `first`
  u32* sel; u32* n_sel;       // email e: sel + e * list_stride; n_sel[e] (zeroed by the caller)
  u32* ones; u32* n_ones;
  u64 list_stride;
};
struct ZkClassifyArgs {
  const Fr* scalars; u64 scalar_stride;
  u32 n, E, scalars_mont, ones_apart, n_targets;
  ZkClassifyTarget t[3];
};
// where scalar i of email e goes: bit 2 t = the ones' list of target t, bit 2 t + 1 = its list of other non-zero scalars
ZK_HD u32 zk_msm_classify_code(const ZkClassifyArgs& A, u32 e, u32 i) {
  if (i >= A.n) return 0;
  const Fr k = zk_msm_std(A.scalars[(u64)e * A.scalar_stride + i], A.scalars_mont);
  if (fr_is_zero(k)) return 0;
  const bool one = A.ones_apart && zk_msm_is_one(k);
  u32 code = 0;
  for (u32 t = 0; t < A.n_targets; ++t) {
    const ZkClassifyTarget& T = A.t[t];
    const u32 r = i - T.first;
    if (i >= T.first && r < T.n && !((T.inf[r >> 5] >> (r & 31u)) & 1u)) code |= (one ? 1u : 2u) << (2 * t);
  }
  return code;
}
// list l (= 2 t + which) of email e: counter and entries
ZK_HD u32* zk_msm_classify_counter(const ZkClassifyArgs& A, u32 e, u32 l) { return (l & 1u ? A.t[l >> 1].n_sel : A.t[l >> 1].n_ones) + e; }
ZK_HD u32* zk_msm_classify_list(const ZkClassifyArgs& A, u32 e, u32 l) { return (l & 1u ? A.t[l >> 1].sel : A.t[l >> 1].ones) + (u64)e * A.t[l >> 1].list_stride; }
// the host mirror's form: one scalar at a time (the kernel reserves places per workgroup -- zkwg_kernels_msm.hip -- the lists are the same sets)
ZK_HD void zk_msm_classify_thread(const ZkClassifyArgs& A, u32 e, u32 i) {
  const u32 code = zk_msm_classify_code(A, e, i);
  for (u32 l = 0; l < 2 * A.n_targets; ++l)
    if ((code >> l) & 1u) { u32* cnt = zk_msm_classify_counter(A, e, l); zk_msm_classify_list(A, e, l)[ZK_MSM_ATOMIC_ADD(cnt, 1u)] = i - A.t[l >> 1].first; }
}

// ---- the ones' sum ---------------------------------------------------------------------------------------------------------------------
// partial sum t of email e: list entries [t ONES, (t + 1) ONES); there is always a partial 0 (infinity for an empty list)
ZK_HD u32 zk_msm_ones_parts(u32 n_ones) { const u32 m = (n_ones + ZK_MSM_ONES - 1u) / ZK_MSM_ONES; return m ? m : 1u; }
template <class C>
ZK_HD void zk_msm_ones_thread(const ZkMsmArgsT<C>& A, u32 e, u32 t, u32 h, typename ZkMsmArgsT<C>::X* out) {
  const u32 n1 = A.n_ones[e];
  if (t >= zk_msm_ones_parts(n1)) return;
  const u32 lo = t * ZK_MSM_ONES, hi = lo + ZK_MSM_ONES < n1 ? lo + ZK_MSM_ONES : n1;
  const u32* list = A.ones + (u64)e * A.list_stride;
  // (no software prefetch of the next base: it costs 18 registers and a copy per addition; four (G1) / three (G2) wavefronts per SIMD hide
  // the gather -- measured both ways, profiles/r06/r06_l, r06_s)
  typename ZkMsmArgsT<C>::X acc = ec29_inf<typename C::F>();
  for (u32 j = lo; j < hi; ++j) acc = ec29_add_mixed<typename C::F>(acc, C::load(A.table + list[j], h, false));
  out[(u64)t * C::LANES + h] = acc;
}
// one join level: m_in partial sums -> ceil(m_in / JOIN); m_in follows from the email's list length and the level
ZK_HD u32 zk_msm_ones_level_count(u32 n_ones, u32 level) {
  u32 m = zk_msm_ones_parts(n_ones);
  for (u32 l = 0; l < level; ++l) m = (m + ZK_MSM_JOIN - 1u) / ZK_MSM_JOIN;
  return m;
}
template <class C>
ZK_HD void zk_msm_tree_thread(const ZkMsmArgsT<C>& A, u32 e, u32 level, u32 t, u32 h, const typename ZkMsmArgsT<C>::X* in, typename ZkMsmArgsT<C>::X* out) {
  const u32 m_in = zk_msm_ones_level_count(A.n_ones[e], level);
  const u32 lo = t * ZK_MSM_JOIN, hi = lo + ZK_MSM_JOIN < m_in ? lo + ZK_MSM_JOIN : m_in;
  if (lo >= m_in) return;
  typename ZkMsmArgsT<C>::X acc = in[(u64)lo * C::LANES + h];
  for (u32 i = lo + 1; i < hi; ++i) acc = ec29_add<typename C::F>(acc, in[(u64)i * C::LANES + h]);
  out[(u64)t * C::LANES + h] = acc;
}

// ---- the counting sort -------------------------------------------------------------------------------------------------------------------
// digits of list entry j of email e -> f(bucket, window, base index, negative)
template <class C, class Fn>
ZK_HD void zk_msm_digits_of(const ZkMsmArgsT<C>& A, u32 e, u32 j, Fn f) {
  const u32 i = A.sel_at(e, j);
  if (!A.sel && A.base_inf(i)) return;
  const Fr k = zk_msm_scalar(A, e, i);
  u32 carry = 0;
  for (u32 w = 0; w < A.K; ++w) {
    const int d = zk_msm_digit(k.l, w, A.c, carry);
    if (d) f((A.KS == 1u ? 0u : w) * A.nb + (u32)(d < 0 ? -d : d) - 1u, w, i, d < 0);
  }
}
template <class C>
ZK_HD void zk_msm_count_thread(const ZkMsmArgsT<C>& A, u32 e, u32 j) {
  if (j >= A.sel_count(e)) return;
  u32* cnt = A.count(e);
  zk_msm_digits_of(A, e, j, [&](u32 b, u32, u32, bool) { ZK_MSM_ATOMIC_ADD(&cnt[b], 1u); });
}
// one workgroup of `threads` threads per email (thread t of them): counts -> exclusive offsets in place, cursor = offsets; count[total] = entries.
// Two phases separated by a barrier the caller supplies (host mirror: phase 0 for every t, then phase 1 for every t).
template <class C>
ZK_HD void zk_msm_scan_thread(const ZkMsmArgsT<C>& A, u32 e, u32 t, u32 threads, u32* partial /*[threads + 1]*/, int phase) {
  u32* cnt = A.count(e);
  u32* cur = A.cursor(e);
  const u32 total = A.KS * A.nb, per = (total + threads - 1) / threads;
  const u32 lo = t * per < total ? t * per : total, hi = lo + per < total ? lo + per : total;
  if (phase == 0) {
    u32 s = 0;
    for (u32 k = lo; k < hi; ++k) s += cnt[k];
    partial[t + 1] = s;
    if (t == 0) partial[0] = 0;
    return;
  }
  u32 base = 0;
  for (u32 k = 0; k <= t; ++k) base += partial[k];      // (threads is small: 256 .. 1024)
  for (u32 k = lo; k < hi; ++k) { const u32 v = cnt[k]; cnt[k] = base; cur[k] = base; base += v; }
  if (hi == total && lo < total) cnt[total] = base;
}
template <class C>
ZK_HD void zk_msm_scatter_thread(const ZkMsmArgsT<C>& A, u32 e, u32 j) {
  if (j >= A.sel_count(e)) return;
  u32* cur = A.cursor(e);
  u32* ent = A.entry(e);
  zk_msm_digits_of(A, e, j, [&](u32 b, u32 w, u32 i, bool neg) { ent[ZK_MSM_ATOMIC_ADD(&cur[b], 1u)] = (w * A.stride + i) | (neg ? 0x80000000u : 0u); });
}
// The same with WORKGROUP-LOCAL histograms (round 5, profiles/r05/r05_u_msm0_kernel_stats.csv: for 2^21 full-size scalars the 33.5 M
// global atomics of count and again of scatter, all on 32,768 addresses, were 1.26 + 2.84 ms of a 9.0 ms sum).  Workgroup `wg` of `n_wg`
// owns the list entries [wg per, (wg + 1) per), per = ceil(list length / n_wg): it counts them into `hist` (LDS atomics), then
//   count:   adds its non-zero counters to the email's global ones                     (one global atomic per bucket and workgroup)
//   scatter: reserves hist[b] consecutive places of bucket b with ONE atomicAdd on cursor[b], leaves the first place in hist[b],
//            and walks its entries again, taking places from hist[b] by LDS atomics.
// Phases are separated by workgroup barriers the caller supplies (host mirror: every thread of a phase, then the next phase).
template <class C>
ZK_HD void zk_msm_sort_wg_thread(const ZkMsmArgsT<C>& A, u32 e, u32 wg, u32 n_wg, u32 t, u32 threads, u32* hist, int phase, bool scatter) {
  const u32 total = A.KS * A.nb, len = A.sel_count(e);
  const u32 per = (len + n_wg - 1u) / n_wg;
  const u32 lo = wg * per < len ? wg * per : len, hi = lo + per < len ? lo + per : len;
  if (phase == 0) {
    for (u32 b = t; b < total; b += threads) hist[b] = 0;
    return;
  }
  if (phase == 2) {
    if (lo == hi) return;
    u32* cnt = A.count(e);
    u32* cur = A.cursor(e);
    for (u32 b = t; b < total; b += threads) {
      const u32 v = hist[b];
      if (!v) continue;
      if (scatter) hist[b] = ZK_MSM_ATOMIC_ADD(&cur[b], v);
      else ZK_MSM_ATOMIC_ADD(&cnt[b], v);
    }
    return;
  }
  // phase 1: count; phase 3 (scatter): place
  u32* ent = A.entry(e);
  for (u32 j = lo + t; j < hi; j += threads)
    zk_msm_digits_of(A, e, j, [&](u32 b, u32 w, u32 i, bool neg) {
      if (phase == 1) ZK_MSM_ATOMIC_ADD(&hist[b], 1u);
      else ent[ZK_MSM_ATOMIC_ADD(&hist[b], 1u)] = (w * A.stride + i) | (neg ? 0x80000000u : 0u);
    });
}

// ---- sliced bucket sums ----------------------------------------------------------------------------------------------------------------
template <class C>
ZK_HD u32 zk_msm_slice_size(const ZkMsmArgsT<C>& A, int level) { return level == 0 ? A.s0 : ZK_MSM_S1; }
// slices of `level`: the items of bucket b are [in[b], in[b + 1]) -- entries for level 0 (in = count), the slices of the level below
// otherwise; out[b] = first slice of bucket b, out[total] = number of slices.  One workgroup per email, two phases around a barrier.
template <class C>
ZK_HD void zk_msm_slice_scan_thread(const ZkMsmArgsT<C>& A, u32 e, int level, u32 t, u32 threads, u32* partial, int phase) {
  const u32* in = level == 0 ? A.count(e) : A.soff(e, level - 1);
  u32* out = A.soff(e, level);
  const u32 S = zk_msm_slice_size(A, level);
  const u32 total = A.KS * A.nb, per = (total + threads - 1) / threads;
  const u32 lo = t * per < total ? t * per : total, hi = lo + per < total ? lo + per : total;
  if (phase == 0) {
    u32 s = 0;
    for (u32 k = lo; k < hi; ++k) s += (in[k + 1] - in[k] + S - 1u) / S;
    partial[t + 1] = s;
    if (t == 0) partial[0] = 0;
    return;
  }
  u32 base = 0;
  for (u32 k = 0; k <= t; ++k) base += partial[k];
  for (u32 k = lo; k < hi; ++k) { out[k] = base; base += (in[k + 1] - in[k] + S - 1u) / S; }
  if (hi == total && lo < total) out[total] = base;
}
template <class C>
ZK_HD u32 zk_msm_slice_count(const ZkMsmArgsT<C>& A, u32 e, int level) {
  const u32 n = A.soff(e, level)[A.KS * A.nb];
  return n < A.off.cap[level] ? n : A.off.cap[level];       // (the capacity is a proven bound)
}
// slice t of `level` of email e
// (LEVEL0 is a template argument so that the kernel of the mixed additions does not carry the register budget of the full ones)
template <class C, bool LEVEL0>
ZK_HD void zk_msm_slice_sum_thread(const ZkMsmArgsT<C>& A, u32 e, int level, u32 t, u32 h) {
  typedef typename ZkMsmArgsT<C>::X X;
  const u32 total = A.KS * A.nb;
  const u32* off = A.soff(e, level);
  // the bucket whose slices contain t: the last b with off[b] <= t
  u32 lo = 0, hi = total;
  while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (off[mid] <= t) lo = mid; else hi = mid; }
  const u32 b = lo, j = t - off[b], S = zk_msm_slice_size(A, level);
  const u32* in = level == 0 ? A.count(e) : A.soff(e, level - 1);
  const u32 first = in[b] + j * S, last = first + S < in[b + 1] ? first + S : in[b + 1];
  X acc = ec29_inf<typename C::F>();
  if (LEVEL0) {
    const u32* ent = A.entry(e);
    for (u32 k = first; k < last; ++k) {
      const u32 v = ent[k];
      acc = ec29_add_mixed<typename C::F>(acc, C::load(A.table + (v & 0x7fffffffu), h, (v >> 31) != 0));
    }
  } else {
    const X* items = A.part(e, level - 1);
    for (u32 k = first; k < last; ++k) acc = ec29_add<typename C::F>(acc, items[(u64)k * C::LANES + h]);
  }
  A.part(e, level)[(u64)t * C::LANES + h] = acc;
}
// bucket b = the sum of its slices of the last level
template <class C>
ZK_HD void zk_msm_bucket_join_thread(const ZkMsmArgsT<C>& A, u32 e, u32 b, u32 h) {
  typedef typename ZkMsmArgsT<C>::X X;
  if (b >= A.KS * A.nb) return;
  const u32* off = A.soff(e, 2);
  const X* items = A.part(e, 2);
  X acc = ec29_inf<typename C::F>();
  for (u32 k = off[b], end = off[b + 1]; k < end; ++k) acc = ec29_add<typename C::F>(acc, items[(u64)k * C::LANES + h]);
  A.bucket(e)[(u64)b * C::LANES + h] = acc;
}

// ---- the weighted bucket sum  sum_b (b + 1) bucket[b]  by BIT PLANES ------------------------------------------------------------------
// With T_j = sum of the buckets whose weight has bit j set, the sum is sum_j 2^j T_j: the c plane sums are PLAIN sums (PFAN-way joins: log
// depth, every level c rows wide), and one lane folds them with c - 1 doublings -- 66 dependent point operations at c = 16 where the tree
// of (plain, weighted) nodes of round 5's first version needed 135.
ZK_HD u32 zk_msm_plane_n0(u32 nb) { return (nb + ZK_MSM_PFAN - 1u) / ZK_MSM_PFAN; }
// level 0, item g = (w c + j) n0 + q: the buckets [q F, (q + 1) F) of set w whose weight b + 1 has bit j set
template <class C>
ZK_HD void zk_msm_plane0_thread(const ZkMsmArgsT<C>& A, u32 e, u32 g, u32 h) {
  typedef typename ZkMsmArgsT<C>::X X;
  const u32 n0 = zk_msm_plane_n0(A.nb);
  if (g >= A.KS * A.c * n0) return;
  const u32 row = g / n0, q = g - row * n0, w = row / A.c, j = row - w * A.c;
  const u32 lo = q * ZK_MSM_PFAN, hi = lo + ZK_MSM_PFAN < A.nb ? lo + ZK_MSM_PFAN : A.nb;
  const X* bk = A.bucket(e);
  X acc = ec29_inf<typename C::F>();
  for (u32 b = lo; b < hi; ++b)
    if (((b + 1u) >> j) & 1u) acc = ec29_add<typename C::F>(acc, bk[((u64)w * A.nb + b) * C::LANES + h]);
  A.node_s(e)[(u64)g * C::LANES + h] = acc;
}
// a join level: `rows` rows of n_in sums -> rows of n_out = ceil(n_in / F); item g = row n_out + q
template <class C>
ZK_HD void zk_msm_plane_join_thread(const typename ZkMsmArgsT<C>::X* in, u32 rows, u32 n_in, typename ZkMsmArgsT<C>::X* out, u32 g, u32 h) {
  typedef typename ZkMsmArgsT<C>::X X;
  const u32 n_out = (n_in + ZK_MSM_PFAN - 1u) / ZK_MSM_PFAN;
  if (g >= rows * n_out) return;
  const u32 row = g / n_out, q = g - row * n_out;
  const u32 lo = q * ZK_MSM_PFAN, hi = lo + ZK_MSM_PFAN < n_in ? lo + ZK_MSM_PFAN : n_in;
  const X* r = in + (u64)row * n_in * C::LANES;
  X acc = r[(u64)lo * C::LANES + h];
  for (u32 k = lo + 1; k < hi; ++k) acc = ec29_add<typename C::F>(acc, r[(u64)k * C::LANES + h]);
  out[(u64)g * C::LANES + h] = acc;
}
// window w from its c plane sums T[w c + j]
template <class C>
ZK_HD void zk_msm_plane_window_thread(const ZkMsmArgsT<C>& A, u32 e, const typename ZkMsmArgsT<C>::X* T, u32 w, u32 h) {
  typedef typename ZkMsmArgsT<C>::X X;
  if (w >= A.KS) return;
  X acc = T[((u64)w * A.c + A.c - 1u) * C::LANES + h];
  for (u32 j = A.c - 1u; j-- > 0;) acc = ec29_add<typename C::F>(ec29_dbl<typename C::F>(acc), T[((u64)w * A.c + j) * C::LANES + h]);
  A.window(e)[(u64)w * C::LANES + h] = acc;
}
// the sum of email e: windows (Horner over the classic layout's K sets; precomputed windows: nothing to combine) + the ones' sum
template <class C>
ZK_HD void zk_msm_combine_thread(const ZkMsmArgsT<C>& A, u32 e, u32 h, const typename ZkMsmArgsT<C>::X* ones_sum) {
  typedef typename ZkMsmArgsT<C>::X X;
  const X* win = A.window(e);
  X total = ec29_inf<typename C::F>();
  if (A.KS == 1u) total = win[h];
  else
    for (u32 w = A.K; w-- > 0;) {
      for (u32 s = 0; s < A.c; ++s) total = ec29_dbl<typename C::F>(total);
      total = ec29_add<typename C::F>(total, win[(u64)w * C::LANES + h]);
    }
  if (ones_sum) total = ec29_add<typename C::F>(total, ones_sum[h]);
  C::store_out(A.out + e, total, h);
}

// ---- tables (once per key) -----------------------------------------------------------------------------------------------------------------
// copy w of base i: ext[w n + i] = 2^(c w) base_i, affine, 2^261 form (canonical-word arithmetic of zkwg_g1.h / zkwg_g2.h: one inversion
// per copy, once per key); K = 1: just the conversion (classic layout)
template <class Aff, class Xyzz, class Add, class Dbl, class ToAffine, class ToTable>
ZK_HD void zk_msm_table_thread(const Aff* bases, Aff* ext, u32 n, u32 c, u32 K, u32 i, Xyzz inf, Add add_mixed, Dbl dbl, ToAffine to_affine, ToTable to_table) {
  if (i >= n) return;
  const Aff p = bases[i];
  ext[i] = to_table(p);
  Xyzz acc = add_mixed(inf, p);
  for (u32 w = 1; w < K; ++w) {
    for (u32 s = 0; s < c; ++s) acc = dbl(acc);
    ext[(size_t)w * n + i] = to_table(to_affine(acc));
  }
}
