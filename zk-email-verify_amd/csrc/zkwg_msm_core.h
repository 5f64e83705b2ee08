// G1 multi-exponentiation on the device: the per-thread bodies of the kernels of zkwg_kernels_msm.hip, shared with the host
// mirror of the CPU tests (tests/native/hosttest.cpp runs exactly these functions, thread by thread).  Bucket method with signed
// c-bit windows (zkwg_g1.h), organised around one counting sort per call:
//
//   zk_msm_count    one thread per scalar: leave Montgomery form (one product), cut into K signed digits, count[window * nb + |d| - 1]++
//   zk_msm_scan     exclusive prefix sums of the K * nb counters (one workgroup)
//   zk_msm_scatter  one thread per scalar: the same digits again; entry[cursor[bucket]++] = base index | sign << 31
//                   (the order inside a bucket depends on the atomics; the sum does not -- point addition is exact and commutative)
//   zk_msm_buckets  one thread per bucket: its run of entries, accumulated with g1_add_mixed
//   zk_msm_reduce   sum_b (b + 1) bucket[b] per window as a tree of 32-way groups: a node is (S, A) = (plain sum, weighted sum with
//                   weights 1 .. span); 32 nodes of span s join into A = sum A_i + s * sum_i i S_i (running sum from the top, log2 s
//                   doublings), S = sum S_i.  nb = 2^(c-1) buckets take ceil((c-1)/5) levels.
//   zk_msm_combine  one thread: Horner over the windows, c doublings each
//
// DRAFT (branch next/msm, round 4): written and mirrored on the CPU after the round's GPU budget was spent; it has not run on a GPU.
#pragma once
#include "zkwg_g1.h"
#include "zkwg_g2.h"

#define ZK_MSM_FAN 8u      // fan-in of the weighted bucket tree: a node's serial work is 3 additions per child; the levels are what a lone sum waits for
#define ZK_MSM_JOIN 8u     // fan-in of the joins of the ones' partial sums
#define ZK_MSM_ONES 8u     // bases per lane of zk_msm_ones (64 and 16-way joins until r05_s: 124 dependent additions for a 1.8 M-wire witness, 50 now)

// the group the sums run over: G1 (pi_a, pib1, pi_c, the H sum) or G2 (pi_b) -- same kernels, other point arithmetic
struct ZkCurveG1 {
  typedef G1Affine Affine; typedef G1Xyzz Xyzz;
  static ZK_HD Xyzz inf() { return g1_xyzz_inf(); }
  static ZK_HD bool is_inf(const Affine& p) { return g1_is_inf(p); }
  static ZK_HD Affine neg(const Affine& p) { return g1_neg(p); }
  static ZK_HD Xyzz add_mixed(const Xyzz& a, const Affine& p) { return g1_add_mixed(a, p); }
  static ZK_HD Xyzz add(const Xyzz& a, const Xyzz& b) { return g1_add(a, b); }
  static ZK_HD Xyzz dbl(const Xyzz& a) { return g1_dbl(a); }
};
struct ZkCurveG2 {
  typedef G2Affine Affine; typedef G2Xyzz Xyzz;
  static ZK_HD Xyzz inf() { return g2_xyzz_inf(); }
  static ZK_HD bool is_inf(const Affine& p) { return g2_is_inf(p); }
  static ZK_HD Affine neg(const Affine& p) { return g2_neg(p); }
  static ZK_HD Xyzz add_mixed(const Xyzz& a, const Affine& p) { return g2_add_mixed(a, p); }
  static ZK_HD Xyzz add(const Xyzz& a, const Xyzz& b) { return g2_add(a, b); }
  static ZK_HD Xyzz dbl(const Xyzz& a) { return g2_dbl(a); }
};

template <class C>
struct ZkMsmArgsT {
  typedef typename C::Affine G1Affine;
  typedef typename C::Xyzz G1Xyzz;
  const G1Affine* bases;      // n points, Montgomery form, (0, 0) = infinity
  const Fr* scalars;          // n scalars of this call
  u32 n, c, K, nb;            // points, window bits, windows, buckets per window = 2^(c-1)
  // Precomputed windows (KS = 1, stride = n): `bases` holds K copies of the points, copy w = 2^(c w) * base (zk_msm_shift_thread, once per
  // key), so digit d of window w selects copy w and ALL windows share one set of nb buckets: there is no Horner pass over the windows (254
  // dependent doublings on one lane: 3.8 ms of a 15 ms sum, profiles/r05/r05_i_prove_kernel_stats.csv) and one bucket tree instead of K.
  // Classic layout: KS = K bucket sets, stride = 0.  The memory is what 288 GB are for: 20 x 47 MB per witness-sized sum.
  u32 KS, stride;
  u32 scalars_mont;           // 1: the scalars are in Montgomery form (the H evaluations of zkwg_ntt_api.hip, a Montgomery witness)
  u32 lds_sort;               // 1: count / scatter with workgroup-local histograms (zk_msm_sort_wg_thread) when KS nb fits LDS
  u32 plane_sums;             // 1: the weighted bucket sum by bit planes (zk_msm_plane*: log depth); 0: the (S, A) tree of zk_msm_reduce
  u32 ones_apart;             // 1: scalars equal to 1 do not enter the buckets (a witness is mostly bits: they would all land in ONE
                              // bucket of window 0); their bases are summed by zk_msm_ones + the 64-way tree and added at the end
  G1Xyzz* ones;               // [2 x ceil(n / ZK_MSM_ONES)] tree scratch of the ones' sum (ping-pong halves); ones[0] = the sum at the end
  u32* count;                 // [K * nb + 1] counters, then exclusive offsets (zk_msm_scan)
  u32* cursor;                // [K * nb] running write positions of zk_msm_scatter
  u32* entry;                 // [n * K] base index | sign << 31, grouped by bucket
  G1Xyzz* bucket;             // [K * nb]
  // a bucket's run of entries is summed in SLICES, by several threads: witness scalars are bits, bytes and a few field elements, so a
  // handful of buckets (the common byte values) receive thousands of entries while most receive none; one thread per bucket took 58 ms
  // for a 736 k-wire witness against 16 ms for 2^20 random scalars (profiles/r05/r05_d_bench_prove.json).  Level 0: slices of
  // ZK_MSM_S0 entries (mixed additions of bases); levels 1, 2: slices of ZK_MSM_S1 partial sums of the level below; then one thread per
  // bucket joins what is left (one item unless the bucket held more than S0 S1 S1 entries).
  u32* soff[3];               // [K * nb + 1] per level: first slice of every bucket
  G1Xyzz* part[3];            // per level: the slices' sums
  u32 cap[3];                 // slices a level can hold (n K / S + K nb bounds it)
  G1Xyzz* node_s; G1Xyzz* node_a;   // bit-plane sums, ping-pong: node_s [KS c n0], node_a [KS c ceil(n0 / F)], n0 = ceil(nb / ZK_MSM_PFAN)
  G1Xyzz* window;             // [K] weighted bucket sums
  G1Xyzz* out;                // [1]
};
typedef ZkMsmArgsT<ZkCurveG1> ZkMsmArgs;

#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_MSM_ATOMIC_INC(p) atomicAdd((p), 1u)
#define ZK_MSM_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#else
#define ZK_MSM_ATOMIC_INC(p) ((*(p))++)
static inline u32 zk_msm_host_fetch_add(u32* p, u32 v) { const u32 o = *p; *p = o + v; return o; }
#define ZK_MSM_ATOMIC_ADD(p, v) zk_msm_host_fetch_add((p), (v))
#endif

// scalar i in standard form
template <class C>
ZK_HD Fr zk_msm_scalar(const ZkMsmArgsT<C>& A, u32 i) { return A.scalars_mont ? fr_from_mont(A.scalars[i]) : A.scalars[i]; }
ZK_HD bool zk_msm_is_one(const Fr& k) { return k.l[0] == 1 && (k.l[1] | k.l[2] | k.l[3]) == 0; }
// the bases whose scalar is 1, ZK_MSM_ONES per thread (ones_apart); then zk_msm_tree_thread joins ZK_MSM_JOIN partial sums per thread until one is left
template <class C>
ZK_HD void zk_msm_ones_thread(const ZkMsmArgsT<C>& A, u32 t) {
  const u32 lo = t * ZK_MSM_ONES, hi = lo + ZK_MSM_ONES < A.n ? lo + ZK_MSM_ONES : A.n;
  if (lo >= A.n) return;
  typename C::Xyzz acc = C::inf();
  for (u32 i = lo; i < hi; ++i)
    if (zk_msm_is_one(zk_msm_scalar(A, i))) acc = C::add_mixed(acc, A.bases[i]);
  A.ones[t] = acc;
}
template <class C>
ZK_HD void zk_msm_tree_thread_c(const typename C::Xyzz* in, u32 n_in, typename C::Xyzz* out, u32 t) {
  const u32 lo = t * ZK_MSM_JOIN, hi = lo + ZK_MSM_JOIN < n_in ? lo + ZK_MSM_JOIN : n_in;
  if (lo >= n_in) return;
  typename C::Xyzz acc = in[lo];
  for (u32 i = lo + 1; i < hi; ++i) acc = C::add(acc, in[i]);
  out[t] = acc;
}
template <class C>
ZK_HD void zk_msm_count_thread(const ZkMsmArgsT<C>& A, u32 i) {
  if (i >= A.n || C::is_inf(A.bases[i])) return;
  const Fr k = zk_msm_scalar(A, i);
  if (A.ones_apart && zk_msm_is_one(k)) return;
  u32 carry = 0;
  for (u32 w = 0; w < A.K; ++w) {
    const int d = zk_msm_digit(k.l, w, A.c, carry);
    if (d) ZK_MSM_ATOMIC_INC(&A.count[(A.KS == 1u ? 0u : w) * A.nb + (u32)(d < 0 ? -d : d) - 1u]);
  }
}
// one workgroup of `threads` threads (thread t of them): counts -> exclusive offsets in place, cursor = offsets; count[total] = entries.
// Two phases separated by a barrier the caller supplies (host mirror: phase 0 for every t, then phase 1 for every t).
template <class C>
ZK_HD void zk_msm_scan_thread(const ZkMsmArgsT<C>& A, u32 t, u32 threads, u32* partial /*[threads + 1]*/, int phase) {
  const u32 total = A.KS * A.nb, per = (total + threads - 1) / threads;
  const u32 lo = t * per < total ? t * per : total, hi = lo + per < total ? lo + per : total;
  if (phase == 0) {
    u32 s = 0;
    for (u32 k = lo; k < hi; ++k) s += A.count[k];
    partial[t + 1] = s;
    if (t == 0) partial[0] = 0;
    return;
  }
  u32 base = 0;
  for (u32 k = 0; k <= t; ++k) base += partial[k];      // (threads is small: 256 .. 1024)
  for (u32 k = lo; k < hi; ++k) { const u32 v = A.count[k]; A.count[k] = base; A.cursor[k] = base; base += v; }
  if (hi == total && lo < total) A.count[total] = base;
  if (total == 0 && t == 0) A.count[0] = 0;
}
template <class C>
ZK_HD void zk_msm_scatter_thread(const ZkMsmArgsT<C>& A, u32 i) {
  if (i >= A.n || C::is_inf(A.bases[i])) return;
  const Fr k = zk_msm_scalar(A, i);
  if (A.ones_apart && zk_msm_is_one(k)) return;
  u32 carry = 0;
  for (u32 w = 0; w < A.K; ++w) {
    const int d = zk_msm_digit(k.l, w, A.c, carry);
    if (!d) continue;
    const u32 b = (A.KS == 1u ? 0u : w) * A.nb + (u32)(d < 0 ? -d : d) - 1u;
    const u32 at = ZK_MSM_ATOMIC_INC(&A.cursor[b]);
    A.entry[at] = (w * A.stride + i) | (d < 0 ? 0x80000000u : 0u);
  }
}
// ---- the counting sort with WORKGROUP-LOCAL histograms (round 5, profiles/r05/r05_u_msm0_kernel_stats.csv: for 2^21 full-size scalars
// the 33.5 M global atomics of zk_msm_count and again of zk_msm_scatter, all on 32,768 addresses, were 1.26 + 2.84 ms of a 9.0 ms sum).
// When the bucket counters fit LDS (KS nb <= ZK_MSM_LDS_BUCKETS: one bucket set of up to 2^15 -- the precomputed-windows layout at
// c <= 16), workgroup `wg` owns the scalars [wg per_wg, (wg + 1) per_wg): it counts them into `hist` (LDS atomics), then
//   count:   adds its non-zero counters to A.count                                     (KS nb global atomics per workgroup, not one per digit)
//   scatter: reserves hist[b] consecutive places of bucket b with ONE atomicAdd on A.cursor[b], leaves the first place in hist[b],
//            and walks its scalars again, taking places from hist[b] by LDS atomics.
// Phases are separated by workgroup barriers the caller supplies (host mirror: every thread of a phase, then the next phase).
#define ZK_MSM_LDS_BUCKETS 32768u
template <class C>
ZK_HD void zk_msm_sort_wg_thread(const ZkMsmArgsT<C>& A, u32 wg, u32 per_wg, u32 t, u32 threads, u32* hist, int phase, bool scatter) {
  const u32 total = A.KS * A.nb;
  const u32 lo = wg * per_wg, hi = lo + per_wg < A.n ? lo + per_wg : A.n;
  if (phase == 0) {
    for (u32 b = t; b < total; b += threads) hist[b] = 0;
    return;
  }
  if (phase == 2) {
    for (u32 b = t; b < total; b += threads) {
      const u32 v = hist[b];
      if (!v) continue;
      if (scatter) hist[b] = ZK_MSM_ATOMIC_ADD(&A.cursor[b], v);
      else ZK_MSM_ATOMIC_ADD(&A.count[b], v);
    }
    return;
  }
  // phase 1: count; phase 3 (scatter): place
  for (u32 i = lo + t; i < hi; i += threads) {
    if (C::is_inf(A.bases[i])) continue;
    const Fr k = zk_msm_scalar(A, i);
    if (A.ones_apart && zk_msm_is_one(k)) continue;
    u32 carry = 0;
    for (u32 w = 0; w < A.K; ++w) {
      const int d = zk_msm_digit(k.l, w, A.c, carry);
      if (!d) continue;
      const u32 b = (A.KS == 1u ? 0u : w) * A.nb + (u32)(d < 0 ? -d : d) - 1u;
      if (phase == 1) ZK_MSM_ATOMIC_ADD(&hist[b], 1u);
      else A.entry[ZK_MSM_ATOMIC_ADD(&hist[b], 1u)] = (w * A.stride + i) | (d < 0 ? 0x80000000u : 0u);
    }
  }
}
ZK_HD u32 zk_msm_sort_per_wg(u32 n) { const u32 per = (n + 255u) / 256u; return per < 1024u ? 1024u : per; }   // about 256 workgroups

template <class C>
ZK_HD void zk_msm_bucket_thread(const ZkMsmArgsT<C>& A, u32 b) {
  if (b >= A.KS * A.nb) return;
  typename C::Xyzz acc = C::inf();
  for (u32 k = A.count[b], e = A.count[b + 1]; k < e; ++k) {
    const u32 v = A.entry[k];
    typename C::Affine p = A.bases[v & 0x7fffffffu];
    if (v >> 31) p = C::neg(p);
    acc = C::add_mixed(acc, p);
  }
  A.bucket[b] = acc;
}
// one level of the reduction tree.  Nodes of the level below: `n_in` per window with span `span` (weights 1 .. span inside a node);
// in_a == nullptr: the nodes are the buckets themselves (S = A = bucket, span 1).  Thread g builds node g of the level above
// (`n_out` = ceil(n_in / ZK_MSM_FAN) per window).  The last level (n_out == 1) writes the window's sum to A.window.
template <class C>
ZK_HD void zk_msm_reduce_thread(const ZkMsmArgsT<C>& A, u32 g, const typename C::Xyzz* in_s, const typename C::Xyzz* in_a, u32 n_in, u32 span, typename C::Xyzz* out_s, typename C::Xyzz* out_a) {
  const u32 n_out = (n_in + ZK_MSM_FAN - 1u) / ZK_MSM_FAN;
  if (g >= A.KS * n_out) return;
  const u32 w = g / n_out, q = g - w * n_out;
  const u32 lo = q * ZK_MSM_FAN, hi = lo + ZK_MSM_FAN < n_in ? lo + ZK_MSM_FAN : n_in;
  const typename C::Xyzz* S = in_s + (size_t)w * n_in;
  const typename C::Xyzz* Aw = (in_a ? in_a : in_s) + (size_t)w * n_in;
  // sum_i i S_i for i = 1 .. m - 1 (running sum from the top), sum S_i, sum A_i
  typename C::Xyzz run = C::inf(), acc = C::inf(), sum_a = C::inf();
  for (u32 k = hi; k-- > lo;) {
    sum_a = C::add(sum_a, Aw[k]);
    if (k > lo) { run = C::add(run, S[k]); acc = C::add(acc, run); }
  }
  const typename C::Xyzz sum_s = C::add(run, S[lo]);
  for (u32 s = span; s > 1; s >>= 1) acc = C::dbl(acc);            // span is a power of two
  const typename C::Xyzz node_a = C::add(sum_a, acc);
  if (n_out == 1) { A.window[w] = node_a; return; }
  out_s[(size_t)w * n_out + q] = sum_s;
  out_a[(size_t)w * n_out + q] = node_a;
}
// ---- the weighted bucket sum  sum_b (b + 1) bucket[b]  by BIT PLANES (round 5, after profiles/r05/r05_s_prove_kernel_stats.csv: the
// (S, A) tree above is five dependent levels of 21 additions + up to 12 doublings on ever fewer lanes -- 3.4 ms of a 4 ms G1 sum and
// 10.5 ms of a 16 ms G2 sum are that latency).  With T_j = sum of the buckets whose weight has bit j set, the sum is sum_j 2^j T_j:
// the c plane sums are PLAIN sums (ZK_MSM_PFAN-way joins: log depth, every level c times wider than the tree's), and one lane folds
// them with c - 1 doublings.  c/2 times the additions of the tree -- irrelevant next to n K mixed additions -- for 66 dependent
// point operations instead of 135 (c = 16).
#define ZK_MSM_PFAN 8u
ZK_HD u32 zk_msm_plane_n0(u32 nb) { return (nb + ZK_MSM_PFAN - 1u) / ZK_MSM_PFAN; }
// level 0, thread g = (w c + j) n0 + q: the buckets [q F, (q + 1) F) of set w whose weight b + 1 has bit j set
template <class C>
ZK_HD void zk_msm_plane0_thread(const ZkMsmArgsT<C>& A, u32 g, typename C::Xyzz* out) {
  const u32 n0 = zk_msm_plane_n0(A.nb);
  if (g >= A.KS * A.c * n0) return;
  const u32 row = g / n0, q = g - row * n0, w = row / A.c, j = row - w * A.c;
  const u32 lo = q * ZK_MSM_PFAN, hi = lo + ZK_MSM_PFAN < A.nb ? lo + ZK_MSM_PFAN : A.nb;
  typename C::Xyzz acc = C::inf();
  for (u32 b = lo; b < hi; ++b)
    if (((b + 1u) >> j) & 1u) acc = C::add(acc, A.bucket[(size_t)w * A.nb + b]);
  out[g] = acc;
}
// a join level: `rows` rows of n_in sums -> rows of n_out = ceil(n_in / F); thread g = row n_out + q
template <class C>
ZK_HD void zk_msm_plane_join_thread(const typename C::Xyzz* in, u32 rows, u32 n_in, typename C::Xyzz* out, u32 g) {
  const u32 n_out = (n_in + ZK_MSM_PFAN - 1u) / ZK_MSM_PFAN;
  if (g >= rows * n_out) return;
  const u32 row = g / n_out, q = g - row * n_out;
  const u32 lo = q * ZK_MSM_PFAN, hi = lo + ZK_MSM_PFAN < n_in ? lo + ZK_MSM_PFAN : n_in;
  const typename C::Xyzz* r = in + (size_t)row * n_in;
  typename C::Xyzz acc = r[lo];
  for (u32 k = lo + 1; k < hi; ++k) acc = C::add(acc, r[k]);
  out[g] = acc;
}
// window w from its c plane sums T[w c + j]
template <class C>
ZK_HD void zk_msm_plane_window_thread(const ZkMsmArgsT<C>& A, const typename C::Xyzz* T, u32 w) {
  if (w >= A.KS) return;
  typename C::Xyzz acc = T[(size_t)w * A.c + A.c - 1u];
  for (u32 j = A.c - 1u; j-- > 0;) acc = C::add(C::dbl(acc), T[(size_t)w * A.c + j]);
  A.window[w] = acc;
}
template <class C>
ZK_HD void zk_msm_combine_thread(const ZkMsmArgsT<C>& A) {
  typename C::Xyzz total = C::inf();
  if (A.KS == 1u) total = A.window[0];        // precomputed windows: nothing to combine
  else
    for (u32 w = A.K; w-- > 0;) {
      for (u32 s = 0; s < A.c; ++s) total = C::dbl(total);
      total = C::add(total, A.window[w]);
    }
  if (A.ones_apart) total = C::add(total, A.ones[0]);
  A.out[0] = total;
}
// copy w of base i for the precomputed-windows layout: ext[w n + i] = 2^(c w) base_i, affine (one inversion per copy; once per key)
template <class C, class ToAffine>
ZK_HD void zk_msm_shift_thread(const typename C::Affine* bases, typename C::Affine* ext, u32 n, u32 c, u32 K, u32 i, ToAffine to_affine) {
  if (i >= n) return;
  const typename C::Affine p = bases[i];
  ext[i] = p;
  typename C::Xyzz acc = C::add_mixed(C::inf(), p);
  for (u32 w = 1; w < K; ++w) {
    for (u32 s = 0; s < c; ++s) acc = C::dbl(acc);
    ext[(size_t)w * n + i] = to_affine(acc);
  }
}
#define ZK_MSM_S0 16u      // (64 / 32 / 32 until r05_s: 128 dependent additions in front of a bucket; 32 now, for 4 x the slices of level 0)
#define ZK_MSM_S1 8u
ZK_HD u32 zk_msm_slice_size(int level) { return level == 0 ? ZK_MSM_S0 : ZK_MSM_S1; }
// slices of `level`: the items of bucket b are [in[b], in[b + 1]) -- entries for level 0 (in = A.count), the slices of the level below
// otherwise; out[b] = first slice of bucket b, out[total] = number of slices.  One workgroup, two phases around a barrier (as zk_msm_scan).
template <class C>
ZK_HD void zk_msm_slice_scan_thread(const ZkMsmArgsT<C>& A, int level, u32 t, u32 threads, u32* partial, int phase) {
  const u32* in = level == 0 ? A.count : A.soff[level - 1];
  u32* out = A.soff[level];
  const u32 S = zk_msm_slice_size(level);
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
  if (total == 0 && t == 0) out[0] = 0;
}
// slice t of `level` (nothing beyond the level's slice count or capacity: the capacity is a proven bound)
template <class C>
ZK_HD void zk_msm_slice_sum_thread(const ZkMsmArgsT<C>& A, int level, u32 t) {
  const u32 total = A.KS * A.nb;
  const u32* off = A.soff[level];
  if (t >= off[total] || t >= A.cap[level]) return;
  // the bucket whose slices contain t: the last b with off[b] <= t
  u32 lo = 0, hi = total;
  while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (off[mid] <= t) lo = mid; else hi = mid; }
  const u32 b = lo, j = t - off[b], S = zk_msm_slice_size(level);
  const u32* in = level == 0 ? A.count : A.soff[level - 1];
  const u32 first = in[b] + j * S, last = first + S < in[b + 1] ? first + S : in[b + 1];
  typename C::Xyzz acc = C::inf();
  if (level == 0) {
    for (u32 k = first; k < last; ++k) {
      const u32 v = A.entry[k];
      typename C::Affine p = A.bases[v & 0x7fffffffu];
      if (v >> 31) p = C::neg(p);
      acc = C::add_mixed(acc, p);
    }
  } else {
    const typename C::Xyzz* items = A.part[level - 1];
    for (u32 k = first; k < last; ++k) acc = C::add(acc, items[k]);
  }
  A.part[level][t] = acc;
}
// bucket b = the sum of its slices of the last level
template <class C>
ZK_HD void zk_msm_bucket_join_thread(const ZkMsmArgsT<C>& A, u32 b) {
  if (b >= A.KS * A.nb) return;
  typename C::Xyzz acc = C::inf();
  for (u32 k = A.soff[2][b], e = A.soff[2][b + 1]; k < e; ++k) acc = C::add(acc, A.part[2][k]);
  A.bucket[b] = acc;
}
// (G1 by name: the host mirror of the CPU tests)
ZK_HD void zk_msm_tree_thread(const G1Xyzz* in, u32 n_in, G1Xyzz* out, u32 t) { zk_msm_tree_thread_c<ZkCurveG1>(in, n_in, out, t); }
