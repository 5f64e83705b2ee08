// Multi-exponentiation kernels over BN254 G1 and G2 (bodies: zkwg_msm_core.h, shared with the host mirror of the CPU tests).
// Every kernel runs E emails at once: blockIdx.y = email, blockIdx.x strides over the email's items (their number is only known on the
// device: list lengths, slice counts).  G2 runs on lane pairs (zkwg_ec29.h): item = thread / 2, half = thread & 1.
#include "zkwg_dev.h"
#include "zkwg_msm_core.h"

#define ZK_MSM_GRID 16384u     // wavefronts per SERIES of the grid-stride kernels (shared by its emails; 1,024 SIMDs x 3 .. 8 wavefronts each)

// One workgroup classifies ZK_CLS_ROWS x 256 scalars of one email: row j = the scalars base + 256 j + thread (coalesced).  A row's
// members of list l get their ranks from a ballot; the 4 wavefronts x ROWS row counts are summed once per list, ONE global atomic per
// list reserves the workgroup's places (one atomic per wavefront and list, all on the same two counters per email, made this kernel
// 6 ms per 8 x 2^21 scalars -- profiles/r06/r06_d_msm_kernel_stats_wit21_e8_c13.csv -- where reading them takes 0.15 ms).
#define ZK_CLS_ROWS 16u
__global__ __launch_bounds__(256) void zk_msm_classify(ZkClassifyArgs A) {
  __shared__ u32 cnt[6][4 * ZK_CLS_ROWS + 1];          // per list: counts of (row, wavefront), then their exclusive offsets; [..][64] = the workgroup's base
  const u32 e = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, nl = 2 * A.n_targets;
  for (u32 base = blockIdx.x * (256u * ZK_CLS_ROWS); base < A.n; base += gridDim.x * (256u * ZK_CLS_ROWS)) {
    u32 codes[ZK_CLS_ROWS];
#pragma unroll
    for (u32 j = 0; j < ZK_CLS_ROWS; ++j) codes[j] = zk_msm_classify_code(A, e, base + 256u * j + threadIdx.x);
#pragma unroll
    for (u32 j = 0; j < ZK_CLS_ROWS; ++j)
      for (u32 l = 0; l < nl; ++l) {
        const u64 m = __ballot((codes[j] >> l) & 1u);
        if (lane == 0) cnt[l][4 * j + wave] = (u32)__builtin_popcountll(m);
      }
    __syncthreads();
    if (threadIdx.x < nl) {
      const u32 l = threadIdx.x;
      u32 s = 0;
      for (u32 q = 0; q < 4 * ZK_CLS_ROWS; ++q) { const u32 v = cnt[l][q]; cnt[l][q] = s; s += v; }
      cnt[l][4 * ZK_CLS_ROWS] = s ? atomicAdd(zk_msm_classify_counter(A, e, l), s) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (u32 j = 0; j < ZK_CLS_ROWS; ++j)
      for (u32 l = 0; l < nl; ++l) {
        const bool mine = (codes[j] >> l) & 1u;
        const u64 m = __ballot(mine);
        if (mine) zk_msm_classify_list(A, e, l)[cnt[l][4 * ZK_CLS_ROWS] + cnt[l][4 * j + wave] + (u32)__builtin_popcountll(m & ((1ull << lane) - 1ull))] =
            base + 256u * j + threadIdx.x - A.t[l >> 1].first;
      }
    __syncthreads();
  }
}
template <class C> __global__ __launch_bounds__(256) void zk_msm_count(ZkMsmArgsT<C> A) {
  const u32 e = blockIdx.y, len = A.sel_count(e);
  for (u32 j = blockIdx.x * 256u + threadIdx.x; j < len; j += gridDim.x * 256u) zk_msm_count_thread(A, e, j);
}
template <class C> __global__ __launch_bounds__(1024) void zk_msm_scan(ZkMsmArgsT<C> A) {
  __shared__ u32 partial[1025];
  zk_msm_scan_thread(A, blockIdx.y, threadIdx.x, 1024u, partial, 0);
  __syncthreads();
  zk_msm_scan_thread(A, blockIdx.y, threadIdx.x, 1024u, partial, 1);
}
template <class C> __global__ __launch_bounds__(256) void zk_msm_scatter(ZkMsmArgsT<C> A) {
  const u32 e = blockIdx.y, len = A.sel_count(e);
  for (u32 j = blockIdx.x * 256u + threadIdx.x; j < len; j += gridDim.x * 256u) zk_msm_scatter_thread(A, e, j);
}
// count (SCATTER = false) / scatter (true) with the workgroup's histogram in LDS: 128 KB, one workgroup of 1,024 lanes per CU
template <class C, bool SCATTER> __global__ __launch_bounds__(1024) void zk_msm_sort_wg(ZkMsmArgsT<C> A) {
  __shared__ u32 hist[ZK_MSM_LDS_BUCKETS];
  for (int phase = 0; phase < (SCATTER ? 4 : 3); ++phase) {
    zk_msm_sort_wg_thread(A, blockIdx.y, blockIdx.x, gridDim.x, threadIdx.x, 1024u, hist, phase, SCATTER);
    __syncthreads();
  }
}
template <class C> __global__ __launch_bounds__(1024) void zk_msm_slice_scan(ZkMsmArgsT<C> A, int level) {
  __shared__ u32 partial[1025];
  zk_msm_slice_scan_thread(A, blockIdx.y, level, threadIdx.x, 1024u, partial, 0);
  __syncthreads();
  zk_msm_slice_scan_thread(A, blockIdx.y, level, threadIdx.x, 1024u, partial, 1);
}
template <class C, bool LEVEL0> __global__ __launch_bounds__(64) void zk_msm_slice_sum(ZkMsmArgsT<C> A, int level) {
  const u32 e = blockIdx.y, n = zk_msm_slice_count(A, e, level);
  constexpr u32 per = 64u / C::LANES;
  const u32 h = threadIdx.x % C::LANES;
  for (u32 t = blockIdx.x * per + threadIdx.x / C::LANES; t < n; t += gridDim.x * per) zk_msm_slice_sum_thread<C, LEVEL0>(A, e, level, t, h);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_bucket_join(ZkMsmArgsT<C> A) {
  constexpr u32 per = 64u / C::LANES;
  zk_msm_bucket_join_thread(A, blockIdx.y, blockIdx.x * per + threadIdx.x / C::LANES, threadIdx.x % C::LANES);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_plane0(ZkMsmArgsT<C> A) {
  constexpr u32 per = 64u / C::LANES;
  zk_msm_plane0_thread(A, blockIdx.y, blockIdx.x * per + threadIdx.x / C::LANES, threadIdx.x % C::LANES);
}
// join level of the bit planes: in / out = node_s / node_a of the email (flip = 1: the other way)
template <class C> __global__ __launch_bounds__(64) void zk_msm_plane_join(ZkMsmArgsT<C> A, u32 rows, u32 n_in, u32 flip) {
  constexpr u32 per = 64u / C::LANES;
  const u32 e = blockIdx.y;
  zk_msm_plane_join_thread<C>(flip ? A.node_a(e) : A.node_s(e), rows, n_in, flip ? A.node_s(e) : A.node_a(e), blockIdx.x * per + threadIdx.x / C::LANES, threadIdx.x % C::LANES);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_plane_window(ZkMsmArgsT<C> A, u32 flip) {
  constexpr u32 per = 64u / C::LANES;
  const u32 e = blockIdx.y;
  zk_msm_plane_window_thread(A, e, flip ? A.node_a(e) : A.node_s(e), blockIdx.x * per + threadIdx.x / C::LANES, threadIdx.x % C::LANES);
}
// the ones' partial sums live in the two halves of the email's `ones` area (half = ceil(n / ONES) sums), levels alternate
template <class C> __global__ __launch_bounds__(64) void zk_msm_ones(ZkMsmArgsT<C> A, u32 half, u32 into_second) {
  constexpr u32 per = 64u / C::LANES;
  const u32 e = blockIdx.y, n = zk_msm_ones_parts(A.n_ones[e]);
  typename ZkMsmArgsT<C>::X* out = A.ones_acc(e) + (into_second ? (u64)half * C::LANES : 0);
  const u32 h = threadIdx.x % C::LANES;
  for (u32 t = blockIdx.x * per + threadIdx.x / C::LANES; t < n; t += gridDim.x * per) zk_msm_ones_thread(A, e, t, h, out);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_tree(ZkMsmArgsT<C> A, u32 half, u32 level, u32 from_second) {
  constexpr u32 per = 64u / C::LANES;
  const u32 e = blockIdx.y;
  typename ZkMsmArgsT<C>::X* lo = A.ones_acc(e);
  typename ZkMsmArgsT<C>::X* hi = lo + (u64)half * C::LANES;
  const u32 n = (zk_msm_ones_level_count(A.n_ones[e], level) + ZK_MSM_JOIN - 1u) / ZK_MSM_JOIN;
  const u32 h = threadIdx.x % C::LANES;
  for (u32 t = blockIdx.x * per + threadIdx.x / C::LANES; t < n; t += gridDim.x * per) zk_msm_tree_thread(A, e, level, t, h, from_second ? hi : lo, from_second ? lo : hi);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_combine(ZkMsmArgsT<C> A, u32 half, u32 ones_in_second) {
  constexpr u32 per = 64u / C::LANES;
  const u32 e = blockIdx.x * per + threadIdx.x / C::LANES;
  if (e >= A.E) return;
  const typename ZkMsmArgsT<C>::X* ones = A.ones ? A.ones_acc(e) + (ones_in_second ? (u64)half * C::LANES : 0) : nullptr;
  zk_msm_combine_thread(A, e, threadIdx.x % C::LANES, ones);
}

static inline u32 zk_grid(u64 items, u32 per_wg, u32 cap) { const u64 g = (items + per_wg - 1) / per_wg; return (u32)(g < 1 ? 1 : g > cap ? cap : g); }

// the launch series of E multi-exponentiations over one base set on `st` (the counters are zeroed here)
template <class C>
static void zk_msm_launch_t(const ZkMsmArgsT<C>& A, hipStream_t st) {
  constexpr u32 per = 64u / C::DEV_LANES;     // points per wavefront (host code: C::LANES is the CPU mirror's 1 here)
  const u32 total = A.KS * A.nb, E = A.E;
  for (u32 e = 0; e < E; ++e) hipMemsetAsync(A.count(e), 0, ((size_t)total + 1) * 4, st);
  const bool lds_sort = A.lds_sort && total <= ZK_MSM_LDS_BUCKETS;
  const u32 n_wg = zk_grid(A.n, 8192u, 256u);                 // at least 8 list entries per lane of a sorting workgroup at full length
  if (lds_sort) hipLaunchKernelGGL((zk_msm_sort_wg<C, false>), dim3(n_wg, E), dim3(1024), 0, st, A);
  else hipLaunchKernelGGL(zk_msm_count<C>, dim3(zk_grid(A.n, 256u, 4096u), E), dim3(256), 0, st, A);
  hipLaunchKernelGGL(zk_msm_scan<C>, dim3(1, E), dim3(1024), 0, st, A);
  if (lds_sort) hipLaunchKernelGGL((zk_msm_sort_wg<C, true>), dim3(n_wg, E), dim3(1024), 0, st, A);
  else hipLaunchKernelGGL(zk_msm_scatter<C>, dim3(zk_grid(A.n, 256u, 4096u), E), dim3(256), 0, st, A);
  for (int level = 0; level < 3; ++level) {
    hipLaunchKernelGGL(zk_msm_slice_scan<C>, dim3(1, E), dim3(1024), 0, st, A, level);
    if (level == 0) hipLaunchKernelGGL((zk_msm_slice_sum<C, true>), dim3(zk_grid(A.off.cap[level], per, (ZK_MSM_GRID + E - 1) / E), E), dim3(64), 0, st, A, level);
    else hipLaunchKernelGGL((zk_msm_slice_sum<C, false>), dim3(zk_grid(A.off.cap[level], per, (ZK_MSM_GRID + E - 1) / E), E), dim3(64), 0, st, A, level);
  }
  hipLaunchKernelGGL(zk_msm_bucket_join<C>, dim3((total + per - 1) / per, E), dim3(64), 0, st, A);
  // sum_b (b + 1) bucket[b] by bit planes: level 0 into node_s, joins alternate node_a / node_s, one lane per window folds
  const u32 rows = A.KS * A.c;
  u32 n_in = zk_msm_plane_n0(A.nb), flip = 0;
  hipLaunchKernelGGL(zk_msm_plane0<C>, dim3((rows * n_in + per - 1) / per, E), dim3(64), 0, st, A);
  while (n_in > 1) {
    const u32 n_out = (n_in + ZK_MSM_PFAN - 1) / ZK_MSM_PFAN;
    hipLaunchKernelGGL(zk_msm_plane_join<C>, dim3((rows * n_out + per - 1) / per, E), dim3(64), 0, st, A, rows, n_in, flip);
    flip ^= 1u; n_in = n_out;
  }
  hipLaunchKernelGGL(zk_msm_plane_window<C>, dim3((A.KS + per - 1) / per, E), dim3(64), 0, st, A, flip);
  u32 half = 0, in_second = 0;
  if (A.ones) {
    half = (A.n + ZK_MSM_ONES - 1) / ZK_MSM_ONES;
    hipLaunchKernelGGL(zk_msm_ones<C>, dim3(zk_grid(half, per, (ZK_MSM_GRID + E - 1) / E), E), dim3(64), 0, st, A, half, 0u);
    u32 m = half, level = 0;
    while (m > 1) {
      const u32 m2 = (m + ZK_MSM_JOIN - 1) / ZK_MSM_JOIN;
      hipLaunchKernelGGL(zk_msm_tree<C>, dim3(zk_grid(m2, per, (ZK_MSM_GRID + E - 1) / E), E), dim3(64), 0, st, A, half, level, in_second);
      in_second ^= 1u; m = m2; ++level;
    }
  }
  hipLaunchKernelGGL(zk_msm_combine<C>, dim3((E + per - 1) / per), dim3(64), 0, st, A, half, in_second);
}
void zk_msm_launch_g1(const ZkMsmArgsT<ZkEcG1>& A, hipStream_t st) { zk_msm_launch_t<ZkEcG1>(A, st); }
void zk_msm_launch_g2(const ZkMsmArgsT<ZkEcG2>& A, hipStream_t st) { zk_msm_launch_t<ZkEcG2>(A, st); }
void zk_msm_classify_launch(const ZkClassifyArgs& A, hipStream_t st) {
  hipLaunchKernelGGL(zk_msm_classify, dim3(zk_grid(A.n, 256u * ZK_CLS_ROWS, 4096u), A.E), dim3(256), 0, st, A);
}

// ---- tables: K shifted copies of the bases in 2^261 form (once per key), and the bases' infinity bits -----------------------------------
template <int G> __global__ __launch_bounds__(64) void zk_msm_table(const void* bases, void* ext, u32 n, u32 c, u32 K) {
  const u32 i = blockIdx.x * 64u + threadIdx.x;
  if constexpr (G == 1)
    zk_msm_table_thread((const G1Affine*)bases, (G1Affine*)ext, n, c, K, i, g1_xyzz_inf(), [](const G1Xyzz& a, const G1Affine& p) { return g1_add_mixed(a, p); },
                        [](const G1Xyzz& a) { return g1_dbl(a); }, [](const G1Xyzz& a) { return g1_to_affine(a); }, [](const G1Affine& p) { return zk_g1_to_table_form(p); });
  else
    zk_msm_table_thread((const G2Affine*)bases, (G2Affine*)ext, n, c, K, i, g2_xyzz_inf(), [](const G2Xyzz& a, const G2Affine& p) { return g2_add_mixed(a, p); },
                        [](const G2Xyzz& a) { return g2_dbl(a); }, [](const G2Xyzz& a) { return g2_to_affine(a); }, [](const G2Affine& p) { return zk_g2_to_table_form(p); });
}
template <int G> __global__ __launch_bounds__(256) void zk_msm_inf_bits(const void* bases, u32* bits, u32 n) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  bool inf = false;
  if (i < n) {
    if constexpr (G == 1) inf = g1_is_inf(((const G1Affine*)bases)[i]);
    else inf = g2_is_inf(((const G2Affine*)bases)[i]);
  }
  const u64 m = __ballot(inf);
  if ((threadIdx.x & 63u) == 0 && i < n) { bits[i >> 5] = (u32)m; if (i + 32 < ((n + 31u) & ~31u)) bits[(i >> 5) + 1] = (u32)(m >> 32); }
}
void zk_msm_table_launch(int group, const void* bases, void* ext, u32* inf_bits, u32 n, u32 c, u32 K, hipStream_t st) {
  if (group == 1) {
    hipLaunchKernelGGL(zk_msm_inf_bits<1>, dim3((n + 255) / 256), dim3(256), 0, st, bases, inf_bits, n);
    hipLaunchKernelGGL(zk_msm_table<1>, dim3((n + 63) / 64), dim3(64), 0, st, bases, ext, n, c, K);
  } else {
    hipLaunchKernelGGL(zk_msm_inf_bits<2>, dim3((n + 255) / 256), dim3(256), 0, st, bases, inf_bits, n);
    hipLaunchKernelGGL(zk_msm_table<2>, dim3((n + 63) / 64), dim3(64), 0, st, bases, ext, n, c, K);
  }
}

// ---- fixed-base multiples: out[i] = k_i G for the group's generator -- how a key with a known trapdoor is turned into bases
// (tests, tools: oracle/pyref/groth16.py makes the scalars).  One thread per scalar: double-and-add from the top over a 4-bit
// window table of the generator held in LDS (canonical-word arithmetic: a tool's kernel, not the prover's).
template <int G, class Aff, class Xyzz>
__global__ __launch_bounds__(64) void zk_fixed_base(const Aff gen, const Fr* __restrict__ k, Aff* __restrict__ out, u32 n) {
  __shared__ Xyzz tab[16];
  auto add = [](const Xyzz& a, const Xyzz& b) { if constexpr (G == 1) return g1_add(a, b); else return g2_add(a, b); };
  auto dbl = [](const Xyzz& a) { if constexpr (G == 1) return g1_dbl(a); else return g2_dbl(a); };
  if (threadIdx.x == 0) {
    if constexpr (G == 1) { tab[0] = g1_xyzz_inf(); for (int j = 1; j < 16; ++j) tab[j] = g1_add_mixed(tab[j - 1], gen); }
    else { tab[0] = g2_xyzz_inf(); for (int j = 1; j < 16; ++j) tab[j] = g2_add_mixed(tab[j - 1], gen); }
  }
  __syncthreads();
  const u32 i = blockIdx.x * 64u + threadIdx.x;
  if (i >= n) return;
  const Fr s = k[i];
  Xyzz acc = tab[0];
  for (int w = 63; w >= 0; --w) {
    acc = dbl(dbl(dbl(dbl(acc))));
    const u32 d = (u32)(s.l[w >> 4] >> (4 * (w & 15))) & 15u;
    if (d) acc = add(acc, tab[d]);
  }
  if constexpr (G == 1) out[i] = g1_to_affine(acc);
  else out[i] = g2_to_affine(acc);
}
void zk_fixed_base_g1_launch(const G1Affine& gen, const Fr* k, G1Affine* out, u32 n, hipStream_t st) {
  hipLaunchKernelGGL((zk_fixed_base<1, G1Affine, G1Xyzz>), dim3((n + 63) / 64), dim3(64), 0, st, gen, k, out, n);
}
void zk_fixed_base_g2_launch(const G2Affine& gen, const Fr* k, G2Affine* out, u32 n, hipStream_t st) {
  hipLaunchKernelGGL((zk_fixed_base<2, G2Affine, G2Xyzz>), dim3((n + 63) / 64), dim3(64), 0, st, gen, k, out, n);
}
