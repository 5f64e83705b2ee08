// Multi-exponentiation kernels over BN254 G1 and G2 (bodies: zkwg_msm_core.h, shared with the host mirror of the CPU tests).
// First run on a GPU in round 5 (profiles/r05/r05_a_msm_*): bit-exact against the oracle up to n = 70,000, linear at n = 2^18.
#include "zkwg_dev.h"
#include "zkwg_msm_core.h"

template <class C> __global__ __launch_bounds__(256) void zk_msm_count(ZkMsmArgsT<C> A) { zk_msm_count_thread(A, blockIdx.x * 256u + threadIdx.x); }
template <class C> __global__ __launch_bounds__(1024) void zk_msm_scan(ZkMsmArgsT<C> A) {
  __shared__ u32 partial[1025];
  zk_msm_scan_thread(A, threadIdx.x, 1024u, partial, 0);
  __syncthreads();
  zk_msm_scan_thread(A, threadIdx.x, 1024u, partial, 1);
}
template <class C> __global__ __launch_bounds__(256) void zk_msm_scatter(ZkMsmArgsT<C> A) { zk_msm_scatter_thread(A, blockIdx.x * 256u + threadIdx.x); }
// count (SCATTER = false) / scatter (true) with the workgroup's histogram in LDS: 128 KB, one workgroup of 1,024 lanes per CU
template <class C, bool SCATTER> __global__ __launch_bounds__(1024) void zk_msm_sort_wg(ZkMsmArgsT<C> A, u32 per_wg) {
  __shared__ u32 hist[ZK_MSM_LDS_BUCKETS];
  for (int phase = 0; phase < (SCATTER ? 4 : 3); ++phase) {
    zk_msm_sort_wg_thread(A, blockIdx.x, per_wg, threadIdx.x, 1024u, hist, phase, SCATTER);
    __syncthreads();
  }
}
template <class C> __global__ __launch_bounds__(1024) void zk_msm_slice_scan(ZkMsmArgsT<C> A, int level) {
  __shared__ u32 partial[1025];
  zk_msm_slice_scan_thread(A, level, threadIdx.x, 1024u, partial, 0);
  __syncthreads();
  zk_msm_slice_scan_thread(A, level, threadIdx.x, 1024u, partial, 1);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_slice_sum(ZkMsmArgsT<C> A, int level) { zk_msm_slice_sum_thread(A, level, blockIdx.x * 64u + threadIdx.x); }
template <class C> __global__ __launch_bounds__(64) void zk_msm_bucket_join(ZkMsmArgsT<C> A) { zk_msm_bucket_join_thread(A, blockIdx.x * 64u + threadIdx.x); }
template <class C> __global__ __launch_bounds__(64) void zk_msm_reduce(ZkMsmArgsT<C> A, const typename C::Xyzz* in_s, const typename C::Xyzz* in_a, u32 n_in, u32 span,
                                                                        typename C::Xyzz* out_s, typename C::Xyzz* out_a) {
  zk_msm_reduce_thread(A, blockIdx.x * 64u + threadIdx.x, in_s, in_a, n_in, span, out_s, out_a);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_plane0(ZkMsmArgsT<C> A, typename C::Xyzz* out) { zk_msm_plane0_thread(A, blockIdx.x * 64u + threadIdx.x, out); }
template <class C> __global__ __launch_bounds__(64) void zk_msm_plane_join(const typename C::Xyzz* in, u32 rows, u32 n_in, typename C::Xyzz* out) {
  zk_msm_plane_join_thread<C>(in, rows, n_in, out, blockIdx.x * 64u + threadIdx.x);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_plane_window(ZkMsmArgsT<C> A, const typename C::Xyzz* T) { zk_msm_plane_window_thread(A, T, blockIdx.x * 64u + threadIdx.x); }
template <class C> __global__ __launch_bounds__(64) void zk_msm_ones(ZkMsmArgsT<C> A) { zk_msm_ones_thread(A, blockIdx.x * 64u + threadIdx.x); }
template <class C> __global__ __launch_bounds__(64) void zk_msm_tree(const typename C::Xyzz* in, u32 n_in, typename C::Xyzz* out) {
  zk_msm_tree_thread_c<C>(in, n_in, out, blockIdx.x * 64u + threadIdx.x);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_combine(ZkMsmArgsT<C> A) { if (threadIdx.x == 0 && blockIdx.x == 0) zk_msm_combine_thread(A); }

// launches of one multi-exponentiation on `st` (A.count zeroed here)
template <class C>
static void zk_msm_launch_t(const ZkMsmArgsT<C>& A, hipStream_t st) {
  typedef typename C::Xyzz X;
  const u32 total = A.KS * A.nb;
  hipMemsetAsync(A.count, 0, ((size_t)total + 1) * 4, st);
  const bool lds_sort = A.lds_sort && total <= ZK_MSM_LDS_BUCKETS;
  const u32 per_wg = zk_msm_sort_per_wg(A.n), n_wg = (A.n + per_wg - 1) / per_wg;
  if (lds_sort) hipLaunchKernelGGL((zk_msm_sort_wg<C, false>), dim3(n_wg), dim3(1024), 0, st, A, per_wg);
  else hipLaunchKernelGGL(zk_msm_count<C>, dim3((A.n + 255) / 256), dim3(256), 0, st, A);
  hipLaunchKernelGGL(zk_msm_scan<C>, dim3(1), dim3(1024), 0, st, A);
  if (lds_sort) hipLaunchKernelGGL((zk_msm_sort_wg<C, true>), dim3(n_wg), dim3(1024), 0, st, A, per_wg);
  else hipLaunchKernelGGL(zk_msm_scatter<C>, dim3((A.n + 255) / 256), dim3(256), 0, st, A);
  for (int level = 0; level < 3; ++level) {
    hipLaunchKernelGGL(zk_msm_slice_scan<C>, dim3(1), dim3(1024), 0, st, A, level);
    hipLaunchKernelGGL(zk_msm_slice_sum<C>, dim3((A.cap[level] + 63) / 64), dim3(64), 0, st, A, level);
  }
  hipLaunchKernelGGL(zk_msm_bucket_join<C>, dim3((total + 63) / 64), dim3(64), 0, st, A);
  if (A.plane_sums) {
    // sum_b (b + 1) bucket[b] by bit planes (zkwg_msm_core.h): level 0 into node_s, joins alternate node_a / node_s, one lane per window folds
    const u32 rows = A.KS * A.c;
    u32 n_in = zk_msm_plane_n0(A.nb);
    hipLaunchKernelGGL(zk_msm_plane0<C>, dim3((rows * n_in + 63) / 64), dim3(64), 0, st, A, A.node_s);
    X* cur = A.node_s;
    while (n_in > 1) {
      const u32 n_out = (n_in + ZK_MSM_PFAN - 1) / ZK_MSM_PFAN;
      X* nxt = cur == A.node_s ? A.node_a : A.node_s;
      hipLaunchKernelGGL(zk_msm_plane_join<C>, dim3((rows * n_out + 63) / 64), dim3(64), 0, st, (const X*)cur, rows, n_in, nxt);
      cur = nxt; n_in = n_out;
    }
    hipLaunchKernelGGL(zk_msm_plane_window<C>, dim3((A.KS + 63) / 64), dim3(64), 0, st, A, (const X*)cur);
  } else {
    const X* in_s = A.bucket; const X* in_a = nullptr;
    u32 n_in = A.nb, span = 1, half = A.KS * ((A.nb + ZK_MSM_FAN - 1) / ZK_MSM_FAN), flip = 0;
    for (;;) {
      const u32 n_out = (n_in + ZK_MSM_FAN - 1) / ZK_MSM_FAN;
      X* out_s = A.node_s + (size_t)flip * half;
      X* out_a = A.node_a + (size_t)flip * half;
      hipLaunchKernelGGL(zk_msm_reduce<C>, dim3((A.KS * n_out + 63) / 64), dim3(64), 0, st, A, in_s, in_a, n_in, span, out_s, out_a);
      if (n_out == 1) break;
      in_s = out_s; in_a = out_a; n_in = n_out; span *= ZK_MSM_FAN; flip ^= 1;
    }
  }
  if (A.ones_apart) {
    // the sum of the bases with scalar 1: ZK_MSM_ONES per thread, then ZK_MSM_JOIN-way joins; the halves of A.ones alternate and the last join lands in ones[0]
    const u32 half1 = (A.n + ZK_MSM_ONES - 1) / ZK_MSM_ONES;
    u32 m = half1, levels = 0;
    for (u32 q = m; q > 1; q = (q + ZK_MSM_JOIN - 1) / ZK_MSM_JOIN) ++levels;
    X* cur = A.ones + ((levels & 1u) ? half1 : 0);
    {
      ZkMsmArgsT<C> B = A; B.ones = cur;
      hipLaunchKernelGGL(zk_msm_ones<C>, dim3((half1 + 63) / 64), dim3(64), 0, st, B);
    }
    while (m > 1) {
      const u32 m2 = (m + ZK_MSM_JOIN - 1) / ZK_MSM_JOIN;
      X* nxt = cur == A.ones ? A.ones + half1 : A.ones;
      hipLaunchKernelGGL(zk_msm_tree<C>, dim3((m2 + 63) / 64), dim3(64), 0, st, (const X*)cur, m, nxt);
      cur = nxt; m = m2;
    }
  }
  hipLaunchKernelGGL(zk_msm_combine<C>, dim3(1), dim3(64), 0, st, A);
}
template <class C> __global__ __launch_bounds__(64) void zk_msm_shift(const typename C::Affine* bases, typename C::Affine* ext, u32 n, u32 c, u32 K) {
  const u32 i = blockIdx.x * 64u + threadIdx.x;
  if constexpr (sizeof(typename C::Affine) == sizeof(G1Affine))
    zk_msm_shift_thread<C>(bases, ext, n, c, K, i, [](const G1Xyzz& a) { return g1_to_affine(a); });
  else
    zk_msm_shift_thread<C>(bases, ext, n, c, K, i, [](const G2Xyzz& a) { return g2_to_affine(a); });
}
void zk_msm_shift_launch(int group, const void* bases, void* ext, u32 n, u32 c, u32 K, hipStream_t st) {
  if (group == 1) hipLaunchKernelGGL(zk_msm_shift<ZkCurveG1>, dim3((n + 63) / 64), dim3(64), 0, st, (const G1Affine*)bases, (G1Affine*)ext, n, c, K);
  else hipLaunchKernelGGL(zk_msm_shift<ZkCurveG2>, dim3((n + 63) / 64), dim3(64), 0, st, (const G2Affine*)bases, (G2Affine*)ext, n, c, K);
}
void zk_msm_launch(const ZkMsmArgs& A, hipStream_t st) { zk_msm_launch_t<ZkCurveG1>(A, st); }
void zk_msm_launch_g2(const ZkMsmArgsT<ZkCurveG2>& A, hipStream_t st) { zk_msm_launch_t<ZkCurveG2>(A, st); }

// ---- fixed-base multiples: out[i] = k_i G for the group's generator -- how a key with a known trapdoor is turned into bases
// (tests, tools: oracle/pyref/groth16.py makes the scalars).  One thread per scalar: double-and-add from the top over a 4-bit
// window table of the generator held in LDS.
template <class C>
__global__ __launch_bounds__(64) void zk_fixed_base(const typename C::Affine gen, const Fr* __restrict__ k, typename C::Affine* __restrict__ out, u32 n) {
  __shared__ typename C::Xyzz tab[16];
  if (threadIdx.x == 0) {
    tab[0] = C::inf();
    for (int j = 1; j < 16; ++j) tab[j] = C::add_mixed(tab[j - 1], gen);
  }
  __syncthreads();
  const u32 i = blockIdx.x * 64u + threadIdx.x;
  if (i >= n) return;
  const Fr s = k[i];
  typename C::Xyzz acc = C::inf();
  for (int w = 63; w >= 0; --w) {
    acc = C::dbl(C::dbl(C::dbl(C::dbl(acc))));
    const u32 d = (u32)(s.l[w >> 4] >> (4 * (w & 15))) & 15u;
    if (d) acc = C::add(acc, tab[d]);
  }
  if constexpr (sizeof(typename C::Affine) == sizeof(G1Affine)) out[i] = g1_to_affine(*(const G1Xyzz*)&acc);
  else out[i] = g2_to_affine(*(const G2Xyzz*)&acc);
}
void zk_fixed_base_g1_launch(const G1Affine& gen, const Fr* k, G1Affine* out, u32 n, hipStream_t st) {
  hipLaunchKernelGGL(zk_fixed_base<ZkCurveG1>, dim3((n + 63) / 64), dim3(64), 0, st, gen, k, out, n);
}
void zk_fixed_base_g2_launch(const G2Affine& gen, const Fr* k, G2Affine* out, u32 n, hipStream_t st) {
  hipLaunchKernelGGL(zk_fixed_base<ZkCurveG2>, dim3((n + 63) / 64), dim3(64), 0, st, gen, k, out, n);
}
