// G1 multi-exponentiation kernels (bodies: zkwg_msm_core.h).  DRAFT, branch next/msm: never run on a GPU yet.
#include "zkwg_dev.h"
#include "zkwg_msm_core.h"

__global__ __launch_bounds__(256) void zk_msm_count(ZkMsmArgs A) { zk_msm_count_thread(A, blockIdx.x * 256u + threadIdx.x); }
__global__ __launch_bounds__(1024) void zk_msm_scan(ZkMsmArgs A) {
  __shared__ u32 partial[1025];
  zk_msm_scan_thread(A, threadIdx.x, 1024u, partial, 0);
  __syncthreads();
  zk_msm_scan_thread(A, threadIdx.x, 1024u, partial, 1);
}
__global__ __launch_bounds__(256) void zk_msm_scatter(ZkMsmArgs A) { zk_msm_scatter_thread(A, blockIdx.x * 256u + threadIdx.x); }
// 4 wavefronts per SIMD (110 VGPRs for the accumulate loop): 64-thread workgroups so that neighbouring buckets -- runs of similar
// length -- share a wavefront
__global__ __launch_bounds__(64) void zk_msm_buckets(ZkMsmArgs A) { zk_msm_bucket_thread(A, blockIdx.x * 64u + threadIdx.x); }
__global__ __launch_bounds__(64) void zk_msm_reduce(ZkMsmArgs A, const G1Xyzz* in_s, const G1Xyzz* in_a, u32 n_in, u32 span, G1Xyzz* out_s, G1Xyzz* out_a) {
  zk_msm_reduce_thread(A, blockIdx.x * 64u + threadIdx.x, in_s, in_a, n_in, span, out_s, out_a);
}
__global__ __launch_bounds__(64) void zk_msm_ones(ZkMsmArgs A) { zk_msm_ones_thread(A, blockIdx.x * 64u + threadIdx.x); }
__global__ __launch_bounds__(64) void zk_msm_tree(const G1Xyzz* in, u32 n_in, G1Xyzz* out) { zk_msm_tree_thread(in, n_in, out, blockIdx.x * 64u + threadIdx.x); }
__global__ __launch_bounds__(64) void zk_msm_combine(ZkMsmArgs A) { if (threadIdx.x == 0 && blockIdx.x == 0) zk_msm_combine_thread(A); }

// launches of one multi-exponentiation on `st` (A.count zeroed here)
void zk_msm_launch(const ZkMsmArgs& A, hipStream_t st) {
  const u32 total = A.K * A.nb;
  hipMemsetAsync(A.count, 0, ((size_t)total + 1) * 4, st);
  hipLaunchKernelGGL(zk_msm_count, dim3((A.n + 255) / 256), dim3(256), 0, st, A);
  hipLaunchKernelGGL(zk_msm_scan, dim3(1), dim3(1024), 0, st, A);
  hipLaunchKernelGGL(zk_msm_scatter, dim3((A.n + 255) / 256), dim3(256), 0, st, A);
  hipLaunchKernelGGL(zk_msm_buckets, dim3((total + 63) / 64), dim3(64), 0, st, A);
  const G1Xyzz* in_s = A.bucket; const G1Xyzz* in_a = nullptr;
  u32 n_in = A.nb, span = 1, half = A.K * ((A.nb + 31) / 32), flip = 0;
  for (;;) {
    const u32 n_out = (n_in + 31) / 32;
    G1Xyzz* out_s = A.node_s + (size_t)flip * half;
    G1Xyzz* out_a = A.node_a + (size_t)flip * half;
    hipLaunchKernelGGL(zk_msm_reduce, dim3((A.K * n_out + 63) / 64), dim3(64), 0, st, A, in_s, in_a, n_in, span, out_s, out_a);
    if (n_out == 1) break;
    in_s = out_s; in_a = out_a; n_in = n_out; span *= 32; flip ^= 1;
  }
  if (A.ones_apart) {
    // the sum of the bases with scalar 1: 64 per thread, then 64-way joins; the halves of A.ones alternate and the last join lands in ones[0]
    const u32 half1 = (A.n + 63) / 64;
    u32 m = half1, levels = 0;
    for (u32 q = m; q > 1; q = (q + 63) / 64) ++levels;
    G1Xyzz* cur = A.ones + ((levels & 1u) ? half1 : 0);
    {
      ZkMsmArgs B = A; B.ones = cur;
      hipLaunchKernelGGL(zk_msm_ones, dim3((half1 + 63) / 64), dim3(64), 0, st, B);
    }
    while (m > 1) {
      const u32 m2 = (m + 63) / 64;
      G1Xyzz* nxt = cur == A.ones ? A.ones + half1 : A.ones;
      hipLaunchKernelGGL(zk_msm_tree, dim3((m2 + 63) / 64), dim3(64), 0, st, (const G1Xyzz*)cur, m, nxt);
      cur = nxt; m = m2;
    }
  }
  hipLaunchKernelGGL(zk_msm_combine, dim3(1), dim3(64), 0, st, A);
}
