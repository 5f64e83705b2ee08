// Microbenchmark (round 6): the 9 x 29-bit Montgomery product of zkwg_fr29.h as the compiler emits it against the same product with
// every column's multiply-adds CHAINED onto the shifted carry by inline assembly.  The compiler splits each column into a fresh chain
// (v_mad_u64_u32 ..., 0) that it re-joins with a v_lshl_add_u64 -- 16 extra VALU instructions per product; the assembly form has none of
// them but gets an s_nop after every statement (the hazard recogniser cannot see into it).  s_nop takes no VALU issue slot: which form
// is faster is this measurement.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zk-email-verify_amd/csrc tools/madchain.hip -o tools/madchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "zkwg_fr29.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void zmad(u64& acc, u32 a, u32 b) { u64 cy; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "v"(b)); }
__device__ __forceinline__ void zmads(u64& acc, u32 a, u32 b) { u64 cy; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "s"(b)); }
__device__ __forceinline__ Fr29 fr29_mul_asm(const Fr29& a, const Fr29& b) {
  u32 q[9];
  Fr29 r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) zmad(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; ++i) zmads(acc, q[i], ZKR29_P(k - i));
    q[k] = ((u32)acc * ZK29_N0) & ZK29_M;
    zmads(acc, q[k], ZKR29_P(0));
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) zmad(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 8; i < 9; ++i) zmads(acc, q[i], ZKR29_P(k - i));
    r.l[k - 9] = (u32)acc & ZK29_M;
    acc >>= 29;
  }
  r.l[8] = (u32)acc;
  return r;
}
template <int CH, bool ASM>
__global__ __launch_bounds__(256) void prod(const Fr* in, Fr* out, u32 iters) {
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  Fr29 x[CH];
  const Fr29 y = fr29_from_fr(in[(tid + 1) & 1023]);
  for (int k = 0; k < CH; ++k) x[k] = fr29_from_fr(in[(tid + 7 * k) & 1023]);
  for (u32 i = 0; i < iters; ++i)
    for (int k = 0; k < CH; ++k) x[k] = ASM ? fr29_mul_asm(x[k], y) : fr29_mul(x[k], y);
  Fr s = fr29_to_fr(x[0]);
  for (int k = 1; k < CH; ++k) s = fr_add(s, fr29_to_fr(x[k]));
  out[tid] = s;
}
template <class F> static float time_ms(F f, int reps = 3) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  return best;
}
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  std::vector<Fr> h(1024);
  u64 s = 0x243f6a8885a308d3ull;
  for (auto& v : h) { for (int i = 0; i < 4; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v.l[i] = s; } v.l[3] &= 0x0fffffffffffffffull; }
  Fr *din, *d0, *d1;
  const u32 maxb = cus * 8;
  CK(hipMalloc(&din, 1024 * sizeof(Fr))); CK(hipMalloc(&d0, (size_t)maxb * 256 * sizeof(Fr))); CK(hipMalloc(&d1, (size_t)maxb * 256 * sizeof(Fr)));
  CK(hipMemcpy(din, h.data(), 1024 * sizeof(Fr), hipMemcpyHostToDevice));
  // same results?
  hipLaunchKernelGGL((prod<2, false>), dim3(cus), dim3(256), 0, 0, din, d0, 37u);
  hipLaunchKernelGGL((prod<2, true>), dim3(cus), dim3(256), 0, 0, din, d1, 37u);
  CK(hipDeviceSynchronize());
  std::vector<Fr> r0((size_t)cus * 256), r1((size_t)cus * 256);
  CK(hipMemcpy(r0.data(), d0, r0.size() * sizeof(Fr), hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), d1, r1.size() * sizeof(Fr), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < r0.size(); ++i) for (int k = 0; k < 4; ++k) bad += r0[i].l[k] != r1[i].l[k];
  printf("{\"mismatching_words\": %zu", bad);
  const u32 iters = 512;
  for (int wps : {1, 2, 4, 8}) {
    const u32 blocks = cus * wps;
    const double n1 = (double)blocks * 256 * iters;
    const float c1 = time_ms([&] { hipLaunchKernelGGL((prod<1, false>), dim3(blocks), dim3(256), 0, 0, din, d0, iters); });
    const float c2 = time_ms([&] { hipLaunchKernelGGL((prod<2, false>), dim3(blocks), dim3(256), 0, 0, din, d0, iters); });
    const float a1 = time_ms([&] { hipLaunchKernelGGL((prod<1, true>), dim3(blocks), dim3(256), 0, 0, din, d0, iters); });
    const float a2 = time_ms([&] { hipLaunchKernelGGL((prod<2, true>), dim3(blocks), dim3(256), 0, 0, din, d0, iters); });
    printf(",\n \"%d_waves_per_simd\": {\"compiler_1_chain_G_per_s\": %.1f, \"compiler_2_chains\": %.1f, \"asm_chained_1_chain\": %.1f, \"asm_chained_2_chains\": %.1f}", wps,
           n1 / c1 / 1e6, 2 * n1 / c2 / 1e6, n1 / a1 / 1e6, 2 * n1 / a2 / 1e6);
  }
  printf("}\n");
  return 0;
}
