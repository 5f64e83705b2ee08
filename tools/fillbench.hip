// Store-pattern microbenchmark: how should zk_expand shape its writes?  (tuning tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// each WG writes `chunks_per_wg` consecutive 16-byte chunks; WG i covers [i*cpw, (i+1)*cpw)
template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void fill_k(uint4* dst, unsigned chunks_per_wg, unsigned long long total) {
  unsigned long long base = (unsigned long long)blockIdx.x * chunks_per_wg;
  uint4 v = make_uint4(blockIdx.x & 1, 0, 0, 0);
  for (unsigned c = threadIdx.x; c < chunks_per_wg; c += THREADS * UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      unsigned cc = c + u * THREADS;
      if (cc < chunks_per_wg && base + cc < total) dst[base + cc] = v;
    }
  }
}
// persistent variant: grid = G WGs, each loops over portions p = blockIdx.x, +G, ...
template <int THREADS>
__global__ __launch_bounds__(THREADS) void fill_persist(uint4* dst, unsigned chunks_per_wg, unsigned long long nport) {
  uint4 v = make_uint4(1, 0, 0, 0);
  for (unsigned long long p = blockIdx.x; p < nport; p += gridDim.x) {
    unsigned long long base = p * chunks_per_wg;
    for (unsigned c = threadIdx.x; c < chunks_per_wg; c += THREADS) dst[base + c] = v;
  }
}

// wave-contiguous: each wave writes KB_PER_WAVE consecutive KiB (dense 1 KiB per store instruction)
template <int THREADS, int KB_PER_WAVE>
__global__ __launch_bounds__(THREADS) void fill_wavecontig(uint4* dst, unsigned long long total) {
  const unsigned waves = THREADS / 64, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long base = ((unsigned long long)blockIdx.x * waves + wave) * (KB_PER_WAVE * 64ull);
  uint4 v = make_uint4(blockIdx.x & 1, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < KB_PER_WAVE; ++k) {
    unsigned long long c = base + k * 64 + lane;
    if (c < total) dst[c] = v;
  }
}
// like fill_wavecontig, but every even lane first loads one u64 from a small L2-resident table (index
// derived from the chunk id) and writes a bit of it: the zk_expand access pattern without the tables
template <int THREADS, int KB_PER_WAVE>
__global__ __launch_bounds__(THREADS) void fill_wavecontig_ld(uint4* dst, const unsigned long long* tab, unsigned long long total) {
  const unsigned waves = THREADS / 64, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long base = ((unsigned long long)blockIdx.x * waves + wave) * (KB_PER_WAVE * 64ull);
  unsigned long long w[KB_PER_WAVE];
#pragma unroll
  for (int k = 0; k < KB_PER_WAVE; ++k) {
    unsigned long long c = base + k * 64 + lane;
    w[k] = (lane & 1) ? 0ull : tab[(c >> 6) % 952 + ((c >> 16) & 1023) * 952];
  }
#pragma unroll
  for (int k = 0; k < KB_PER_WAVE; ++k) {
    unsigned long long c = base + k * 64 + lane;
    if (c < total) dst[c] = make_uint4((unsigned)(w[k] >> (lane >> 1)) & 1u, 0, 0, 0);
  }
}
template <class F> float timeit(F f, int iters = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main() {
  const unsigned long long bytes = 24ull << 30;
  const unsigned long long total = bytes / 16;
  uint4* d; CK(hipMalloc((void**)&d, bytes));
  CK(hipMemset(d, 0, bytes));
  float ms = timeit([&] { CK(hipMemsetAsync(d, 0, bytes, 0)); });
  printf("hipMemset               : %7.3f ms %6.0f GB/s\n", ms, bytes / ms / 1e6);
  for (unsigned kb : {8u, 16u, 32u, 64u, 128u, 256u}) {
    unsigned cpw = kb * 1024 / 16;
    unsigned grid = (unsigned)((total + cpw - 1) / cpw);
    ms = timeit([&] { hipLaunchKernelGGL((fill_k<256, 1>), dim3(grid), dim3(256), 0, 0, d, cpw, total); });
    printf("fill 256thr u1 %4u KB/WG: %7.3f ms %6.0f GB/s\n", kb, ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((fill_k<256, 4>), dim3(grid), dim3(256), 0, 0, d, cpw, total); });
    printf("fill 256thr u4 %4u KB/WG: %7.3f ms %6.0f GB/s\n", kb, ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((fill_k<512, 2>), dim3(grid), dim3(512), 0, 0, d, cpw, total); });
    printf("fill 512thr u2 %4u KB/WG: %7.3f ms %6.0f GB/s\n", kb, ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((fill_k<1024, 1>), dim3(grid), dim3(1024), 0, 0, d, cpw, total); });
    printf("fill 1024thr   %4u KB/WG: %7.3f ms %6.0f GB/s\n", kb, ms, bytes / ms / 1e6);
  }
#define WC(T, K) do { unsigned long long per = (T / 64) * (K * 64ull); unsigned grid = (unsigned)((total + per - 1) / per); \
    ms = timeit([&] { hipLaunchKernelGGL((fill_wavecontig<T, K>), dim3(grid), dim3(T), 0, 0, d, total); }); \
    printf("wavecontig %4d thr %2d KB/wave: %7.3f ms %6.0f GB/s\n", T, K, ms, bytes / ms / 1e6); } while (0)
  WC(256, 1); WC(256, 2); WC(256, 4); WC(256, 8); WC(512, 2); WC(512, 4); WC(1024, 1); WC(1024, 2); WC(1024, 4); WC(64, 4); WC(64, 8); WC(128, 4);
  unsigned long long* tab; CK(hipMalloc((void**)&tab, 1024 * 952 * 8)); CK(hipMemset(tab, 0x55, 1024 * 952 * 8));
#define WL(T, K) do { unsigned long long per = (T / 64) * (K * 64ull); unsigned grid = (unsigned)((total + per - 1) / per); \
    ms = timeit([&] { hipLaunchKernelGGL((fill_wavecontig_ld<T, K>), dim3(grid), dim3(T), 0, 0, d, tab, total); }); \
    printf("wavecontig+load %4d thr %2d KB/wave: %7.3f ms %6.0f GB/s\n", T, K, ms, bytes / ms / 1e6); } while (0)
  WL(256, 1); WL(256, 2); WL(256, 4); WL(256, 8); WL(256, 16); WL(64, 4); WL(64, 8); WL(512, 4); WL(1024, 4);
  for (unsigned g : {256u * 4, 256u * 8, 256u * 16}) {
    unsigned cpw = 32 * 1024 / 16;
    ms = timeit([&] { hipLaunchKernelGGL((fill_persist<256>), dim3(g), dim3(256), 0, 0, d, cpw, total / cpw); });
    printf("persist grid %5u 32KB   : %7.3f ms %6.0f GB/s\n", g, ms, bytes / ms / 1e6);
  }
  return 0;
}
