// Does the "slow part of HBM"  show at the granularity of physical allocation chunks?  
// N physical chunks of `gib` GiB from hipMemCreate, each mapped at its own address and filled with zk_expand's store shape
// (32 KiB per 256-thread workgroup, each XCD one contiguous eighth) and, for contrast, with 4 KiB per workgroup; GB/s per chunk.
// Then the fastest K chunks are mapped back to back into ONE address range and the whole range is filled: what a ring built
// from chosen chunks would see.
//   hipcc -O3 --offload-arch=gfx950 tools/chunkbench.hip -o tools/chunkbench && tools/chunkbench [chunks=96] [gib=1] [keep=58]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s (%d) at line %d\n", hipGetErrorString(e), (int)e, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill_piece(uint4* dst, unsigned cpw) {
  unsigned p = blockIdx.x;
  const unsigned per = gridDim.x >> 3;
  if (p < per * 8u) p = (p & 7u) * per + (p >> 3);
  uint4* d = dst + (unsigned long long)p * cpw;
  const uint4 v = make_uint4(p, 1, 2, 3);
  for (unsigned c = threadIdx.x; c < cpw; c += 256) d[c] = v;
}
static float time_fill(void* ptr, size_t bytes, unsigned kb, int reps) {
  const unsigned cpw = kb * 64, np = (unsigned)(bytes / 16 / cpw);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(fill_piece, dim3(np), dim3(256), 0, 0, (uint4*)ptr, cpw);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill_piece, dim3(np), dim3(256), 0, 0, (uint4*)ptr, cpw);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / reps;
}
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 96;
  const size_t gib = argc > 2 ? atoll(argv[2]) : 1;
  const int keep = argc > 3 ? atoi(argv[3]) : 58;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t chunk = ((gib << 30) + gran - 1) / gran * gran;
  printf("allocation granularity %zu bytes, chunk %zu bytes, %d chunks\n", gran, chunk, N);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h(N);
  std::vector<float> ms32(N), ms4(N);
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, chunk, 0, nullptr, 0));
  int got = 0;
  for (int i = 0; i < N; ++i) {
    if (hipMemCreate(&h[i], chunk, &prop, 0) != hipSuccess) { printf("hipMemCreate stopped at chunk %d\n", i); break; }
    ++got;
    CK(hipMemMap(va, chunk, 0, h[i], 0));
    CK(hipMemSetAccess(va, chunk, &acc, 1));
    ms32[i] = time_fill(va, chunk, 32, 6);
    ms4[i] = time_fill(va, chunk, 4, 6);
    CK(hipMemUnmap(va, chunk));
  }
  CK(hipMemAddressFree(va, chunk));
  printf("chunk: GB/s with 32 KiB per workgroup | 4 KiB per workgroup\n");
  for (int i = 0; i < got; ++i) printf("%3d  %7.0f  %7.0f\n", i, chunk / ms32[i] / 1e6, chunk / ms4[i] / 1e6);
  std::vector<int> order(got);
  for (int i = 0; i < got; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return ms32[a] < ms32[b]; });
  const int K = std::min(keep, got);
  auto range_rate = [&](const std::vector<int>& idx, const char* what) {
    void* base = nullptr;
    const size_t total = (size_t)idx.size() * chunk;
    CK(hipMemAddressReserve(&base, total, 0, nullptr, 0));
    for (size_t k = 0; k < idx.size(); ++k) CK(hipMemMap((char*)base + k * chunk, chunk, 0, h[idx[k]], 0));
    CK(hipMemSetAccess(base, total, &acc, 1));
    const float ms = time_fill(base, total, 32, 3);
    printf("%-46s %3zu chunks = %.1f GB: %7.0f GB/s (32 KiB per workgroup)\n", what, idx.size(), total / 1e9, total / ms / 1e6);
    CK(hipMemUnmap(base, total));
    CK(hipMemAddressFree(base, total));
  };
  std::vector<int> fast(order.begin(), order.begin() + K), first(K), slow(order.end() - K, order.end());
  for (int i = 0; i < K; ++i) first[i] = i;
  std::sort(fast.begin(), fast.end());
  range_rate(first, "one range of the FIRST chunks (allocation order)");
  range_rate(fast, "one range of the FASTEST chunks");
  range_rate(slow, "one range of the SLOWEST chunks");
  float lo = 1e30f, hi = 0;
  for (int i = 0; i < got; ++i) { lo = std::min(lo, ms32[i]); hi = std::max(hi, ms32[i]); }
  printf("per-chunk rate: fastest %.0f, slowest %.0f GB/s (spread %.1f %%)\n", chunk / lo / 1e6, chunk / hi / 1e6, 100.0 * (hi - lo) / lo);
  for (int i = 0; i < got; ++i) CK(hipMemRelease(h[i]));
  return 0;
}
