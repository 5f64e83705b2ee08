// Which store patterns feel WHERE a buffer was placed?  (tuning tool; see tools/placement_probe.py)
// T buffers of `gb` GiB each from hipMalloc; per buffer: several fill shapes, GB/s each.
//   hipcc -O3 --offload-arch=gfx950 tools/placebench.hip -o tools/placebench && tools/placebench [tiles=7] [gib=27]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// piece of `cpw` 16-byte chunks per workgroup; mode 0: plain order, 1: each XCD (blockIdx % 8) one contiguous eighth of the
// buffer, 2: pseudo-random piece order (multiplicative permutation)
template <int MODE>
__global__ __launch_bounds__(256) void fill_piece(uint4* dst, unsigned cpw, unsigned npieces, unsigned mult) {
  unsigned p = blockIdx.x;
  if (MODE == 1) { const unsigned per = gridDim.x >> 3; if (p < per * 8u) p = (p & 7u) * per + (p >> 3); }
  if (MODE == 2) p = (unsigned)(((unsigned long long)p * mult) % npieces);
  uint4* d = dst + (unsigned long long)p * cpw;
  const uint4 v = make_uint4(p, 1, 2, 3);
  for (unsigned c = threadIdx.x; c < cpw; c += 256) d[c] = v;
}
// one 16-byte store per `stride` bytes: translation / row-activation bound
__global__ __launch_bounds__(256) void touch(uint4* dst, unsigned long long stride16, unsigned long long n) {
  const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i * stride16] = make_uint4((unsigned)i, 0, 0, 0);
}
template <class F> float timeit(F f, int iters = 4) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / iters;
}
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 7;
  const unsigned long long gib = argc > 2 ? atoll(argv[2]) : 27;
  const unsigned long long bytes = gib << 30, total = bytes / 16;
  std::vector<uint4*> buf(T);
  for (int t = 0; t < T; ++t) { CK(hipMalloc((void**)&buf[t], bytes)); CK(hipMemset(buf[t], 0, bytes)); }
  printf("%-34s", "pattern \\ buffer");
  for (int t = 0; t < T; ++t) printf(" %7d", t);
  printf("   (GB/s; touch rows: M stores/s)\n");
  auto row = [&](const char* name, auto launch, double unit_bytes) {
    printf("%-34s", name);
    for (int t = 0; t < T; ++t) { const float ms = timeit([&] { launch(buf[t]); }); printf(" %7.0f", unit_bytes / ms / 1e6); }
    printf("\n"); fflush(stdout);
  };
  for (unsigned kb : {4u, 8u, 32u}) {
    const unsigned cpw = kb * 64, np = (unsigned)(total / cpw);
    char nm[64];
    snprintf(nm, sizeof nm, "fill %2u KiB/WG plain", kb);
    row(nm, [&](uint4* d) { hipLaunchKernelGGL(fill_piece<0>, dim3(np), dim3(256), 0, 0, d, cpw, np, 0u); }, (double)bytes);
    snprintf(nm, sizeof nm, "fill %2u KiB/WG XCD eighths", kb);
    row(nm, [&](uint4* d) { hipLaunchKernelGGL(fill_piece<1>, dim3(np), dim3(256), 0, 0, d, cpw, np, 0u); }, (double)bytes);
    snprintf(nm, sizeof nm, "fill %2u KiB/WG random order", kb);
    row(nm, [&](uint4* d) { hipLaunchKernelGGL(fill_piece<2>, dim3(np), dim3(256), 0, 0, d, cpw, np, 2654435761u % np | 1u); }, (double)bytes);
  }
  for (unsigned long long stride : {4096ull, 65536ull, 2097152ull}) {
    const unsigned long long n = bytes / stride;
    char nm[64];
    snprintf(nm, sizeof nm, "touch 16 B per %llu KiB", stride >> 10);
    row(nm, [&](uint4* d) { hipLaunchKernelGGL(touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, stride / 16, n); }, (double)n * 1e3);
  }
  return 0;
}
