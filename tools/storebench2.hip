// Store-order microbenchmark (tuning tool, round 3): which assignment of 16-byte chunks to workgroups and to
// time does HBM like?  All variants write the same 24 GiB with plain global_store_dwordx4, 256-thread WGs.
//   contig   : WG i writes one contiguous S-byte range (zk_expand's geometry at S = 64 KiB)
//   inter    : groups of G consecutive WGs share G*S bytes; WG j of a group writes pieces j, j+G, j+2G ... of P bytes
//              (at any time a group's stores fall into one dense G*P window)
//   persist  : grid of NWG workgroups, WG w writes pieces w, w+NWG, ... of P bytes (one dense NWG*P window sweeps the buffer)
//   +xcd     : blockIdx -> unit remap that gives each XCD (blockIdx % 8) its own contiguous eighth
//   +dep     : two dependent table loads before each piece's stores (the segment-table -> image-word chain of zk_expand)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcd_unit(unsigned blk, int remap) {
  if (!remap) return blk;
  const unsigned per = gridDim.x >> 3;
  return blk < per * 8u ? (blk & 7u) * per + (blk >> 3) : blk;
}
__device__ __forceinline__ unsigned dep_value(const unsigned* tab, unsigned long long piece, int dep) {
  if (!dep) return 1u;
  const unsigned i = tab[piece & 4095u];
  return tab[4096u + (i & 4095u)] & 1u;
}
// pieces of PCH chunks; WG `unit` of group (unit / G) writes pieces (unit % G) + G * k, k < NP
template <int NP>
__global__ __launch_bounds__(256) void k_inter(uint4* dst, unsigned G, unsigned PCH, int remap, int dep, const unsigned* tab) {
  const unsigned unit = xcd_unit(blockIdx.x, remap);
  const unsigned long long gbase = (unsigned long long)(unit / G) * G * NP * PCH;
  const unsigned j = unit % G;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const unsigned long long piece = gbase / PCH + j + (unsigned long long)G * k;
    const unsigned v = dep_value(tab, piece, dep);
    for (unsigned c = threadIdx.x; c < PCH; c += 256) dst[piece * PCH + c] = make_uint4(v, 0, 0, 0);
  }
}
__global__ __launch_bounds__(256) void k_persist(uint4* dst, unsigned long long npieces, unsigned PCH, int remap, int dep, const unsigned* tab) {
  // remap: XCD x owns pieces [x * npieces/8, (x+1) * npieces/8); its WGs stride through them
  const unsigned nwg = gridDim.x;
  if (remap) {
    const unsigned x = blockIdx.x & 7u, w = blockIdx.x >> 3, per = nwg >> 3;
    const unsigned long long share = npieces >> 3, base = share * x;
    for (unsigned long long p = w; p < share; p += per) {
      const unsigned v = dep_value(tab, base + p, dep);
      for (unsigned c = threadIdx.x; c < PCH; c += 256) dst[(base + p) * PCH + c] = make_uint4(v, 0, 0, 0);
    }
  } else {
    for (unsigned long long p = blockIdx.x; p < npieces; p += nwg) {
      const unsigned v = dep_value(tab, p, dep);
      for (unsigned c = threadIdx.x; c < PCH; c += 256) dst[p * PCH + c] = make_uint4(v, 0, 0, 0);
    }
  }
}
// one piece per WG (PCH chunks), optional dependent loads first: the torch.fill_ shape at PCH = 256
__global__ __launch_bounds__(256) void k_single(uint4* dst, unsigned PCH, int remap, int dep, const unsigned* tab) {
  const unsigned long long piece = xcd_unit(blockIdx.x, remap);
  const unsigned v = dep_value(tab, piece, dep);
  for (unsigned c = threadIdx.x; c < PCH; c += 256) dst[piece * PCH + c] = make_uint4(v, 0, 0, 0);
}
template <class F> float timeit(F f, int iters = 4) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}
int main() {
  const unsigned long long bytes = 24ull << 30, total = bytes / 16;
  uint4* d; CK(hipMalloc((void**)&d, bytes)); CK(hipMemset(d, 0, bytes));
  unsigned* tab; CK(hipMalloc((void**)&tab, 8192 * 4));
  { unsigned h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (unsigned)(i * 2654435761u >> 7); CK(hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice)); }
  float ms;
  for (int dep = 0; dep < 2; ++dep) for (int remap = 0; remap < 2; ++remap) {
    for (unsigned kb : {4u, 8u, 16u, 64u}) {
      const unsigned pch = kb * 64;
      ms = timeit([&] { hipLaunchKernelGGL(k_single, dim3((unsigned)(total / pch)), dim3(256), 0, 0, d, pch, remap, dep, tab); });
      printf("single  %3u KB/WG            xcd=%d dep=%d: %7.3f ms %6.0f GB/s\n", kb, remap, dep, ms, bytes / ms / 1e6);
    }
    // 64 KiB per WG as 16 pieces of 4 KiB / 4 of 16 KiB / 8 of 8 KiB, groups of G
    for (unsigned G : {4u, 16u, 64u, 256u}) {
      ms = timeit([&] { hipLaunchKernelGGL((k_inter<16>), dim3((unsigned)(total / (16 * 256))), dim3(256), 0, 0, d, G, 256u, remap, dep, tab); });
      printf("inter   16 x 4 KB  G=%3u      xcd=%d dep=%d: %7.3f ms %6.0f GB/s\n", G, remap, dep, ms, bytes / ms / 1e6);
    }
    for (unsigned G : {4u, 16u, 64u}) {
      ms = timeit([&] { hipLaunchKernelGGL((k_inter<4>), dim3((unsigned)(total / (4 * 1024))), dim3(256), 0, 0, d, G, 1024u, remap, dep, tab); });
      printf("inter    4 x 16 KB G=%3u      xcd=%d dep=%d: %7.3f ms %6.0f GB/s\n", G, remap, dep, ms, bytes / ms / 1e6);
    }
    for (unsigned G : {16u, 64u}) {
      ms = timeit([&] { hipLaunchKernelGGL((k_inter<4>), dim3((unsigned)(total / (4 * 256))), dim3(256), 0, 0, d, G, 256u, remap, dep, tab); });
      printf("inter    4 x 4 KB  G=%3u      xcd=%d dep=%d: %7.3f ms %6.0f GB/s\n", G, remap, dep, ms, bytes / ms / 1e6);
    }
    for (unsigned nwg : {2048u, 4096u}) for (unsigned kb : {4u, 16u, 64u}) {
      const unsigned pch = kb * 64;
      ms = timeit([&] { hipLaunchKernelGGL(k_persist, dim3(nwg), dim3(256), 0, 0, d, total / pch, pch, remap, dep, tab); });
      printf("persist %3u KB pieces nwg=%4u xcd=%d dep=%d: %7.3f ms %6.0f GB/s\n", kb, nwg, remap, dep, ms, bytes / ms / 1e6);
    }
  }
  return 0;
}
