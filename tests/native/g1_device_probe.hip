// Compile probe (tests/test_g1_cpu.py): the device path of zkwg_fq.h / zkwg_g1.h for gfx950 -- does the mixed addition a bucket
// kernel is made of stay in registers?  Not product code; never launched.
#include "zkwg_g1.h"
__global__ __launch_bounds__(256) void zk_g1_probe_accumulate(const G1Affine* pts, const u32* first, G1Xyzz* out) {
  const u32 b = blockIdx.x * 256u + threadIdx.x;
  G1Xyzz acc = g1_xyzz_inf();
  for (u32 i = first[b]; i < first[b + 1]; ++i) acc = g1_add_mixed(acc, pts[i]);
  out[b] = acc;
}
__global__ __launch_bounds__(256) void zk_g1_probe_add(const G1Xyzz* a, G1Xyzz* out) {
  const u32 b = blockIdx.x * 256u + threadIdx.x;
  out[b] = g1_add(a[2 * b], a[2 * b + 1]);
}
