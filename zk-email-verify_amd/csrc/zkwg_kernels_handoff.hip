// Prover hand-off: a GPU Groth16 prover -- the second half of `snarkjs.groth16.fullProve`
// (packages/helpers/src/chunked-zkey.ts:80) -- consumes the witness scalars on the device; its NTT / MSM kernels
// usually want them in Montgomery form (x * 2^256 mod r), the `.wtns` format holds standard form.  This converts a
// device-resident witness in place, either way.  HBM-bound (32 B read + 32 B written per value); the 0 / 1 values
// that make up 95 % of an EmailVerifier witness take a constant instead of a product.
#include "zkwg_kernels.h"

__global__ __launch_bounds__(256) void zk_mont_convert(Fr* __restrict__ v, u64 n, int to_mont) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Fr x = v[i];
  if (fr_is_zero(x)) return;
  if (to_mont) {
    const bool one = x.l[0] == 1 && (x.l[1] | x.l[2] | x.l[3]) == 0;
    x = one ? fr_R() : fr_to_mont(x);
  } else {
    x = fr_eq(x, fr_R()) ? fr_from_u64(1) : fr_from_mont(x);
  }
  v[i] = x;
}

// the 96-byte result rows (w[1 .. 3] = pubkeyHash, shaHi, shaLo) of `count` witnesses `stride` bytes apart -> a packed table: the ring of
// the resident pipeline is mapped chunk by chunk (zkwg_vmm.hip), and a strided hipMemcpy2DAsync out of such a range is not something
// every runtime version accepts; a kernel reads it like any other device memory
__global__ __launch_bounds__(256) void zk_rows_copy(const u8* __restrict__ src, u64 stride, u8* __restrict__ dst, u32 count) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count * 6u) return;
  const u32 e = i / 6u, q = i - e * 6u;
  ((uint4*)(dst + (u64)e * 96u))[q] = ((const uint4*)(src + (u64)e * stride + 32u))[q];
}
extern "C" int zk_rows_copy_launch(const u8* src, u64 stride, u8* dst, u32 count, hipStream_t st) {
  if (!count) return 0;
  hipLaunchKernelGGL(zk_rows_copy, dim3((count * 6u + 255u) / 256u), dim3(256), 0, st, src, stride, dst, count);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// C.w = A.w o B.w row by row (Montgomery form), for a prover whose constraint system came from a zkey: section 4 of the file holds the
// rows of A and B only, and snarkjs' buildABC1 computes the third block exactly like this (groth16_prove.js; reference call site
// packages/helpers/src/chunked-zkey.ts:80-84).  d_abc: n_emails records A.w | B.w | C.w of n_rows values each, abc_stride bytes apart.
__global__ __launch_bounds__(256) void zk_abc_c_from_ab(Fr* __restrict__ abc, u64 stride_fr, u64 n_rows) {
  Fr* rec = abc + (u64)blockIdx.y * stride_fr;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (u64)gridDim.x * 256) {
    const Fr a = rec[i], b = rec[n_rows + i];
    rec[2 * n_rows + i] = (fr_is_zero(a) || fr_is_zero(b)) ? fr_zero() : fr_mont_mul(a, b);
  }
}
void zk_abc_c_from_ab_launch(void* d_abc, uint64_t abc_stride, uint64_t n_rows, uint32_t n_emails, hipStream_t st) {
  if (!n_emails || !n_rows) return;
  const u64 g = (n_rows + 255) / 256;
  hipLaunchKernelGGL(zk_abc_c_from_ab, dim3((u32)(g > 4096 ? 4096 : g), n_emails), dim3(256), 0, st, (Fr*)d_abc, abc_stride / 32, n_rows);
}
