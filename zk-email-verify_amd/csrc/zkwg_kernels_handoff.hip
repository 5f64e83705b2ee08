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
