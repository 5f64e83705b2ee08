// zk_r1cs_check -- `checkConstraints` on the device (circom_tester's
// `circuit.checkConstraints(witness)`, packages/circuits/tests/email-verifier.test.ts:44):
// one thread per (constraint, witness); first_bad[e] = smallest index of a violated constraint of
// witness e (0xffffffffffffffff when all hold).
#include "zkwg_kernels.h"
#include "zkwg_r1cs.h"
#include "zkwg_full.h"

__global__ __launch_bounds__(256) void zk_r1cs_check(const u64* __restrict__ row_ptr, const u32* __restrict__ wire,
                                                     const Fr* __restrict__ coef, const u8* __restrict__ kind,
                                                     u32 m, const u8* __restrict__ wit, u64 stride,
                                                     unsigned long long* __restrict__ first_bad) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  const u32 e = blockIdx.y;
  if (i >= m) return;
  const Fr* w = (const Fr*)(wit + (u64)e * stride);
  if (!zk_r1cs_check_one(row_ptr, wire, coef, kind, i, w)) atomicMin(first_bad + e, (unsigned long long)i);
}

// zk_linear_fill -- the signals a compiled circuit numbers but the schedule does not produce (aliases, constants,
// linear combinations; zkwg_full.h): w[dst] = sum coef * w[src] over wires zk_expand has already written.
// One thread per (row, witness); rows are sorted by destination.
__global__ __launch_bounds__(256) void zk_linear_fill(const u64* __restrict__ row_ptr, const u32* __restrict__ dst,
                                                      const u32* __restrict__ src, const Fr* __restrict__ coef,
                                                      const u8* __restrict__ kind, u64 n_rows, u8* __restrict__ wit,
                                                      u64 stride) {
  const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  Fr* w = (Fr*)(wit + (u64)blockIdx.y * stride);
  w[dst[r]] = zk_linear_row(row_ptr, src, coef, kind, r, w);
}
