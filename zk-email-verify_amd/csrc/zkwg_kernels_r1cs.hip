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

// zk_r1cs_eval -- first stage of a Groth16 prover on the device-resident witness (snarkjs `groth16.prove`: the
// evaluations A.w, B.w, C.w of every constraint, which the NTT stage turns into the quotient polynomial; second half of
// `fullProve`, packages/helpers/src/chunked-zkey.ts:80).  One thread per (linear combination, witness); out[e] holds
// the m values of A, then B, then C.  A witness in Montgomery form (zkwg_expand_montgomery_device) gives evaluations
// in Montgomery form -- coefficient (Montgomery) x value (Montgomery) -> Montgomery -- with no conversion pass.
__global__ __launch_bounds__(256) void zk_r1cs_eval(const u64* __restrict__ row_ptr, const u32* __restrict__ wire,
                                                    const Fr* __restrict__ coef, const u8* __restrict__ kind, u32 m,
                                                    const u8* __restrict__ wit, u64 stride, u8* __restrict__ out, u64 out_stride, int mont) {
  const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;   // A rows, B rows, C rows
  if (t >= 3ull * m) return;
  const u32 which = (u32)(t / m);
  const u64 i = t - (u64)which * m;
  const Fr* w = (const Fr*)(wit + (u64)blockIdx.y * stride);
  bool canon = true;   // a non-reduced witness value is reduced mod r, never dropped (zkwg_check_constraints is what rejects such a witness)
  ((Fr*)(out + (u64)blockIdx.y * out_stride))[t] = zk_r1cs_lc(row_ptr, wire, coef, kind, 3 * i + which, w, &canon, mont != 0, true);
}

