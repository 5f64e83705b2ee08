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
  bool canon = true;
  ((Fr*)(out + (u64)blockIdx.y * out_stride))[t] = zk_r1cs_lc(row_ptr, wire, coef, kind, 3 * i + which, w, &canon, mont != 0);
}

// zk_o0_gather -- the witness of a fully numbered (`--O0` / `--O1`) circuit from the compact kept-v1 witness zk_expand
// staged: desc[w] names, for every wire of the compiled circuit, the kept-v1 slot it copies (produced signals and
// their aliases: 95 % of the wires); the other wires are linear rows written by zk_o0_rows.  One 16-byte chunk per
// lane, consecutive lanes write consecutive chunks; the reads follow the circuit's own locality (the compiler numbers
// a component's signals together).
#define ZK_O0_UNROLL 4   // chunks per thread: the table reads, then the gathers, then the stores of all of them are in flight together
__global__ __launch_bounds__(256) void zk_o0_gather(const u32* __restrict__ desc, u64 n_wires, const u8* __restrict__ kept,
                                                    u64 kept_stride, u8* __restrict__ out, u64 out_stride) {
  const uint4* __restrict__ kw = (const uint4*)(kept + (u64)blockIdx.y * kept_stride);
  uint4* __restrict__ o = (uint4*)(out + (u64)blockIdx.y * out_stride);
  const u64 c0 = (u64)blockIdx.x * (256 * ZK_O0_UNROLL) + threadIdx.x;
  u32 d[ZK_O0_UNROLL];
#pragma unroll
  for (int k = 0; k < ZK_O0_UNROLL; ++k) {
    const u64 c = c0 + (u64)k * 256;
    d[k] = c < 2 * n_wires ? desc[c >> 1] : 0xfffffffeu;
  }
  uint4 v[ZK_O0_UNROLL];
#pragma unroll
  for (int k = 0; k < ZK_O0_UNROLL; ++k)
    if (d[k] != 0xfffffffeu) v[k] = kw[(u64)d[k] * 2 + ((u32)(c0 + (u64)k * 256) & 1u)];
#pragma unroll
  for (int k = 0; k < ZK_O0_UNROLL; ++k)
    if (d[k] != 0xfffffffeu) o[c0 + (u64)k * 256] = v[k];   // (0xfffffffe: a linear row, zk_o0_rows writes it)
}

// zk_o0_rows -- the derived signals that are not plain aliases (zkwg_full.h): LANES lanes per row stride over its
// terms, the partial sums are folded with shuffles, lane 0 of the group writes the wire.  Same arithmetic as
// zk_linear_row.  4 lanes for the short rows (negations, constants times a signal, sums of a few terms), 16 for the
// long ones (running sums of MultiOR / CalculateTotal chains flattened over produced signals, Bits2Num outputs ...).
template <int LANES>
__device__ __forceinline__ void zk_o0_rows_body(const u32* __restrict__ rows, u32 n_rows, const u64* __restrict__ row_ptr,
                                                const u32* __restrict__ dst, const u32* __restrict__ src,
                                                const Fr* __restrict__ coef, const u8* __restrict__ kind,
                                                const u8* __restrict__ kept, u64 kept_stride, u8* __restrict__ out, u64 out_stride) {
  const u32 g = blockIdx.x * (256u / LANES) + threadIdx.x / LANES, l = threadIdx.x % LANES;
  const bool live = g < n_rows;
  const u32 r = live ? rows[g] : 0u;
  const Fr* __restrict__ w = (const Fr*)(kept + (u64)blockIdx.y * kept_stride);
  Fr acc = fr_zero();
  if (live)
    for (u64 t = row_ptr[r] + l; t < row_ptr[r + 1]; t += LANES) {
      const Fr x = w[src[t]];
      const u8 k = kind[t];
      if (k == ZK_COEF_ONE) acc = fr_add(acc, x);
      else if (k == ZK_COEF_MINUS_ONE) acc = fr_sub(acc, x);
      else if (!fr_is_zero(x)) {
        const bool one = x.l[0] == 1 && (x.l[1] | x.l[2] | x.l[3]) == 0;
        acc = fr_add(acc, one ? coef[t] : fr_mont_mul(fr_to_mont(x), coef[t]));
      }
    }
#pragma unroll
  for (int off = LANES / 2; off >= 1; off >>= 1) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32 lo = __shfl_down((u32)acc.l[i], off, LANES), hi = __shfl_down((u32)(acc.l[i] >> 32), off, LANES);
      o.l[i] = (u64)lo | ((u64)hi << 32);
    }
    acc = fr_add(acc, o);
  }
  if (live && l == 0) ((Fr*)(out + (u64)blockIdx.y * out_stride))[dst[r]] = acc;
}
__global__ __launch_bounds__(256) void zk_o0_rows_4(const u32* rows, u32 n_rows, const u64* row_ptr, const u32* dst, const u32* src, const Fr* coef,
                                                    const u8* kind, const u8* kept, u64 kept_stride, u8* out, u64 out_stride) {
  zk_o0_rows_body<4>(rows, n_rows, row_ptr, dst, src, coef, kind, kept, kept_stride, out, out_stride);
}
__global__ __launch_bounds__(256) void zk_o0_rows_16(const u32* rows, u32 n_rows, const u64* row_ptr, const u32* dst, const u32* src, const Fr* coef,
                                                     const u8* kind, const u8* kept, u64 kept_stride, u8* out, u64 out_stride) {
  zk_o0_rows_body<16>(rows, n_rows, row_ptr, dst, src, coef, kind, kept, kept_stride, out, out_stride);
}
