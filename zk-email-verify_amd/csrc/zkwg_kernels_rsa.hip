// zk_rsa -- RSAVerifier65537(121,17): one wavefront per email (zkwg_rsa_wave.h, the wavefront-parallel
//           device path; zkwg_rsa_core.h is the phase-sequential restatement the host tests run).
// zk_poseidon9 -- pubkeyHash = PoseidonLarge(121,17)(pubkey) -> Poseidon(9): one LANE per email with the
//           sparse partial rounds of zkwg_poseidon_sparse.h (3.5 k multiplier issues per email instead of
//           43 k for the wave-collective dense rounds it replaces; latency-bound, tiny in issue slots).
#include "zkwg_dev.h"
#include "zkwg_kernels.h"
#include "zkwg_rsa_wave.h"
#include "zkwg_poseidon_sparse.h"
#include "zkwg_poseidon_core.h"

__global__ __launch_bounds__(64) void zk_rsa(ZkSched s, ZkBufs B) {
  __shared__ ZkRsaLds S;
  const u32 e = blockIdx.x;
  if (e >= B.n_emails) return;
  const u8* rec = B.in + (u64)e * s.in_stride;
  u64* bits = B.bits + (u64)e * s.img_bits;
  u32* small = B.small + (u64)e * s.img_small;
  Fr* frv = B.frv + (u64)e * s.img_fr;
  const u32* digest = s.rsa.msg_from_digest ? small + s.rsa.m_digest : nullptr;
  if (s.main_kind == 2 && threadIdx.x == 0) small[s.m_one] = 1;  // no SHA chain kernel in this main
  zkw_rsa_email(S, s.rsa, rec, digest, bits, small, frv);
  // generic input path (RSA main has no SHA chain kernel to look at the range flags)
  const bool range_bad = s.main_kind == 2 && *(const u32*)(rec + s.in_off[ZK_IN_RANGE_FLAGS]) != 0;
  if (threadIdx.x == 0 && (!S.ok || range_bad)) B.status[e] = 4;
}

// pubkeyHash <== PoseidonLarge(n, k)(pubkey)   (email-verifier.circom:173, utils/hash.circom:15-39):
// poseidonInput[i] = in[2i] + 2^121 in[2i+1] (i < 8), in[16] (i = 8); state = [0, inputs]
__global__ __launch_bounds__(64) void zk_poseidon9(ZkSched s, ZkBufs B) {
  __shared__ Fr st[10 * 64];
  const u32 lane = threadIdx.x;
  const u32 e = blockIdx.x * 64 + lane;
  if (e >= B.n_emails) return;
  const u8* rec = B.in + (u64)e * s.in_stride;
  Fr* frv = B.frv + (u64)e * s.img_fr;
  const u64 top_mask = (1ull << 57) - 1;
  Fr* stl = st + lane;
  stl[0] = fr_zero();
#pragma unroll 1
  for (u32 i = 0; i < 9; ++i) {
    const u64* lo = (const u64*)(rec + s.rsa.in_mod + 16 * (2 * i));
    Fr v{{lo[0], lo[1] & top_mask, 0, 0}};
    if (i < 8) {
      const u64* hp = (const u64*)(rec + s.rsa.in_mod + 16 * (2 * i + 1));
      const u64 h0 = hp[0], h1 = hp[1] & top_mask;
      v.l[1] |= h0 << 57;
      v.l[2] = (h0 >> 7) | (h1 << 57);
      v.l[3] = h1 >> 7;
    }
    stl[(1 + i) * 64] = v;
  }
  Fr tmp[10];
  const Fr h = zk_poseidon_sparse<10>(stl, 64, B.pos_c, 60, frv + s.f_pos, tmp, 1);
  frv[s.f_out] = h;
}

// Small batches (a few hundred emails): one lane per email leaves the chip empty and costs ~3.6 ms of
// latency; here one WAVEFRONT per email runs the dense rounds wave-collectively (zkwg_poseidon_core.h,
// ~0.7 ms, 12 x the multiplier issues -- irrelevant when the batch cannot fill the chip anyway).
// B.pos_m: C[680] then M[100] in Montgomery form.
__global__ __launch_bounds__(64) void zk_poseidon9_wave(ZkSched s, ZkBufs B) {
  __shared__ ZkPosLds PS;
  __shared__ u64 limb[ZK_RSA_K][2];
  const u32 e = blockIdx.x;
  if (e >= B.n_emails) return;
  const u8* rec = B.in + (u64)e * s.in_stride;
  Fr* frv = B.frv + (u64)e * s.img_fr;
  const u64 top_mask = (1ull << 57) - 1;
  if (threadIdx.x < ZK_RSA_K) {
    const u64* pm = (const u64*)(rec + s.rsa.in_mod + 16 * threadIdx.x);
    limb[threadIdx.x][0] = pm[0];
    limb[threadIdx.x][1] = pm[1] & top_mask;
  }
  __syncthreads();
  zk_poseidon_large(PS, limb, B.pos_m, B.pos_m + 680, frv + s.f_pos, frv + s.f_out);
}
