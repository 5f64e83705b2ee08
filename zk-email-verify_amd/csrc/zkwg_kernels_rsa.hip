// zk_rsa -- RSAVerifier65537(121,17): one wavefront per email (see zkwg_rsa_core.h).
#include "zkwg_dev.h"
#include "zkwg_kernels.h"
#include "zkwg_rsa_core.h"
#include "zkwg_poseidon_core.h"

__global__ __launch_bounds__(64) void zk_rsa(ZkSched s, ZkBufs B) {
  __shared__ ZkRsaLds S;
  __shared__ u32 lt_eq[40];
  const u32 e = blockIdx.x;
  if (e >= B.n_emails) return;
  const u8* rec = B.in + (u64)e * s.in_stride;
  u64* bits = B.bits + (u64)e * s.img_bits;
  u32* small = B.small + (u64)e * s.img_small;
  Fr* frv = B.frv + (u64)e * s.img_fr;
  const u32* digest = s.rsa.msg_from_digest ? small + s.rsa.m_digest : nullptr;
  if (s.main_kind == 2 && threadIdx.x == 0) small[s.m_one] = 1;  // no SHA chain kernel in this main
  zk_rsa_email(S, s.rsa, rec, digest, bits, small, frv, lt_eq);
  // generic input path (RSA main has no SHA chain kernel to look at the range flags)
  const bool range_bad = s.main_kind == 2 && *(const u32*)(rec + s.in_off[ZK_IN_RANGE_FLAGS]) != 0;
  if (threadIdx.x == 0 && (!S.ok || range_bad)) B.status[e] = 4;
  if (s.main_kind == 0) {
    // pubkeyHash <== PoseidonLarge(n, k)(pubkey)   (email-verifier.circom:173)
    __shared__ ZkPosLds PS;
    zk_poseidon_large(PS, S.p121, B.pos_c, B.pos_m, frv + s.f_pos, frv + s.f_out);
  }
}
