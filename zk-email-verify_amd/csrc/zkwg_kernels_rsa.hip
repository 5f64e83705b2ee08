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
#include "zkwg_fpmul_core.h"

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

// zk_poseidon9_g16 -- the same permutation with 16 LANES per email (4 emails per wavefront): lane j < 10 owns state
// element j, kept in MONTGOMERY form so that the S-box is 3 dependent products (x^2, x^4, x^5) instead of the 5 of the
// standard-form lane-per-email kernel.  Every step is one Montgomery product executed by all lanes with lane-specific
// operands, so nothing serialises: in a partial round lane 0 squares while lanes 1..9 form their first-row products
// u_j s_j and lanes 10..12 convert the previous round's S-box signals back to standard form for the image; the
// first-row dot product is summed by lane 0 from LDS.  A partial round is 4 dependent products (24 for one lane), a
// full round 6 + the dense mix: ~0.35 k dependent products per permutation instead of ~1.9 k, at ~3 x the issue
// slots -- the kernel is latency-bound either way (a few hundred wavefronts on 1,024 SIMDs).
// Tables: B.pos_c = the sparse-round table (zkwg_poseidon_sparse.h), B.pos_m + 780 = its additive constants in
// Montgomery form (c_first[4][10] | c_part[60][10] | c_last[4][10]).
__global__ __launch_bounds__(64) void zk_poseidon9_g16(ZkSched s, ZkBufs B) {
  constexpr u32 T = 10, RP = 60;
  __shared__ Fr sh_y[4][4];     // per email group: x5, x2, x4 of lane 0 (Montgomery)
  __shared__ Fr sh_p[4][16];    // per email group: one value per lane (first-row products / the state for a dense mix)
  const u32 lane = threadIdx.x, g = lane >> 4, j = lane & 15u;
  const u32 e_raw = blockIdx.x * 4u + g;
  const bool live = e_raw < B.n_emails;
  const u32 e = live ? e_raw : B.n_emails - 1u;      // idle groups shadow the last email (no stores)
  const u8* rec = B.in + (u64)e * s.in_stride;
  Fr* frv = B.frv + (u64)e * s.img_fr;
  Fr* emit = frv + s.f_pos;
  const Fr* tab = B.pos_c;
  const Fr* mt = tab + 4 * T;
  const Fr* s_part = mt + T * T + RP * T;
  const Fr* bt = s_part + RP * (2 * T - 1);
  const Fr* cm_first = B.pos_m + 780;
  const Fr* cm_part = cm_first + 4 * T;
  const Fr* cm_last = cm_part + RP * T;
  const Fr one = fr_from_u64(1);
  // poseidonInput[i] = in[2i] + 2^121 in[2i+1] (i < 8), in[16] (i = 8); state = [0, inputs]
  Fr x = fr_zero();
  if (j >= 1 && j < T) {
    const u32 i = j - 1;
    const u64 top_mask = (1ull << 57) - 1;
    const u64* lo = (const u64*)(rec + s.rsa.in_mod + 16 * (2 * i));
    x = Fr{{lo[0], lo[1] & top_mask, 0, 0}};
    if (i < 8) {
      const u64* hp = (const u64*)(rec + s.rsa.in_mod + 16 * (2 * i + 1));
      const u64 h0 = hp[0], h1 = hp[1] & top_mask;
      x.l[1] |= h0 << 57;
      x.l[2] = (h0 >> 7) | (h1 << 57);
      x.l[3] = h1 >> 7;
    }
  }
  x = fr_mont_mul(x, fr_R2());
  Fr pend = fr_zero();           // lanes 10..12: an S-box signal of lane 0 waiting for its conversion
  u32 pend_idx = 0xffffffffu;
  // dense mix: x_i = sum_j x_j mat[j*T + i]
  auto dense = [&](const Fr* __restrict__ mat) {
    sh_p[g][j] = x;
    __syncthreads();
    if (j < T) {
      FrWide w;
      fr_wide_zero(w);
      for (u32 k = 0; k < T; ++k) fr_wide_mac(w, sh_p[g][k], mat[k * T + j]);
      x = fr_wide_redc(w);
    }
    __syncthreads();
  };
  for (u32 half = 0; half < 2; ++half) {
    if (half == 1) {
      for (u32 k = 0; k < RP; ++k) {
        const Fr* sk = s_part + k * (2 * T - 1);
        if (j < T) x = fr_add(x, cm_part[k * T + j]);
        // A: lane 0: x^2 | lanes 1..9: u_j * s_j | lanes 10..12: previous signals -> standard form
        const Fr r1 = fr_mont_mul(j < T ? x : pend, j == 0 ? x : (j < T ? sk[j] : one));
        if (j >= T && pend_idx != 0xffffffffu && live) emit[pend_idx] = r1;
        sh_p[g][j] = r1;
        const Fr r2 = fr_mont_mul(r1, r1);          // B: lane 0: x^4
        const Fr r3 = fr_mont_mul(r2, x);           // C: lane 0: x^5
        if (j == 0) { sh_y[g][0] = r3; sh_y[g][1] = r1; sh_y[g][2] = r2; }
        __syncthreads();
        const Fr y0 = sh_y[g][0];
        if (j >= T && j < T + 3) { pend = sh_y[g][j - T]; pend_idx = 3u * (8u * T + k) + (j - T); }
        // D: lane 0: y0 * n00 | lanes 1..9: y0 * w_j
        const Fr r4 = fr_mont_mul(y0, j == 0 ? sk[0] : (j < T ? sk[T - 1 + j] : one));
        if (j == 0) {
          Fr acc = r4;
          for (u32 q = 1; q < T; ++q) acc = fr_add(acc, sh_p[g][q]);
          x = acc;
        } else if (j < T) x = fr_add(x, r4);
        __syncthreads();
      }
      dense(bt);   // z = B u
    }
    for (u32 r = 0; r < 4; ++r) {
      const Fr* c = (half ? cm_last : cm_first) + r * T;
      if (j < T) x = fr_add(x, c[j]);
      const Fr a0 = j < T ? x : pend;
      const Fr x2 = fr_mont_mul(a0, j < T ? x : one);       // lanes 10..12 flush a pending conversion
      if (j >= T && pend_idx != 0xffffffffu && live) { emit[pend_idx] = x2; pend_idx = 0xffffffffu; }
      const Fr x4 = fr_mont_mul(x2, x2);
      const Fr x5 = fr_mont_mul(x4, x);
      const Fr e5 = fr_mont_mul(x5, one), e2 = fr_mont_mul(x2, one), e4 = fr_mont_mul(x4, one);
      if (j < T && live) {
        Fr* o = emit + 3u * ((half * 4u + r) * T + j);
        o[0] = e5; o[1] = e2; o[2] = e4;
      }
      if (j < T) x = x5;
      dense(mt);
    }
  }
  const Fr h = fr_mont_mul(x, one);
  if (j == 0 && live) frv[s.f_out] = h;
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

// main = FpMul(n, k) with small parameters (fp-mul-test.circom: FpMul(2,4)): one lane per email (zkwg_fpmul_core.h)
__global__ __launch_bounds__(64) void zk_fpmul_small(ZkSched s, ZkBufs B) {
  const u32 e = blockIdx.x * 64u + threadIdx.x;
  if (e >= B.n_emails) return;
  const u8* rec = B.in + (u64)e * s.in_stride;
  int st = zk_fpmul_small_core(s.fpg, s.m_one, rec, B.bits + (u64)e * s.img_bits, B.small + (u64)e * s.img_small, B.frv + (u64)e * s.img_fr);
  if (*(const u32*)(rec + s.in_off[ZK_IN_RANGE_FLAGS]) != 0) st = 4;   // generic input path: a value that did not fit its chunk
  if (st) B.status[e] = st;
}
