// SHA-256 witness kernels for gfx950.
//
//   zk_sha_chain   K1: native SHA-256 chaining states of every block (one lane per
//                      (email, frame)); tiny.
//   zk_sha_expand  K2: one wavefront per (email, SHA block): expands the kept signals
//                      of circomlib's Sha256compression (30,952 field elements, 0.94 MiB)
//                      straight into the block's witness slots.  HBM-write-bound; this is
//                      the roofline-setting kernel.
//
// Reference gate sequence being evaluated: packages/circuits/lib/sha.circom:89-203
// (Sha256General) / :212-292 (Sha256Partial) -> circomlib sha256compression
// (SURVEY.md Appendix A.2).
#include "zkwg_dev.h"
#include "zkwg_kernels.h"

// ------------------------------------------------------------------ K1: chain
__global__ __launch_bounds__(64) void zk_sha_chain(ZkSched s, const u8* __restrict__ in,
                                                  u32* __restrict__ hst, u32 n_emails) {
  u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 e = gid / s.nframes, fi = gid % s.nframes;
  if (e >= n_emails) return;
  const ZkShaFrame& f = s.fr[fi];
  const u8* rec = in + (u64)e * s.in_stride;
  u32* out = hst + ((u64)e * s.hstates_per_email + f.hstate_base) * 8;
  u32 st[8];
  if (f.partial) {
#pragma unroll
    for (int j = 0; j < 8; ++j) st[j] = zk_ldbe32(rec + f.in_pre + 4 * j);
  } else {
    zk_sha256_iv(st);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = st[j];
  for (u32 b = 0; b < f.nblocks; ++b) {
    zk_sha256_compress(st, rec + f.in_data + 64 * b);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(b + 1) * 8 + j] = st[j];
  }
}

// ------------------------------------------------------------------ K2: expand
// Trace of one compression: one u64 per kept bit-group (ZK_TRACE_GROUPS of them);
// bit k of a group = the signal with index k (circomlib bit vectors are LSB-first).
__device__ inline void zk_sha_trace(u64* __restrict__ tr, u32* __restrict__ wsh,
                                    const u32* __restrict__ hin, const u8* __restrict__ blk) {
  for (int t = 0; t < 16; ++t) wsh[t] = zk_ldbe32(blk + 4 * t);
  for (int t = 16; t < 64; ++t) {  // sigmaPlus[t-16]
    u32 x2 = wsh[t - 2], x15 = wsh[t - 15];
    u32 a1 = zk_rotr(x2, 17), b1 = zk_rotr(x2, 19), c1 = x2 >> 10;
    u32 a0 = zk_rotr(x15, 7), b0 = zk_rotr(x15, 18), c0 = x15 >> 3;
    u32 s1 = a1 ^ b1 ^ c1, s0 = a0 ^ b0 ^ c0;
    u64 sum = (u64)s1 + wsh[t - 7] + s0 + wsh[t - 16];
    u64* g = tr + ZK_G_SP + (t - 16) * 5;
    g[0] = s1; g[1] = b1 & c1; g[2] = s0; g[3] = b0 & c0; g[4] = sum;
    wsh[t] = (u32)sum;
  }
  u32 a = hin[0], b = hin[1], c = hin[2], d = hin[3], e = hin[4], f = hin[5], g_ = hin[6], h = hin[7];
  for (int t = 0; t < 64; ++t) {
    u32 ea = zk_rotr(e, 6), eb = zk_rotr(e, 11), ec = zk_rotr(e, 25);
    u32 bs1 = ea ^ eb ^ ec;
    u32 ch = (e & f) ^ (~e & g_);
    u64 t1 = (u64)h + bs1 + ch + ZK_K256[t] + wsh[t];
    u64* g1 = tr + ZK_G_T1 + t * 4;
    g1[0] = ch; g1[1] = bs1; g1[2] = eb & ec; g1[3] = t1;
    u32 aa = zk_rotr(a, 2), ab = zk_rotr(a, 13), ac = zk_rotr(a, 22);
    u32 bs0 = aa ^ ab ^ ac;
    u32 mmid = b & c;
    u32 maj = (a & (b ^ c)) | mmid;
    u64 t2 = (u64)bs0 + maj;
    u64* g2 = tr + ZK_G_T2 + t * 5;
    g2[0] = bs0; g2[1] = ab & ac; g2[2] = maj; g2[3] = mmid; g2[4] = t2;
    u64 sume = (u64)d + (u32)t1;
    u64 suma = (u64)(u32)t1 + (u32)t2;
    tr[ZK_G_SUMA + t] = suma;
    tr[ZK_G_SUME + t] = sume;
    h = g_; g_ = f; f = e; e = (u32)sume; d = c; c = b; b = a; a = (u32)suma;
  }
  u32 fin[8] = {a, b, c, d, e, f, g_, h};
  for (int j = 0; j < 8; ++j) tr[ZK_G_FSUM + j] = (u64)hin[j] + fin[j];
}

// slot (0..ZK_COMP_SLOTS) -> bit value
__device__ __forceinline__ u32 zk_comp_bit(const u64* __restrict__ tr, u32 sl) {
  u32 g, bit;
  if (sl < ZK_SEC_SP_END) {
    u32 i = sl / ZK_SP_SLOTS, r = sl - i * ZK_SP_SLOTS;
    u32 sub = min(r >> 5, 4u);
    g = ZK_G_SP + i * 5 + sub; bit = r - sub * 32;
  } else if (sl < ZK_SEC_T1_END) {
    u32 q = sl - ZK_SEC_SP_END;
    u32 i = q / ZK_T1_SLOTS, r = q - i * ZK_T1_SLOTS;
    u32 sub = min(r >> 5, 3u);
    g = ZK_G_T1 + i * 4 + sub; bit = r - sub * 32;
  } else if (sl < ZK_SEC_T2_END) {
    u32 q = sl - ZK_SEC_T1_END;
    u32 i = q / ZK_T2_SLOTS, r = q - i * ZK_T2_SLOTS;
    u32 sub = min(r >> 5, 4u);
    g = ZK_G_T2 + i * 5 + sub; bit = r - sub * 32;
  } else {
    u32 q = sl - ZK_SEC_T2_END;
    u32 i = q / 33u;
    g = ZK_G_SUMA + i; bit = q - i * 33u;
  }
  return (u32)(tr[g] >> bit) & 1u;
}


__global__ __launch_bounds__(64 * ZK_EXPAND_WAVES) void zk_sha_expand(
    ZkSched s, const u8* __restrict__ in, const u32* __restrict__ hst, uint4* __restrict__ wit,
    u32 n_emails) {
  __shared__ u64 tr_all[ZK_EXPAND_WAVES][ZK_TRACE_GROUPS];
  __shared__ u32 w_all[ZK_EXPAND_WAVES][64];
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u64 unit = (u64)blockIdx.x * ZK_EXPAND_WAVES + wave;
  const u32 e = (u32)(unit / s.total_blocks);
  u32 blk = (u32)(unit % s.total_blocks);
  const bool live = e < n_emails;
  u32 fi = 0;
  if (s.nframes > 1 && blk >= s.fr[0].nblocks) { fi = 1; blk -= s.fr[0].nblocks; }
  const ZkShaFrame& f = s.fr[fi];
  u64* tr = tr_all[wave];
  if (live && lane == 0) {
    const u8* rec = in + (u64)e * s.in_stride;
    const u32* hin = hst + ((u64)e * s.hstates_per_email + f.hstate_base + blk) * 8;
    zk_sha_trace(tr, w_all[wave], hin, rec + f.in_data + 64 * blk);
  }
  __syncthreads();
  if (!live) return;
  uint4* dst = wit + ((u64)e * s.W + f.s_comp + (u64)blk * ZK_COMP_SLOTS) * 2;
  const u32 nchunks = 2u * ZK_COMP_SLOTS;
#pragma unroll 4
  for (u32 c = lane; c < nchunks; c += 64) {
    uint4 v = zk_zero4();
    if ((c & 1u) == 0) v.x = zk_comp_bit(tr, c >> 1);
    zk_st16(dst + c, v);
  }
}

// ------------------------------------------------------------------ per-frame "misc" signals
// Everything of Sha256Bytes[Partial] that is not inside a compression block:
// inBlockIndex, the LessEqThan bits, the 256 ItemAtIndex selectors, the Num2Bits(8)
// of every input byte (and of preHash).  Called by the per-email misc kernels.
// Returns false if one of the frame's constraints fails.
__device__ bool zk_emit_sha_frame(const ZkSched& s, const ZkShaFrame& f, const u8* __restrict__ rec,
                                  const u32* __restrict__ hst_email, const uint4* __restrict__ invtab,
                                  uint4* __restrict__ wit) {
  const u32 len = *(const u32*)(rec + f.in_len);
  const u64 lenbits_v = (u64)len * 8;
  const u32 ibi = (u32)(lenbits_v >> 9);              // inBlockIndex <-- paddedInLength >> 9
  const long long idx = (long long)ibi - 1;           // arraySelectors[k].index
  const u32 NB = f.nblocks;
  const u64 maxbits = (u64)f.max_bytes * 8;
  bool ok = (lenbits_v == (u64)ibi * 512);            // sha.circom:112
  // LessEqThan(lenbits)(paddedInLength, maxBitLength): n2b.in = a + 2^nb - (maxBits+1)
  const long long n2b_in = (long long)lenbits_v + (1ll << f.lenbits) - (long long)(maxbits + 1);
  ok = ok && n2b_in >= 0 && n2b_in < (1ll << f.lenbits);  // fits, and out = 1 - bit[nb] == 1
  ok = ok && idx >= 0 && idx < (long long)NB;         // calcTotalIndex.sum === 1 (array.circom:40)
  const u32 idx_c = (u32)(idx < 0 ? 0 : (idx >= (long long)NB ? NB - 1 : idx));
  const u32* hsel = hst_email + (f.hstate_base + idx_c + 1) * 8;  // state after block idx

  if (threadIdx.x == 0) {
    zk_st16(wit + f.s_inBlockIndex * 2, zk_small(ibi));
    zk_st16(wit + f.s_inBlockIndex * 2 + 1, zk_zero4());
  }
  zk_emit_small(wit, f.s_lenbits, f.lenbits + 1, [&](u32 i) { return (u32)((u64)n2b_in >> i) & 1u; });
  // selectors: per output bit k: nums[NB], then NB x (isz.out, isz.inv)
  const u32 per = 3 * NB;
  const long long half_tab = (long long)(s.inv_table_len / 2);
  zk_emit(wit, f.s_sel, 256 * per, [&](u32 sl, u32 half) {
    u32 k = sl / per, r = sl - k * per;
    if (r < NB) {
      u32 bit = (hsel[k >> 5] >> (31 - (k & 31))) & 1u;  // out[] is MSB-first per word
      return (half == 0 && (long long)r == idx) ? zk_small(bit) : zk_zero4();
    }
    u32 q = r - NB, j = q >> 1;
    if ((q & 1u) == 0) return (half == 0 && (long long)j == idx) ? zk_small(1u) : zk_zero4();
    long long d = idx - (long long)j;                    // isz.in = index - j
    if (d > half_tab) d = half_tab;
    if (d < -half_tab) d = -half_tab;
    return invtab[(u64)(d + half_tab) * 2 + half];
  });
  zk_emit_small(wit, f.s_bytes, f.max_bytes * 8, [&](u32 sl) {
    return (u32)(rec[f.in_data + (sl >> 3)] >> (sl & 7)) & 1u;
  });
  if (f.partial) {
    zk_emit_small(wit, f.s_states, 32 * 8, [&](u32 sl) {
      return (u32)(rec[f.in_pre + (sl >> 3)] >> (sl & 7)) & 1u;
    });
  }
  return ok;
}

// ------------------------------------------------------------------ main = Sha256Bytes(N)
// (reference: packages/circuits/tests/test-circuits/sha-test.circom:5)
__global__ __launch_bounds__(256) void zk_misc_sha_main(ZkSched s, const u8* __restrict__ in,
                                                       const u32* __restrict__ hst,
                                                       const uint4* __restrict__ invtab,
                                                       uint4* __restrict__ wit_all,
                                                       int* __restrict__ status, u32 n_emails) {
  const u32 e = blockIdx.x;
  if (e >= n_emails) return;
  const u8* rec = in + (u64)e * s.in_stride;
  const u32* hst_email = hst + (u64)e * s.hstates_per_email * 8;
  uint4* wit = wit_all + (u64)e * s.W * 2;
  const ZkShaFrame& f = s.fr[0];
  if (threadIdx.x == 0) {
    zk_st16(wit, zk_small(1u));
    zk_st16(wit + 1, zk_zero4());
  }
  const u32 len = *(const u32*)(rec + f.in_len);
  u32 ibi = (u32)(((u64)len * 8) >> 9);
  u32 idx_c = ibi == 0 ? 0 : (ibi > f.nblocks ? f.nblocks - 1 : ibi - 1);
  const u32* hsel = hst_email + (f.hstate_base + idx_c + 1) * 8;
  zk_emit_small(wit, s.s_out, 256, [&](u32 k) { return (hsel[k >> 5] >> (31 - (k & 31))) & 1u; });
  zk_emit_small(wit, s.s_pub_in, f.max_bytes, [&](u32 i) { return (u32)rec[f.in_data + i]; });
  if (threadIdx.x == 0) {
    zk_st16(wit + (s.s_pub_in + f.max_bytes) * 2, zk_small(len));
    zk_st16(wit + (s.s_pub_in + f.max_bytes) * 2 + 1, zk_zero4());
  }
  bool ok = zk_emit_sha_frame(s, f, rec, hst_email, invtab, wit);
  if (threadIdx.x == 0 && !ok) status[e] = 4;
}
