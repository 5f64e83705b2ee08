// SHA-256 compute kernels for gfx950 (they fill the compact image; zk_expand writes the
// witness).
//
//   zk_sha_chain   native SHA-256 chaining state of every block, one lane per
//                  (email, frame); also the frame-level scalars (inBlockIndex, selector
//                  index, selected digest, LessEqThan input) and the frame's assertions.
//   zk_sha_trace   one lane per (email, SHA block): the 952 bit-groups that make up the
//                  30,952 kept signals of circomlib's Sha256compression for that block.
//
// Reference gate sequence: packages/circuits/lib/sha.circom:17-38 (Sha256Bytes), :47-80
// (Sha256BytesPartial), :89-203 (Sha256General), :212-292 (Sha256Partial) and circomlib
// sha256compression.
#include "zkwg_dev.h"
#include "zkwg_kernels.h"

// ------------------------------------------------------------------ chain
__global__ __launch_bounds__(64) void zk_sha_chain(ZkSched s, ZkBufs B) {
  u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 e = gid / s.nframes, fi = gid % s.nframes;
  if (e >= B.n_emails) return;
  const ZkShaFrame& f = s.fr[fi];
  const u8* rec = B.in + (u64)e * s.in_stride;
  u32* out = B.hst + ((u64)e * s.hstates_per_email + f.hstate_base) * 8;
  u64* bits = B.bits + (u64)e * s.img_bits;
  u32* small = B.small + (u64)e * s.img_small;

  // frame scalars + assertions (sha.circom:111-129, utils/array.circom:40)
  const u32 len = *(const u32*)(rec + f.in_len);
  const u64 lenbits_v = (u64)len * 8;                  // sha.paddedInLength <== paddedInLength * 8
  const u32 ibi = (u32)(lenbits_v >> 9);               // inBlockIndex <-- paddedInLength >> 9
  const long long idx = (long long)ibi - 1;            // arraySelectors[k].index
  const u32 NB = f.nblocks;
  const u64 maxbits = (u64)f.max_bytes * 8;
  bool ok = (lenbits_v == (u64)ibi * 512);
  // LessEqThan(nb)(a, maxBits) -> LessThan(nb)(a, maxBits+1): n2b.in = a + 2^nb - (maxBits+1)
  const long long n2b_in = (long long)lenbits_v + (1ll << f.lenbits) - (long long)(maxbits + 1);
  ok = ok && n2b_in >= 0 && n2b_in < (1ll << f.lenbits);
  ok = ok && idx >= 0 && idx < (long long)NB;
  const u32 idx_c = (u32)(idx < 0 ? 0 : (idx >= (long long)NB ? NB - 1 : idx));
  small[f.m_ibi] = ibi;
  small[f.m_idx] = (u32)(int)idx;  // ibi < 2^26, so idx fits an i32
  bits[f.b_lenbits] = (u64)n2b_in;
  if (fi == 0) small[s.m_one] = 1;
  if (f.m_len != ~0u) small[f.m_len] = len;
  if (f.azp) {
    // Num2Bits(log2Ceil(max))(length) (email-verifier.circom:58-59,116-117) and
    // AssertZeroPadding(max)(data, length) (utils/array.circom:149-164)
    small[f.m_len_m1] = (u32)((int)len - 1);
    bits[f.b_len] = len;
    u32 bl = 0;
    for (u32 n = f.max_bytes - 1; n > 0; n >>= 1) ++bl;
    ok = ok && len < (1u << bl);   // (the zero-padding scan itself is spread over zk_sha_trace's lanes)
  }
  // generic input path: an element that did not fit its packed slot fails the circuit's own range
  // check of that signal (Num2Bits(8) / Num2Bits(121) / AssertBit / ..., include/zkwg.h zkwg_pack_field)
  if (fi == 0 && *(const u32*)(rec + s.in_off[ZK_IN_RANGE_FLAGS]) != 0) ok = false;
  if (!ok) B.status[e] = 4;

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
    if (b == idx_c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) small[f.m_digest + j] = st[j];
      // the same digest as one LSB-first 256-bit group: bit k = out[k] (MSB-first words)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bits[f.b_digest + j] = (u64)__builtin_bitreverse32(st[2 * j]) |
                               ((u64)__builtin_bitreverse32(st[2 * j + 1]) << 32);
      if (fi == 0 && s.main_kind == 0) {
        // shaHi / shaLo = PackBits(256,128)(sha) (email-verifier.circom:68-71): big-endian halves
        Fr* frv = B.frv + (u64)e * s.img_fr;
        frv[s.f_out + 1] = Fr{{((u64)st[2] << 32) | st[3], ((u64)st[0] << 32) | st[1], 0, 0}};
        frv[s.f_out + 2] = Fr{{((u64)st[6] << 32) | st[7], ((u64)st[4] << 32) | st[5], 0, 0}};
      }
    }
  }
}

// ------------------------------------------------------------------ trace
// One u64 per kept bit-group; bit k of a group = the signal with index k (circomlib bit
// vectors are LSB-first).  Group order = layout order inside the block (zkwg_sched.h).
__global__ __launch_bounds__(64) void zk_sha_trace(ZkSched s, ZkBufs B) {
  const u64 unit = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 e = (u32)(unit / s.total_blocks);
  if (e >= B.n_emails) return;
  u32 blk = (u32)(unit % s.total_blocks);
  u32 fi = 0;
  if (s.nframes > 1 && blk >= s.fr[0].nblocks) { fi = 1; blk -= s.fr[0].nblocks; }
  const ZkShaFrame& f = s.fr[fi];
  const u8* blkp = B.in + (u64)e * s.in_stride + f.in_data + 64 * blk;
  const u32* hin = B.hst + ((u64)e * s.hstates_per_email + f.hstate_base + blk) * 8;
  u64* tr = B.bits + (u64)e * s.img_bits + f.b_trace + (u64)blk * ZK_TRACE_GROUPS;

  u32 w[64];
  zk_load_block_be(w, blkp);
  if (f.azp) {
    // AssertZeroPadding(max)(data, length) (utils/array.circom:149-164): bytes at index >= length are zero;
    // every lane checks the 64 bytes of its own block
    const u32 len = *(const u32*)(B.in + (u64)e * s.in_stride + f.in_len);
    const u32 base = 64u * blk;
    bool bad = false;
    if (len <= base) {
#pragma unroll
      for (int t = 0; t < 16; ++t) bad = bad || (w[t] != 0);
    } else if (len < base + 64u) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const u32 first = base + 4u * t;                 // message word t holds bytes first..first+3, big-endian
        u32 mask = 0;
        if (len <= first) mask = 0xffffffffu;
        else if (len < first + 4u) mask = 0xffffffffu >> (8u * (len - first));
        bad = bad || ((w[t] & mask) != 0);
      }
    }
    if (bad) B.status[e] = 4;
  }
  if ((fi == 0 && s.mask_header) || (fi == 1 && s.mask_body)) {
    // ByteMask.bit_check[i] = AssertBit(): mask[i] * (mask[i] - 1) === 0  (utils/bytes.circom:155-158)
    const u8* mk = B.in + (u64)e * s.in_stride + s.in_off[fi == 0 ? 9 : 10] + 64u * blk;
    bool bad = false;
    for (int t = 0; t < 64; ++t) bad = bad || (mk[t] > 1);
    if (bad) B.status[e] = 4;
  }
#pragma unroll
  for (int t = 16; t < 64; ++t) {  // sigmaPlus[t-16]
    u32 x2 = w[t - 2], x15 = w[t - 15];
    u32 a1 = zk_rotr(x2, 17), b1 = zk_rotr(x2, 19), c1 = x2 >> 10;
    u32 a0 = zk_rotr(x15, 7), b0 = zk_rotr(x15, 18), c0 = x15 >> 3;
    u32 s1 = a1 ^ b1 ^ c1, s0 = a0 ^ b0 ^ c0;
    u64 sum = (u64)s1 + w[t - 7] + s0 + w[t - 16];
    u64* g = tr + ZK_G_SP + (t - 16) * 5;
    g[0] = s1; g[1] = b1 & c1; g[2] = s0; g[3] = b0 & c0; g[4] = sum;
    w[t] = (u32)sum;
  }
  const u32 h0 = hin[0], h1 = hin[1], h2 = hin[2], h3 = hin[3], h4 = hin[4], h5 = hin[5], h6 = hin[6], h7 = hin[7];
  u32 a = h0, b = h1, c = h2, d = h3, e_ = h4, f_ = h5, g_ = h6, h = h7;
#pragma unroll
  for (int t = 0; t < 64; ++t) {
    u32 eb = zk_rotr(e_, 11), ec = zk_rotr(e_, 25);
    u32 bs1 = zk_rotr(e_, 6) ^ eb ^ ec;
    u32 ch = (e_ & f_) ^ (~e_ & g_);
    u64 t1 = (u64)h + bs1 + ch + ZK_K256[t] + w[t];
    u64* g1 = tr + ZK_G_T1 + t * 4;
    g1[0] = ch; g1[1] = bs1; g1[2] = eb & ec; g1[3] = t1;
    u32 ab = zk_rotr(a, 13), ac = zk_rotr(a, 22);
    u32 bs0 = zk_rotr(a, 2) ^ ab ^ ac;
    u32 mmid = b & c;
    u32 maj = (a & (b ^ c)) | mmid;
    u64 t2 = (u64)bs0 + maj;
    u64* g2 = tr + ZK_G_T2 + t * 5;
    g2[0] = bs0; g2[1] = ab & ac; g2[2] = maj; g2[3] = mmid; g2[4] = t2;
    u64 sume = (u64)d + (u32)t1;
    u64 suma = (u64)(u32)t1 + (u32)t2;
    tr[ZK_G_SUMA + t] = suma;
    tr[ZK_G_SUME + t] = sume;
    h = g_; g_ = f_; f_ = e_; e_ = (u32)sume; d = c; c = b; b = a; a = (u32)suma;
  }
  tr[ZK_G_FSUM + 0] = (u64)h0 + a; tr[ZK_G_FSUM + 1] = (u64)h1 + b;
  tr[ZK_G_FSUM + 2] = (u64)h2 + c; tr[ZK_G_FSUM + 3] = (u64)h3 + d;
  tr[ZK_G_FSUM + 4] = (u64)h4 + e_; tr[ZK_G_FSUM + 5] = (u64)h5 + f_;
  tr[ZK_G_FSUM + 6] = (u64)h6 + g_; tr[ZK_G_FSUM + 7] = (u64)h7 + h;
}
