// Host-side schedule construction: zkwg_config -> ZkSched + segment table.
#pragma once
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/zkwg.h"
#include "zkwg_layout.h"

static bool build_sched(const zkwg_config& cfg, ZkSched& s, std::vector<ZkSeg>& segs,
                        std::vector<u32>& first_seg, u32 portion = ZK_PORTION_DEFAULT) {
  memset(&s, 0, sizeof(s));
  if (cfg.layout != ZKWG_LAYOUT_KEPT_V1) return false;
  if (cfg.remove_soft_line_breaks && (cfg.main_kind != ZKWG_MAIN_EMAIL_VERIFIER || cfg.ignore_body_hash_check)) return false;
  if ((cfg.enable_header_masking || cfg.enable_body_masking) && cfg.main_kind != ZKWG_MAIN_EMAIL_VERIFIER) return false;
  if (cfg.enable_body_masking && cfg.ignore_body_hash_check) return false;
  if (cfg.max_header % 64 != 0 || cfg.max_body % 64 != 0) return false;
  s.main_kind = cfg.main_kind;
  s.n = cfg.n; s.k = cfg.k; s.ignore_body = cfg.ignore_body_hash_check;
  // input record
  u32 off = 0;
  s.in_off[ZKWG_IN_HEADER] = off; off += cfg.max_header;
  s.in_off[ZKWG_IN_BODY] = off; off += cfg.max_body;
  s.in_off[ZKWG_IN_PRECOMPUTED_SHA] = off; off += 32;
  off = (off + 15u) & ~15u;
  s.in_off[ZKWG_IN_PUBKEY] = off; off += 17 * 16;
  s.in_off[ZKWG_IN_SIGNATURE] = off; off += 17 * 16;
  s.in_off[ZKWG_IN_MESSAGE] = off; off += 17 * 16;
  s.in_off[ZKWG_IN_HEADER_LEN] = off; off += 4;
  s.in_off[ZKWG_IN_BODY_LEN] = off; off += 4;
  s.in_off[ZKWG_IN_BODY_HASH_INDEX] = off; off += 4;
  off = (off + 15u) & ~15u;
  s.in_off[ZKWG_IN_HEADER_MASK] = off; off += cfg.enable_header_masking ? cfg.max_header : 0;
  s.in_off[ZKWG_IN_BODY_MASK] = off; off += cfg.enable_body_masking ? cfg.max_body : 0;
  off = (off + 15u) & ~15u;
  s.in_off[ZKWG_IN_DECODED_BODY] = off; off += cfg.remove_soft_line_breaks ? cfg.max_body : 0;
  s.in_stride = (off + 15u) & ~15u;
  s.rslb = cfg.remove_soft_line_breaks ? 1u : 0u;
  s.mask_header = cfg.enable_header_masking ? 1u : 0u;
  s.mask_body = cfg.enable_body_masking ? 1u : 0u;

  auto init_frame = [&](ZkShaFrame& f, u32 max_bytes, u32 partial, u32 in_data, u32 in_len) {
    f.max_bytes = max_bytes;
    f.nblocks = max_bytes / 64;
    f.lenbits = zk_log2ceil((u64)max_bytes * 8);
    f.partial = partial;
    f.in_data = in_data; f.in_len = in_len; f.in_pre = s.in_off[ZKWG_IN_PRECOMPUTED_SHA];
    f.m_len = f.m_len_m1 = f.b_len = ~0u;
    f.azp = 0;
    f.hstate_base = s.hstates_per_email;
    f.block_base = s.total_blocks;
    s.hstates_per_email += f.nblocks + 1;
    s.total_blocks += f.nblocks;
  };

  ZkWalker w;
  u64 max_small = 256;  // largest |d| whose inverse zk_expand looks up
  switch (cfg.main_kind) {
    case ZKWG_MAIN_SHA256_BYTES:
      if (cfg.max_header == 0) return false;
      s.nframes = 1;
      init_frame(s.fr[0], cfg.max_header, 0, s.in_off[ZKWG_IN_HEADER], s.in_off[ZKWG_IN_HEADER_LEN]);
      zk_walk_main_sha(w, s);
      max_small = std::max<u64>(max_small, s.fr[0].nblocks + 2);
      break;
    case ZKWG_MAIN_EMAIL_VERIFIER:
      if (cfg.n != 121 || cfg.k != 17 || cfg.max_header < 128) return false;
      s.body = cfg.ignore_body_hash_check ? 0 : 1;
      if (s.body && cfg.max_body == 0) return false;
      s.nframes = s.body ? 2 : 1;
      init_frame(s.fr[0], cfg.max_header, 0, s.in_off[ZKWG_IN_HEADER], s.in_off[ZKWG_IN_HEADER_LEN]);
      if (s.body) init_frame(s.fr[1], cfg.max_body, 1, s.in_off[ZKWG_IN_BODY], s.in_off[ZKWG_IN_BODY_LEN]);
      zk_walk_main_ev(w, s);
      max_small = std::max<u64>(max_small, 2100);
      max_small = std::max<u64>(max_small, (u64)cfg.max_header + 64);
      max_small = std::max<u64>(max_small, std::max(s.fr[0].nblocks, s.fr[1].nblocks) + 2);
      break;
    case ZKWG_MAIN_RSA_VERIFIER:
      if (cfg.n != 121 || cfg.k != 17) return false;
      s.nframes = 0;
      zk_walk_main_rsa(w, s);
      max_small = std::max<u64>(max_small, 2100);
      break;
    default:
      return false;
  }
  if (w.seg_cur != w.cur) return false;  // the segment table must tile the witness exactly
  s.W = w.cur;
  s.inv_half = (u32)max_small;
  s.img_bits = w.nbits + 1;
  s.img_small = (w.nsmall + 3u) & ~3u;
  s.img_fr = w.nfr + 1;
  segs = std::move(w.segs);
  s.nsegs = (u32)segs.size();
  s.portion = portion;
  s.nportions = (u32)((s.W + portion - 1) / portion);
  first_seg.assign(s.nportions, 0);
  u32 si = 0;
  for (u32 p = 0; p < s.nportions; ++p) {
    u64 slot0 = (u64)p * portion;
    while (si + 1 < s.nsegs && segs[si].slot + segs[si].nslots <= slot0) ++si;
    first_seg[p] = si;
  }
  return true;
}


// ------------------------------------------------------------------ Poseidon(9) constants
// Regenerated from the published procedure of the Poseidon reference implementation
// (circomlib's poseidon_constants.circom is [EXT], absent): Grain LFSR seeded with
// (field=1, sbox=0, n=254, t, R_F=8, R_P), round constants by rejection sampling, MDS =
// Cauchy matrix 1/(x_i + y_j) with x, y the next 2t values of the stream (reduced mod r).
// Output in Montgomery form: C[(R_F+R_P)*t], M[t*t] row-major.
struct ZkGrain {
  u8 st[80];
  int head = 0;
  ZkGrain(u32 t, u32 rf, u32 rp) {
    int k = 0;
    auto put = [&](u32 v, int w) { for (int i = w - 1; i >= 0; --i) st[k++] = (v >> i) & 1; };
    put(1, 2); put(0, 4); put(254, 12); put(t, 12); put(rf, 10); put(rp, 10);
    while (k < 80) st[k++] = 1;
    for (int i = 0; i < 160; ++i) step();
  }
  u8 step() {
    auto g = [&](int i) { return st[(head + i) % 80]; };
    u8 nb = g(62) ^ g(51) ^ g(38) ^ g(23) ^ g(13) ^ g(0);
    st[head] = nb;
    head = (head + 1) % 80;
    return nb;
  }
  u8 next() {
    u8 nb = step();
    while (nb == 0) { step(); nb = step(); }
    return step();
  }
  Fr bits254() {  // 254 bits, MSB first
    Fr v = fr_zero();
    for (int i = 0; i < 254; ++i) {
      v.l[3] = (v.l[3] << 1) | (v.l[2] >> 63);
      v.l[2] = (v.l[2] << 1) | (v.l[1] >> 63);
      v.l[1] = (v.l[1] << 1) | (v.l[0] >> 63);
      v.l[0] = (v.l[0] << 1) | next();
    }
    return v;
  }
};

static inline Fr zk_host_fr_inv(const Fr& a_std) {
  return fr_from_mont(fr_mont_inv(fr_to_mont(a_std)));
}

static inline void build_poseidon_constants(u32 t, u32 rf, u32 rp, std::vector<Fr>& C, std::vector<Fr>& M) {
  ZkGrain g(t, rf, rp);
  C.clear();
  while (C.size() < (size_t)(rf + rp) * t) {
    Fr v = g.bits254();
    if (!fr_geq(v, fr_p())) C.push_back(fr_to_mont(v));
  }
  std::vector<Fr> xy(2 * t);
  for (u32 i = 0; i < 2 * t; ++i) {
    Fr v = g.bits254();
    while (fr_geq(v, fr_p())) { u64 bw; v = fr_sub_raw(v, fr_p(), bw); }
    xy[i] = v;
  }
  M.assign((size_t)t * t, fr_zero());
  for (u32 i = 0; i < t; ++i)
    for (u32 j = 0; j < t; ++j) M[i * t + j] = fr_to_mont(zk_host_fr_inv(fr_add(xy[i], xy[t + j])));
}
