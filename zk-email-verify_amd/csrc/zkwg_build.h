// Host-side schedule construction: zkwg_config -> ZkSched + segment table.
#pragma once
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/zkwg.h"
#include "zkwg_layout.h"

static bool build_sched(const zkwg_config& cfg, ZkSched& s, std::vector<ZkSeg>& segs,
                        std::vector<u32>& first_seg) {
  memset(&s, 0, sizeof(s));
  if (cfg.layout != ZKWG_LAYOUT_KEPT_V1) return false;
  if (cfg.enable_header_masking || cfg.enable_body_masking || cfg.remove_soft_line_breaks) return false;
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
  s.in_stride = (off + 15u) & ~15u;

  auto init_frame = [&](ZkShaFrame& f, u32 max_bytes, u32 partial, u32 in_data, u32 in_len) {
    f.max_bytes = max_bytes;
    f.nblocks = max_bytes / 64;
    f.lenbits = zk_log2ceil((u64)max_bytes * 8);
    f.partial = partial;
    f.in_data = in_data; f.in_len = in_len; f.in_pre = s.in_off[ZKWG_IN_PRECOMPUTED_SHA];
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
  s.nportions = (u32)((s.W + ZK_PORTION - 1) / ZK_PORTION);
  first_seg.assign(s.nportions, 0);
  u32 si = 0;
  for (u32 p = 0; p < s.nportions; ++p) {
    u64 slot0 = (u64)p * ZK_PORTION;
    while (si + 1 < s.nsegs && segs[si].slot + segs[si].nslots <= slot0) ++si;
    first_seg[p] = si;
  }
  return true;
}

