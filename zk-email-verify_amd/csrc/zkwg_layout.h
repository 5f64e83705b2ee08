// Layout walker: the single C++ definition of the compact "kept-v1" witness order.
//
// The walk follows circom's O0 numbering over the component
// tree of the reference circuits and visits only the *kept* signals:
// main I/O, hint-assigned signals (`<--`), and signals assigned by `<==` with a
// quadratic right-hand side.  While walking it
//   * allocates space in the per-email compact image for every value a compute kernel
//     must produce (alloc_bits / alloc_small / alloc_fr),
//   * emits the segment table that zk_expand streams from (seg()), and
//   * optionally emits the name of every slot (`.sym` pass).
//
// Reference sources walked: packages/circuits/email-verifier.circom:42-174,
// lib/{sha,rsa,fp,bigint,base64}.circom, utils/{array,regex,hash}.circom and the
// circomlib templates they instantiate.
#pragma once
#include <string>
#include <vector>
#include <functional>
#include "zkwg_sched.h"

struct ZkWalker {
  u64 cur = 0;        // next free slot (advanced by one/arr/skip)
  u64 seg_cur = 0;    // next slot not yet covered by a segment (advanced by seg)
  bool names = false; // emit names?
  std::function<void(u64 slot, const std::string& name)> sink;
  std::vector<ZkSeg> segs;
  u32 nbits = 0, nsmall = 0, nfr = 0;

  u32 alloc_bits(u32 n) { u32 r = nbits; nbits += n; return r; }
  u32 alloc_small(u32 n) { u32 r = nsmall; nsmall += n; return r; }
  u32 alloc_fr(u32 n) { u32 r = nfr; nfr += n; return r; }

  void seg(u32 type, u64 nslots, u32 src, u32 a = 0, u32 b = 0, u32 c = 0) {
    // a segment holds at most 2^32-1 slots; split longer runs
    while (nslots > 0) {
      u32 n = (u32)std::min<u64>(nslots, 0x40000000ull);
      segs.push_back(ZkSeg{seg_cur, n, type, src, a, b, c});
      seg_cur += n;
      nslots -= n;
      if (nslots) {  // only uniform types may be split
        if (type == ZSEG_SMALL || type == ZSEG_FR || type == ZSEG_IN8) src += n;
      }
    }
  }
  void one(const std::string& nm) {
    if (names) sink(cur, nm);
    ++cur;
  }
  void arr(const std::string& nm, u32 n) {
    if (names)
      for (u32 i = 0; i < n; ++i) sink(cur + i, nm + "[" + std::to_string(i) + "]");
    cur += n;
  }
  void skip(u64 n) { cur += n; }
};

static inline u32 zk_log2ceil(u64 a) {  // utils/functions.circom:7-17
  u64 n = a - 1;
  u32 r = 0;
  while (n > 0) { ++r; n /= 2; }
  return r;
}

static inline std::string zk_idx(const std::string& base, u32 i) { return base + "[" + std::to_string(i) + "]"; }

// circomlib Sha256compression: ZK_COMP_SLOTS kept signals, 952 image words at `src`
static inline void zk_walk_compression(ZkWalker& w, const std::string& p, u32 src) {
  w.seg(ZSEG_SHA_SP, 48 * ZK_SP_SLOTS, src + ZK_G_SP);
  w.seg(ZSEG_SHA_T1, 64 * ZK_T1_SLOTS, src + ZK_G_T1);
  w.seg(ZSEG_SHA_T2, 64 * ZK_T2_SLOTS, src + ZK_G_T2);
  w.seg(ZSEG_BITS, 136 * 33, src + ZK_G_SUMA, 33, 1);
  if (!w.names) { w.skip(ZK_COMP_SLOTS); return; }
  for (u32 i = 0; i < 48; ++i) {
    std::string q = zk_idx(p + ".sigmaPlus", i);
    w.arr(q + ".sigma1.xor3.out", 32); w.arr(q + ".sigma1.xor3.mid", 32);
    w.arr(q + ".sigma0.xor3.out", 32); w.arr(q + ".sigma0.xor3.mid", 32);
    w.arr(q + ".sum.out", 34);
  }
  for (u32 i = 0; i < 64; ++i) {
    std::string q = zk_idx(p + ".t1", i);
    w.arr(q + ".ch.out", 32);
    w.arr(q + ".bigsigma1.xor3.out", 32); w.arr(q + ".bigsigma1.xor3.mid", 32);
    w.arr(q + ".sum.out", 35);
  }
  for (u32 i = 0; i < 64; ++i) {
    std::string q = zk_idx(p + ".t2", i);
    w.arr(q + ".bigsigma0.xor3.out", 32); w.arr(q + ".bigsigma0.xor3.mid", 32);
    w.arr(q + ".maj.out", 32); w.arr(q + ".maj.mid", 32);
    w.arr(q + ".sum.out", 33);
  }
  for (u32 i = 0; i < 64; ++i) w.arr(zk_idx(p + ".suma", i) + ".out", 33);
  for (u32 i = 0; i < 64; ++i) w.arr(zk_idx(p + ".sume", i) + ".out", 33);
  for (u32 i = 0; i < 8; ++i) w.arr(zk_idx(p + ".fsum", i) + ".out", 33);
}

static inline void zk_alloc_sha_frame(ZkWalker& w, ZkShaFrame& f) {
  f.m_ibi = w.alloc_small(1);
  f.m_idx = w.alloc_small(1);
  f.m_digest = w.alloc_small(8);
  f.b_lenbits = w.alloc_bits(1);
  f.b_digest = w.alloc_bits(4);
  f.b_trace = w.alloc_bits(f.nblocks * ZK_TRACE_GROUPS);
}

// Sha256Bytes / Sha256BytesPartial sub-tree (lib/sha.circom:17-38, 47-80, 89-292)
static inline void zk_walk_sha_frame(ZkWalker& w, const std::string& p, const ZkShaFrame& f) {
  const std::string sha = p + ".sha";
  w.seg(ZSEG_SMALL, 1, f.m_ibi);
  w.one(sha + ".inBlockIndex");
  w.seg(ZSEG_BITS, f.lenbits + 1, f.b_lenbits, f.lenbits + 1, 1);
  w.arr(sha + ".bitLengthVerifier.lt.n2b.out", f.lenbits + 1);
  for (u32 i = 0; i < f.nblocks; ++i)
    zk_walk_compression(w, zk_idx(sha + ".sha256compression", i), f.b_trace + i * ZK_TRACE_GROUPS);
  w.seg(ZSEG_SEL, (u64)256 * 3 * f.nblocks, f.m_idx, f.nblocks, f.m_digest);
  if (!w.names) {
    w.skip((u64)256 * 3 * f.nblocks);
  } else {
    for (u32 k = 0; k < 256; ++k) {
      std::string q = zk_idx(sha + ".arraySelectors", k);
      w.arr(q + ".calcTotalValue.nums", f.nblocks);
      for (u32 j = 0; j < f.nblocks; ++j) {
        w.one(zk_idx(q + ".eqs", j) + ".isz.out");
        w.one(zk_idx(q + ".eqs", j) + ".isz.inv");
      }
    }
  }
  w.seg(ZSEG_IN8BITS, (u64)f.max_bytes * 8, f.in_data);
  if (!w.names) w.skip((u64)f.max_bytes * 8);
  else for (u32 i = 0; i < f.max_bytes; ++i) w.arr(zk_idx(p + ".bytes", i) + ".out", 8);
  if (f.partial) {
    w.seg(ZSEG_IN8BITS, 32 * 8, f.in_pre);
    if (!w.names) w.skip(32 * 8);
    else for (u32 i = 0; i < 32; ++i) w.arr(zk_idx(p + ".states", i) + ".out", 8);
  }
}

// main = Sha256Bytes(N), public [paddedIn, paddedInLength]
// (packages/circuits/tests/test-circuits/sha-test.circom:5)
static inline void zk_walk_main_sha(ZkWalker& w, ZkSched& s) {
  s.m_one = w.alloc_small(1);
  s.m_hdr_len = w.alloc_small(1);
  zk_alloc_sha_frame(w, s.fr[0]);
  w.seg(ZSEG_SMALL, 1, s.m_one);
  w.one("one");
  w.seg(ZSEG_BITS, 256, s.fr[0].b_digest, 256, 4);
  w.arr("main.out", 256);
  w.seg(ZSEG_IN8, s.fr[0].max_bytes, s.fr[0].in_data);
  w.arr("main.paddedIn", s.fr[0].max_bytes);
  w.seg(ZSEG_SMALL, 1, s.m_hdr_len);
  w.one("main.paddedInLength");
  zk_walk_sha_frame(w, "main", s.fr[0]);
  s.n_public = 256 + s.fr[0].max_bytes + 1;
}
