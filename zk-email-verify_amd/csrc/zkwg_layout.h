// Layout walker: the single C++ definition of the compact "kept-v1" witness order.
//
// The walk follows circom's O0 numbering over the component
// tree of the reference circuits and visits only the *kept* signals:
// main I/O, hint-assigned signals (`<--`), and signals assigned by `<==` with a
// quadratic right-hand side.  A visitor either just counts (layout pass: records the
// segment offsets the kernels need) or also emits names (`.sym` pass).
//
// Reference sources walked: packages/circuits/email-verifier.circom:42-174,
// lib/{sha,rsa,fp,bigint,base64}.circom, utils/{array,regex,hash}.circom and the
// circomlib templates they instantiate.
#pragma once
#include <string>
#include <functional>
#include "zkwg_sched.h"

struct ZkWalker {
  u64 cur = 0;                       // next free slot
  bool names = false;                // emit names?
  std::function<void(u64 slot, const std::string& name)> sink;

  void one(const std::string& nm) {
    if (names) sink(cur, nm);
    ++cur;
  }
  void arr(const std::string& nm, u32 n) {
    if (names)
      for (u32 i = 0; i < n; ++i) sink(cur + i, nm + "[" + std::to_string(i) + "]");
    cur += n;
  }
};

static inline u32 zk_log2ceil(u64 a) {  // utils/functions.circom:7-17
  u64 n = a - 1;
  u32 r = 0;
  while (n > 0) { ++r; n /= 2; }
  return r;
}

static inline std::string zk_idx(const std::string& base, u32 i) { return base + "[" + std::to_string(i) + "]"; }

// circomlib Sha256compression: ZK_COMP_SLOTS kept signals
static inline void zk_walk_compression(ZkWalker& w, const std::string& p) {
  if (!w.names) { w.cur += ZK_COMP_SLOTS; return; }
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

// Sha256Bytes / Sha256BytesPartial sub-tree (lib/sha.circom:17-38, 47-80, 89-292)
static inline void zk_walk_sha_frame(ZkWalker& w, const std::string& p, ZkShaFrame& f) {
  const std::string sha = p + ".sha";
  f.s_inBlockIndex = w.cur; w.one(sha + ".inBlockIndex");
  f.s_lenbits = w.cur; w.arr(sha + ".bitLengthVerifier.lt.n2b.out", f.lenbits + 1);
  f.s_comp = w.cur;
  for (u32 i = 0; i < f.nblocks; ++i) zk_walk_compression(w, zk_idx(sha + ".sha256compression", i));
  f.s_sel = w.cur;
  if (!w.names) {
    w.cur += (u64)256 * 3 * f.nblocks;
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
  f.s_bytes = w.cur;
  if (!w.names) w.cur += (u64)f.max_bytes * 8;
  else for (u32 i = 0; i < f.max_bytes; ++i) w.arr(zk_idx(p + ".bytes", i) + ".out", 8);
  f.s_states = w.cur;
  if (f.partial) {
    if (!w.names) w.cur += 32 * 8;
    else for (u32 i = 0; i < 32; ++i) w.arr(zk_idx(p + ".states", i) + ".out", 8);
  }
}

// main = Sha256Bytes(N), public [paddedIn, paddedInLength]
static inline void zk_walk_main_sha(ZkWalker& w, ZkSched& s) {
  w.one("one");
  s.s_out = w.cur; w.arr("main.out", 256);
  s.s_pub_in = w.cur; w.arr("main.paddedIn", s.fr[0].max_bytes); w.one("main.paddedInLength");
  s.s_prv_in = w.cur;
  zk_walk_sha_frame(w, "main", s.fr[0]);
  s.n_public = 256 + s.fr[0].max_bytes + 1;
}
