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
#include "zkwg_bh_dfa.h"
#include "zkwg_circom.h"

struct ZkWalker {
  const zkc::Net* net = nullptr;   // BodyHashRegex loaded from a circom template (else the built-in zkwg v1 circuit)
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
      segs.push_back(ZkSeg{seg_cur, n, type, src, a, b, c, 0, 0});
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
  s.fr[0].m_len = s.m_hdr_len;
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

// ------------------------------------------------------------------ RSA (lib/rsa.circom, lib/fp.circom, lib/bigint.circom)
static inline void zk_alloc_blt(ZkWalker& w, ZkBltLayout& L) {
  L.b_lt = w.alloc_bits(34);
  L.f_eq = w.alloc_fr(34);
  L.m_gates = w.alloc_small(48);
}
// BigLessThan(121,17): lt[17] (Num2Bits(122)), eq[17] (isz.out, isz.inv), ors/ands/eq_ands[16]
static inline void zk_walk_blt(ZkWalker& w, const std::string& p, const ZkBltLayout& L) {
  w.seg(ZSEG_BITS, 17 * 122, L.b_lt, 122, 2);
  for (u32 i = 0; i < 17; ++i) w.arr(zk_idx(p + ".lt", i) + ".n2b.out", 122);
  w.seg(ZSEG_FR, 34, L.f_eq);
  for (u32 i = 0; i < 17; ++i) {
    w.one(zk_idx(p + ".eq", i) + ".isz.out");
    w.one(zk_idx(p + ".eq", i) + ".isz.inv");
  }
  w.seg(ZSEG_SMALL, 48, L.m_gates);
  for (u32 i = 0; i < 16; ++i) w.one(zk_idx(p + ".ors", i) + ".out");
  for (u32 i = 0; i < 16; ++i) w.one(zk_idx(p + ".ands", i) + ".out");
  for (u32 i = 0; i < 16; ++i) w.one(zk_idx(p + ".eq_ands", i) + ".out");
}
static inline void zk_alloc_fpmul(ZkWalker& w, ZkFpMulLayout& F) {
  F.f_main = w.alloc_fr(100);
  F.b_qr = w.alloc_bits(68);
  zk_alloc_blt(w, F.blt);
  F.f_carry = w.alloc_fr(33);
  F.b_carry = w.alloc_bits(96);
}
// FpMul(121,17) (lib/fp.circom:16-81)
static inline void zk_walk_fpmul(ZkWalker& w, const std::string& p, const ZkFpMulLayout& F) {
  w.seg(ZSEG_FR, 100, F.f_main);
  w.arr(p + ".v_ab", 33); w.arr(p + ".q", 17); w.arr(p + ".r", 17); w.arr(p + ".v_pq_r", 33);
  w.seg(ZSEG_BITS, 34 * 121, F.b_qr, 121, 2);
  for (u32 i = 0; i < 17; ++i) w.arr(zk_idx(p + ".q_range_check", i) + ".out", 121);
  for (u32 i = 0; i < 17; ++i) w.arr(zk_idx(p + ".r_range_check", i) + ".out", 121);
  zk_walk_blt(w, p + ".r_p_lt_check", F.blt);
  w.seg(ZSEG_FR, 33, F.f_carry);
  w.arr(p + ".tCheck.carry", 33);
  w.seg(ZSEG_BITS, 32 * 131, F.b_carry, 131, 3);
  for (u32 i = 0; i < 32; ++i) w.arr(zk_idx(p + ".tCheck.carryRangeChecks", i) + ".out", 131);
}
static inline void zk_alloc_rsa(ZkWalker& w, ZkRsaLayout& R) {
  R.present = 1;
  R.b_modbits = w.alloc_bits(34);
  R.b_msgbits = w.alloc_bits(34);
  R.m_modzero = w.alloc_small(205);
  R.b_sigbits = w.alloc_bits(34);
  zk_alloc_blt(w, R.blt);
  for (u32 m = 0; m < 17; ++m) zk_alloc_fpmul(w, R.mul[m]);
}
// RSAVerifier65537(121,17) sub-tree (lib/rsa.circom:13-46)
static inline void zk_walk_rsa(ZkWalker& w, const std::string& p, const ZkRsaLayout& R) {
  w.seg(ZSEG_BITS, 17 * 121, R.b_modbits, 121, 2);
  for (u32 i = 0; i < 17; ++i) w.arr(zk_idx(p + ".padder.modulusN2B", i) + ".out", 121);
  w.seg(ZSEG_BITS, 17 * 121, R.b_msgbits, 121, 2);
  for (u32 i = 0; i < 17; ++i) w.arr(zk_idx(p + ".padder.messageN2B", i) + ".out", 121);
  w.seg(ZSEG_ISZ, 2 * 205, R.m_modzero);
  for (u32 i = 0; i < 205; ++i) {
    w.one(zk_idx(p + ".padder.modulusZero", i) + ".out");
    w.one(zk_idx(p + ".padder.modulusZero", i) + ".inv");
  }
  w.seg(ZSEG_BITS, 17 * 121, R.b_sigbits, 121, 2);
  for (u32 i = 0; i < 17; ++i) w.arr(zk_idx(p + ".signatureRangeCheck", i) + ".out", 121);
  zk_walk_blt(w, p + ".bigLessThan", R.blt);
  for (u32 m = 0; m < 16; ++m) zk_walk_fpmul(w, zk_idx(p + ".bigPow.doublers", m), R.mul[m]);
  zk_walk_fpmul(w, p + ".bigPow.adder", R.mul[16]);
}

// main = RSAVerifier65537(121,17), public [modulus]
// (packages/circuits/tests/test-circuits/rsa-test.circom:5)
static inline void zk_walk_main_rsa(ZkWalker& w, ZkSched& s) {
  s.m_one = w.alloc_small(1);
  s.m_hdr_len = w.alloc_small(1);
  s.rsa.msg_from_digest = 0;
  s.rsa.in_mod = s.in_off[3]; s.rsa.in_sig = s.in_off[4]; s.rsa.in_msg = s.in_off[5];
  zk_alloc_rsa(w, s.rsa);
  w.seg(ZSEG_SMALL, 1, s.m_one);
  w.one("one");
  w.seg(ZSEG_LIMB, 17, s.rsa.in_mod);
  w.arr("main.modulus", 17);
  w.seg(ZSEG_LIMB, 17, s.rsa.in_msg);
  w.arr("main.message", 17);
  w.seg(ZSEG_LIMB, 17, s.rsa.in_sig);
  w.arr("main.signature", 17);
  zk_walk_rsa(w, "main", s.rsa);
  s.n_public = 17;
}

// main = FpMul(n, k), no public inputs (packages/circuits/tests/test-circuits/fp-mul-test.circom:5; lib/fp.circom:16-81)
static inline void zk_walk_main_fpmul(ZkWalker& w, ZkSched& s, u32 n, u32 k) {
  ZkFpgLayout& F = s.fpg;
  F.present = 1; F.n = n; F.k = k;
  F.in_a = s.in_off[3]; F.in_b = s.in_off[4]; F.in_p = s.in_off[5];
  u32 lk = 0; for (u32 t = k; t; t >>= 1) ++lk;          // log_ceil(k) of lib/bigint-func.circom:14-23
  const u32 cb = n + lk + 5;                             // CheckCarryToZero: m + EPSILON - n bits per carry
  s.m_one = w.alloc_small(1);
  F.m_out = w.alloc_small(k);
  F.f_main = w.alloc_fr(6 * k - 2);
  F.b_qr = w.alloc_bits(2 * k);
  F.b_lt = w.alloc_bits(k);
  F.f_eq = w.alloc_fr(2 * k);
  F.m_gates = w.alloc_small(3 * (k - 1));
  F.f_carry = w.alloc_fr(2 * k - 1);
  F.b_carry = w.alloc_bits(2 * k - 2);
  w.seg(ZSEG_SMALL, 1 + k, s.m_one);                     // (m_one and m_out are consecutive)
  w.one("one");
  w.arr("main.out", k);
  w.seg(ZSEG_LIMB, k, F.in_a); w.arr("main.a", k);
  w.seg(ZSEG_LIMB, k, F.in_b); w.arr("main.b", k);
  w.seg(ZSEG_LIMB, k, F.in_p); w.arr("main.p", k);
  const std::string p = "main";
  w.seg(ZSEG_FR, 6 * k - 2, F.f_main);
  w.arr(p + ".v_ab", 2 * k - 1); w.arr(p + ".q", k); w.arr(p + ".r", k); w.arr(p + ".v_pq_r", 2 * k - 1);
  w.seg(ZSEG_BITS, 2 * k * n, F.b_qr, n, 1);
  for (u32 i = 0; i < k; ++i) w.arr(zk_idx(p + ".q_range_check", i) + ".out", n);
  for (u32 i = 0; i < k; ++i) w.arr(zk_idx(p + ".r_range_check", i) + ".out", n);
  w.seg(ZSEG_BITS, k * (n + 1), F.b_lt, n + 1, 1);
  for (u32 i = 0; i < k; ++i) w.arr(zk_idx(p + ".r_p_lt_check.lt", i) + ".n2b.out", n + 1);
  w.seg(ZSEG_FR, 2 * k, F.f_eq);
  for (u32 i = 0; i < k; ++i) {
    w.one(zk_idx(p + ".r_p_lt_check.eq", i) + ".isz.out");
    w.one(zk_idx(p + ".r_p_lt_check.eq", i) + ".isz.inv");
  }
  w.seg(ZSEG_SMALL, 3 * (k - 1), F.m_gates);
  for (u32 i = 0; i + 1 < k; ++i) w.one(zk_idx(p + ".r_p_lt_check.ors", i) + ".out");
  for (u32 i = 0; i + 1 < k; ++i) w.one(zk_idx(p + ".r_p_lt_check.ands", i) + ".out");
  for (u32 i = 0; i + 1 < k; ++i) w.one(zk_idx(p + ".r_p_lt_check.eq_ands", i) + ".out");
  w.seg(ZSEG_FR, 2 * k - 1, F.f_carry);
  w.arr(p + ".tCheck.carry", 2 * k - 1);
  w.seg(ZSEG_BITS, (2 * k - 2) * cb, F.b_carry, cb, 1);
  for (u32 i = 0; i + 2 < 2 * k; ++i) w.arr(zk_idx(p + ".tCheck.carryRangeChecks", i) + ".out", cb);
  s.n_public = 0;
}

// ------------------------------------------------------------------ BodyHashRegex DFA circuit (zkwg v1)
// [EXT] zk-regex's generated body_hash_regex.circom is absent; this is zkwg's own circuit of the same
// style over the DFA tables of tools/gen_bh_dfa.py (restated literally in oracle/pyref/zkemail.py
// BodyHashRegexV1).  Component arrays are [k][num_bytes], kind-major.
static inline void zk_alloc_bh_regex(ZkWalker& w, ZkSched& s, u32 N) {
  const u32 nb = N + 1;
  s.m_rev = w.alloc_small(N);
  if (w.net) {   // loaded template: gate values (kept signals, then temporaries) + the match output
    s.net_mode = 1;
    s.net_kept = w.net->n_kept; s.net_total = w.net->n_kept + w.net->n_temp;
    s.net_steps = w.net->n_steps; s.net_pins = w.net->n_pins; s.net_lds_words = w.net->lds_words;
    s.net_lds_masks = w.net->lds_masks; s.net_mask_words = w.net->mask_words; s.net_lanes = w.net->lanes;
    s.m_net = w.alloc_small(s.net_total + 1);   // + a scratch word for the evaluator's idle lanes
    s.m_net_out = w.alloc_small(1);
    s.net_chain_end = w.net->chain.end; s.net_chain_smax = w.net->chain.smax; s.net_chain_mw = w.net->chain.mask_words;
    s.net_bchain_end = w.net->bchain.end; s.net_bchain_smax = w.net->bchain.smax; s.net_bchain_mw = w.net->bchain.mask_words;
    s.net_bchain_fdim = w.net->bchain.fdim;
    s.m_net_st = w.net->chain.end ? w.alloc_small(N / 4 + 2) : 0;
    s.m_net_bst = w.net->bchain.end ? w.alloc_small(N / 4 + 2) : 0;
    s.m_net_pw = w.alloc_small(N + 1);
    s.m_dfa_own = s.m_dfa_st = s.m_dfa_cm = s.m_dfa_pm = s.m_dfa_acc = 0;
    return;
  }
  s.m_dfa_own = w.alloc_small(2 * nb + ZK_DFA_NPUBLIC * N + N);
  s.m_dfa_st = w.alloc_small(nb + 1);
  s.m_dfa_cm = w.alloc_small(nb + 1);
  s.m_dfa_pm = w.alloc_small(nb + 1);
  s.m_dfa_acc = w.alloc_small(1);
}
static inline void zk_walk_bh_regex(ZkWalker& w, const std::string& p, const ZkSched& s, u32 N) {
  const u32 nb = N + 1;
  if (w.net) {   // the kept signals of the loaded template, in the compiler's numbering order (zkwg_circom.h layout_walk)
    for (const zkc::Net::Run& R : w.net->runs) {
      if (R.dense != 0xffffffffu) w.seg(ZSEG_NETQ, R.nslots, R.dense, R.period, R.pos0, R.start);
      else w.seg(ZSEG_NETP, R.nslots, R.pd0, R.period, R.pos0, R.start);
      if (!w.names) w.skip(R.nslots);
      else for (u32 i = 0; i < R.nslots; ++i) w.one(p + w.net->names[R.start + i]);
    }
    return;
  }
  auto arr2 = [&](const std::string& base, u32 k, u32 n, const char* a, const char* b) {
    if (!w.names) { w.skip(2 * (u64)n); return; }
    for (u32 i = 0; i < n; ++i) {
      std::string q = base + "[" + std::to_string(k) + "][" + std::to_string(i) + "]";
      w.one(q + a); w.one(q + b);
    }
  };
  // own signals: outputs (reveal0), then the quadratic intermediates
  w.seg(ZSEG_SMALL, N, s.m_rev);
  w.arr(p + ".reveal0", N);
  w.seg(ZSEG_SMALL, 2 * nb + ZK_DFA_NPUBLIC * N + N, s.m_dfa_own);
  w.arr(p + ".live_c1", nb); w.arr(p + ".live_t", nb);
  w.arr(p + ".prev_states0", ZK_DFA_NPUBLIC * N); w.arr(p + ".is_reveal0", N);
  // eq[n][i]
  u32 n_eq = 0, n_rg = 0;
  for (u32 k = 0; k < ZK_DFA_NPRIM; ++k) if (ZK_DFA_PRIM[k][0] == 0) {
    w.seg(ZSEG_DFA, 2 * (u64)nb, s.m_dfa_st, ZDFA_EQ, ZK_DFA_PRIM[k][1]);
    arr2(p + ".eq", n_eq++, nb, ".isz.out", ".isz.inv");
  }
  // lt[2n][i] = LessThan(8)(lo-1, in), lt[2n+1][i] = LessThan(8)(in, hi+1)
  for (u32 k = 0; k < ZK_DFA_NPRIM; ++k) if (ZK_DFA_PRIM[k][0] == 1) {
    const u32 lo = ZK_DFA_PRIM[k][1], hi = ZK_DFA_PRIM[k][2];
    for (u32 side = 0; side < 2; ++side) {
      // n2b.in = in[0] + 256 - in[1]:  side 0: (lo-1) + 256 - in ; side 1: in + 256 - (hi+1)
      w.seg(ZSEG_DFA, 9 * (u64)nb, s.m_dfa_st, ZDFA_LT, side == 0 ? lo + 255 : 255 - hi, side);
      if (!w.names) w.skip(9 * (u64)nb);
      else for (u32 i = 0; i < nb; ++i)
        w.arr(p + ".lt[" + std::to_string(2 * n_rg + side) + "][" + std::to_string(i) + "].n2b.out", 9);
    }
    ++n_rg;
  }
  n_rg = 0;
  for (u32 k = 0; k < ZK_DFA_NPRIM; ++k) if (ZK_DFA_PRIM[k][0] == 1) {
    w.seg(ZSEG_DFA, nb, s.m_dfa_st, ZDFA_RNG, ZK_DFA_PRIM[k][1], ZK_DFA_PRIM[k][2]);
    if (!w.names) w.skip(nb);
    else for (u32 i = 0; i < nb; ++i) w.one(p + ".and_rng[" + std::to_string(n_rg) + "][" + std::to_string(i) + "].out");
    ++n_rg;
  }
  u32 n_cls = 0;
  for (u32 k = 0; k < ZK_DFA_NCLASS; ++k) if (ZK_DFA_CLASS[k][1] > 1) {
    w.seg(ZSEG_DFA, 2 * (u64)nb, s.m_dfa_st, ZDFA_CLS, k, ZK_DFA_CLASS_MEMBERS[k]);
    arr2(p + ".cls_or", n_cls++, nb, ".is_zero.out", ".is_zero.inv");
  }
  for (u32 t = 0; t < ZK_DFA_NTRANS; ++t) {
    w.seg(ZSEG_DFA, nb, s.m_dfa_st, ZDFA_AND, ZK_DFA_TRANS[t][0], ZK_DFA_TRANS[t][2]);
    if (!w.names) w.skip(nb);
    else for (u32 i = 0; i < nb; ++i) w.one(p + ".and[" + std::to_string(t) + "][" + std::to_string(i) + "].out");
  }
  // states with several incoming non-zero-origin transitions (tmp_or), and with both kinds (st_or)
  u32 n_tmp = 0, n_st = 0;
  for (u32 d = 1; d < ZK_DFA_STATES; ++d) {
    u32 nz = 0;
    for (u32 t = 0; t < ZK_DFA_NTRANS; ++t) if (ZK_DFA_TRANS[t][1] == d && ZK_DFA_TRANS[t][0] != 0) ++nz;
    if (nz > 1) {
      w.seg(ZSEG_DFA, 2 * (u64)nb, s.m_dfa_st, ZDFA_TMP, d);
      arr2(p + ".tmp_or", n_tmp++, nb, ".is_zero.out", ".is_zero.inv");
    }
  }
  w.seg(ZSEG_DFA, 2 * (u64)nb, s.m_dfa_st, ZDFA_FZE);
  if (!w.names) w.skip(2 * (u64)nb);
  else for (u32 i = 0; i < nb; ++i) { w.one(zk_idx(p + ".fze", i) + ".is_zero.out"); w.one(zk_idx(p + ".fze", i) + ".is_zero.inv"); }
  for (u32 d = 1; d < ZK_DFA_STATES; ++d) {
    u32 nz = 0, z = 0;
    for (u32 t = 0; t < ZK_DFA_NTRANS; ++t) if (ZK_DFA_TRANS[t][1] == d) { if (ZK_DFA_TRANS[t][0] != 0) ++nz; else ++z; }
    if (nz && z) {
      w.seg(ZSEG_DFA, 2 * (u64)nb, s.m_dfa_st, ZDFA_ST, d);
      arr2(p + ".st_or", n_st++, nb, ".is_zero.out", ".is_zero.inv");
    }
  }
  w.seg(ZSEG_ISZ, 2, s.m_dfa_acc);
  w.one(p + ".is_accepted.is_zero.out"); w.one(p + ".is_accepted.is_zero.inv");
  w.seg(ZSEG_DFA, 2 * (u64)N, s.m_dfa_st, ZDFA_SUB);
  if (!w.names) w.skip(2 * (u64)N);
  else for (u32 i = 0; i < N; ++i) { w.one(zk_idx(p + ".substr_or", i) + ".is_zero.out"); w.one(zk_idx(p + ".substr_or", i) + ".is_zero.inv"); }
}

// ------------------------------------------------------------------ RemoveSoftLineBreaks
// helpers/remove-soft-line-breaks.circom:14-126 as `main.qpEncodingChecker` (email-verifier.circom:148-156);
// PoseidonModular / Poseidon from utils/hash.circom:49-82 and circomlib poseidon.circom [EXT].
static inline void zk_alloc_rslb(ZkWalker& w, ZkSched& s, u32 M) {
  s.rs_nch = 2 * M / 16;
  s.f_rs_chunk = w.alloc_fr(s.rs_nch);
  s.f_rs_sum_enc = w.alloc_fr(M);
  s.f_rs_rdec = w.alloc_fr(M - 1);
  s.f_rs_sum_dec = w.alloc_fr(M);
  s.f_rs_mux = w.alloc_fr(2 * M - 1);
  s.f_rs_hash = w.alloc_fr(s.rs_nch * ZK_P16_KEPT + (s.rs_nch - 1) * ZK_P2_KEPT);
  s.f_rs_final = w.alloc_fr(2);
}
static inline void zk_walk_poseidon_names(ZkWalker& w, const std::string& p, u32 t, u32 rp) {
  if (!w.names) { w.skip(3 * (8 * t + rp)); return; }
  for (u32 r = 0; r < 8; ++r) for (u32 j = 0; j < t; ++j) {
    std::string q = p + ".pEx.sigmaF[" + std::to_string(r) + "][" + std::to_string(j) + "]";
    w.one(q + ".out"); w.one(q + ".in2"); w.one(q + ".in4");
  }
  for (u32 r = 0; r < rp; ++r) {
    std::string q = zk_idx(p + ".pEx.sigmaP", r);
    w.one(q + ".out"); w.one(q + ".in2"); w.one(q + ".in4");
  }
}
static inline void zk_walk_rslb(ZkWalker& w, const std::string& p, const ZkSched& s, u32 M) {
  const u32 enc = s.fr[1].in_data;
  // intermediates with a quadratic definition, in declaration order (:20-31)
  w.seg(ZSEG_RSLB, M, enc, ZRS_PROC, M); w.arr(p + ".processed", M);
  w.seg(ZSEG_RSLB, M - 2, enc, ZRS_TSB); w.arr(p + ".tempSoftBreak", M - 2);
  w.seg(ZSEG_RSLB, M - 2, enc, ZRS_SB); w.arr(p + ".isSoftBreak", M - 2);
  w.seg(ZSEG_FR, M, s.f_rs_sum_enc); w.arr(p + ".sumEnc", M);
  w.seg(ZSEG_FR, M - 1, s.f_rs_rdec);
  if (!w.names) w.skip(M - 1); else for (u32 i = 1; i < M; ++i) w.one(zk_idx(p + ".rDec", i));
  w.seg(ZSEG_FR, M, s.f_rs_sum_dec); w.arr(p + ".sumDec", M);
  // muxEnc[i] = Mux1 -> MultiMux1(1): c[0] is fed a product for i >= 1 (:99-101); mux.out[0] is quadratic
  w.seg(ZSEG_FR, 2 * M - 1, s.f_rs_mux);
  if (!w.names) w.skip(2 * M - 1);
  else for (u32 i = 0; i < M; ++i) {
    if (i) w.one(zk_idx(p + ".muxEnc", i) + ".c[0]");
    w.one(zk_idx(p + ".muxEnc", i) + ".mux.out[0]");
  }
  // rHasher = PoseidonModular(2M): Poseidon(16) per chunk, Poseidon(2) merges
  const u32 nh = s.rs_nch * ZK_P16_KEPT + (s.rs_nch - 1) * ZK_P2_KEPT;
  w.seg(ZSEG_FR, nh, s.f_rs_hash);
  for (u32 c = 0; c < s.rs_nch; ++c) {
    zk_walk_poseidon_names(w, zk_idx(p + ".rHasher.anon_Poseidon_chunk", c), 17, 68);
    if (c) zk_walk_poseidon_names(w, zk_idx(p + ".rHasher.anon_Poseidon_merge", c), 3, 57);
  }
  // IsEqual()([encoded[i], 61]) / ([encoded[i+1], 13]) / ([encoded[i+2], 10]) (:47-62)
  const char* nm[3] = {".anon_IsEqual_eq", ".anon_IsEqual_cr", ".anon_IsEqual_lf"};
  const u32 ch[3] = {61, 13, 10};
  for (u32 k = 0; k < 3; ++k) {
    w.seg(ZSEG_RSLB, 2 * (M - k), enc, ZRS_EQ, k, ch[k]);
    if (!w.names) w.skip(2 * (M - k));
    else for (u32 i = 0; i < M - k; ++i) { w.one(zk_idx(p + nm[k], i) + ".isz.out"); w.one(zk_idx(p + nm[k], i) + ".isz.inv"); }
  }
  w.seg(ZSEG_FR, 2, s.f_rs_final);
  w.one(p + ".anon_IsEqual_final.isz.out"); w.one(p + ".anon_IsEqual_final.isz.inv");
}

// ------------------------------------------------------------------ EmailVerifier main
// (packages/circuits/email-verifier.circom:42-174, flags (ignoreBodyHashCheck, 0, 0, 0),
//  `component main { public [ pubkey ] }`, tests/test-circuits/email-verifier-test.circom:5)
static inline void zk_frame_len_alloc(ZkWalker& w, ZkShaFrame& f) {
  f.m_len = w.alloc_small(1);
  f.m_len_m1 = w.alloc_small(1);
  f.b_len = w.alloc_bits(1);
  f.azp = 1;
}
static inline void zk_walk_azp(ZkWalker& w, const std::string& p, const ZkShaFrame& f) {
  const u32 bl = zk_log2ceil(f.max_bytes);
  w.seg(ZSEG_LTBITS, (u64)f.max_bytes * (bl + 1), f.m_len_m1, bl);
  if (!w.names) w.skip((u64)f.max_bytes * (bl + 1));
  else for (u32 i = 0; i < f.max_bytes; ++i) w.arr(zk_idx(p + ".lessThans", i) + ".n2b.out", bl + 1);
}
static inline void zk_walk_main_ev(ZkWalker& w, ZkSched& s) {
  const u32 N = s.fr[0].max_bytes;
  const u32 M = s.body ? s.fr[1].max_bytes : 0;
  // image allocations
  s.m_one = w.alloc_small(1);
  s.f_out = w.alloc_fr(3);
  zk_frame_len_alloc(w, s.fr[0]);
  s.m_hdr_len = s.fr[0].m_len;
  zk_alloc_sha_frame(w, s.fr[0]);
  s.rsa.msg_from_digest = 1;
  s.rsa.in_mod = s.in_off[3]; s.rsa.in_sig = s.in_off[4]; s.rsa.in_msg = 0;
  s.rsa.m_digest = s.fr[0].m_digest;
  zk_alloc_rsa(w, s.rsa);
  if (s.body) {
    zk_frame_len_alloc(w, s.fr[1]);
    zk_alloc_sha_frame(w, s.fr[1]);
    s.m_bh_idx = w.alloc_small(1);
    zk_alloc_bh_regex(w, s, N);
    s.m_chars = w.alloc_small(44);
    s.b_shift = w.alloc_bits(1);
    s.sel_bits = zk_log2ceil((u64)N + 44 - 1);
    if (s.rslb) zk_alloc_rslb(w, s, M);
  }
  s.f_pos = w.alloc_fr(420);

  // main I/O: [1], outputs, public inputs, private inputs
  w.seg(ZSEG_SMALL, 1, s.m_one); w.one("one");
  w.seg(ZSEG_FR, 3, s.f_out);
  w.one("main.pubkeyHash"); w.one("main.shaHi"); w.one("main.shaLo");
  if (s.mask_header) { w.seg(ZSEG_IN8MASK, N, s.fr[0].in_data, s.in_off[9]); w.arr("main.maskedHeader", N); }
  if (s.mask_body) { w.seg(ZSEG_IN8MASK, M, s.fr[1].in_data, s.in_off[10]); w.arr("main.maskedBody", M); }
  w.seg(ZSEG_LIMB, 17, s.rsa.in_mod); w.arr("main.pubkey", 17);
  w.seg(ZSEG_IN8, N, s.fr[0].in_data); w.arr("main.emailHeader", N);
  w.seg(ZSEG_SMALL, 1, s.fr[0].m_len); w.one("main.emailHeaderLength");
  w.seg(ZSEG_LIMB, 17, s.rsa.in_sig); w.arr("main.signature", 17);
  if (s.mask_header) { w.seg(ZSEG_IN8, N, s.in_off[9]); w.arr("main.headerMask", N); }
  if (s.body) {
    w.seg(ZSEG_SMALL, 1, s.m_bh_idx); w.one("main.bodyHashIndex");
    w.seg(ZSEG_IN8, 32, s.fr[1].in_pre); w.arr("main.precomputedSHA", 32);
    w.seg(ZSEG_IN8, M, s.fr[1].in_data); w.arr("main.emailBody", M);
    w.seg(ZSEG_SMALL, 1, s.fr[1].m_len); w.one("main.emailBodyLength");
    if (s.rslb) { w.seg(ZSEG_IN8, M, s.in_off[11]); w.arr("main.decodedEmailBodyIn", M); }
    if (s.mask_body) { w.seg(ZSEG_IN8, M, s.in_off[10]); w.arr("main.bodyMask", M); }
  }
  // sub-components in creation order
  const u32 blh = zk_log2ceil(N);
  w.seg(ZSEG_BITS, blh, s.fr[0].b_len, blh, 1);
  w.arr("main.n2bHeaderLength.out", blh);
  zk_walk_azp(w, "main.anon_AssertZeroPadding_header", s.fr[0]);
  zk_walk_sha_frame(w, "main.anon_Sha256Bytes", s.fr[0]);
  zk_walk_rsa(w, "main.rsaVerifier", s.rsa);
  if (s.mask_header) {  // ByteMask(maxHeadersLength) (utils/bytes.circom:173-185): out[i] <== in[i] * mask[i]
    w.seg(ZSEG_IN8MASK, N, s.fr[0].in_data, s.in_off[9]);
    w.arr("main.byteMask_header.out", N);
  }
  if (s.body) {
    const u32 blb = zk_log2ceil(M);
    w.seg(ZSEG_BITS, blb, s.fr[1].b_len, blb, 1);
    w.arr("main.n2bBodyLength.out", blb);
    zk_walk_azp(w, "main.anon_AssertZeroPadding_body", s.fr[1]);
    zk_walk_bh_regex(w, "main.anon_BodyHashRegex", s, N);
    // SelectRegexReveal(N, 44) (utils/regex.circom:17-52)
    const std::string sr = "main.anon_SelectRegexReveal";
    const u32 bl = s.sel_bits, per = 6 + bl + 1;
    w.seg(ZSEG_REGSEL, (u64)(per - 2) + (u64)(N - 1) * per, s.m_bh_idx, bl, s.m_rev, N);
    if (!w.names) w.skip((u64)(per - 2) + (u64)(N - 1) * per);
    else for (u32 i = 0; i < N; ++i) {
      w.one(zk_idx(sr + ".anon_IsEqual", i) + ".isz.out"); w.one(zk_idx(sr + ".anon_IsEqual", i) + ".isz.inv");
      w.one(zk_idx(sr + ".anon_IsZero", i) + ".out"); w.one(zk_idx(sr + ".anon_IsZero", i) + ".inv");
      if (i) { w.one(zk_idx(sr + ".anon_IsPrevZero", i) + ".out"); w.one(zk_idx(sr + ".anon_IsPrevZero", i) + ".inv"); }
      w.arr(zk_idx(sr + ".anon_GreaterThan", i) + ".lt.n2b.out", bl + 1);
    }
    w.seg(ZSEG_VSHIFT, (u64)blh * N, s.m_bh_idx, N, s.m_rev);
    w.arr(sr + ".anon_VarShiftLeft.tmp", blh * N);
    w.seg(ZSEG_BITS, blh, s.b_shift, blh, 1);
    w.arr(sr + ".anon_VarShiftLeft.n2b.out", blh);
    // Base64Decode(32) (lib/base64.circom:14-64)
    const std::string b64 = "main.anon_Base64Decode";
    w.seg(ZSEG_B64BITS, 44 * 6, s.m_chars);
    for (u32 g = 0; g < 11; ++g) for (u32 j = 0; j < 4; ++j)
      w.arr(b64 + ".bitsIn[" + std::to_string(g) + "][" + std::to_string(j) + "].out", 6);
    w.seg(ZSEG_B64, 44 * 68, s.m_chars);
    for (u32 g = 0; g < 11; ++g) for (u32 j = 0; j < 4; ++j) {
      std::string t = b64 + ".translate[" + std::to_string(g) + "][" + std::to_string(j) + "]";
      const char* mids[8] = {"range_AZ", "sum_AZ", "range_az", "sum_az", "range_09", "sum_09", "sum_plus", "sum_slash"};
      for (auto m : mids) w.one(t + "." + m);
      const char* cmps[6] = {"le_Z", "ge_A.lt", "le_z", "ge_a.lt", "le_9", "ge_0.lt"};
      for (auto m : cmps) w.arr(t + "." + m + ".n2b.out", 9);
      const char* eqs[3] = {"equal_plus", "equal_slash", "equal_eqsign"};
      for (auto m : eqs) { w.one(t + "." + m + ".out"); w.one(t + "." + m + ".inv"); }
    }
    zk_walk_sha_frame(w, "main.anon_Sha256BytesPartial", s.fr[1]);
    if (s.rslb) zk_walk_rslb(w, "main.qpEncodingChecker", s, M);
    if (s.mask_body) {
      w.seg(ZSEG_IN8MASK, M, s.fr[1].in_data, s.in_off[10]);
      w.arr("main.byteMask_body.out", M);
    }
  }
  // PoseidonLarge(121,17) -> Poseidon(9): 8x10 + 60 S-boxes x (out, in2, in4)
  w.seg(ZSEG_FR, 420, s.f_pos);
  const std::string pp = "main.anon_PoseidonLarge.anon_Poseidon.pEx";
  for (u32 r = 0; r < 8; ++r) for (u32 j = 0; j < 10; ++j) {
    std::string t = pp + ".sigmaF[" + std::to_string(r) + "][" + std::to_string(j) + "]";
    w.one(t + ".out"); w.one(t + ".in2"); w.one(t + ".in4");
  }
  for (u32 r = 0; r < 60; ++r) {
    std::string t = zk_idx(pp + ".sigmaP", r);
    w.one(t + ".out"); w.one(t + ".in2"); w.one(t + ".in4");
  }
  s.n_public = 3 + (s.mask_header ? N : 0) + (s.mask_body ? M : 0) + 17;
}
