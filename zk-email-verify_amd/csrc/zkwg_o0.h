// One-pass witnesses of a fully numbered circuit (`circom --O0` / `--O1`, the build the reference documents:
// docs/zk-email-docs/UsageGuide/README.md:56-64).
//
// Such a build numbers every alias, constant and linear combination (9.26 M wires for EmailVerifier(1024,1536)
// against the 1.78 M signals that carry information).  zkwg_full.h derives, from the circuit's own `.sym` + `.r1cs`,
// for every wire either the kept-v1 slot it copies or a linear row over kept-v1 slots.  This file lowers that once
// per circuit into what the device needs to write the file's witness directly from the compact image, with no
// staging buffer and no second pass:
//   * an 8-byte DESCRIPTOR per wire: how the wire's value derives from image / record -- the segment arithmetic of
//     its kept-v1 slot resolved at build time to (array, index, shift) for the bit- and byte-valued 95 %, a
//     (segment, index) pair for the rest;
//   * the rows that are real sums (negations, constant multiples, Bits2Num outputs, running sums), split by a static
//     interval analysis into SMALL rows (sources and coefficients small integers, |result| < 2^30: evaluated in
//     64-bit integers, 4 bytes of image each) and FIELD rows (anything else: evaluated mod r, 32 bytes each); their
//     results extend the email's `small` / `fr` image arrays and are referenced by descriptors like any other value.
// zk_o0_rows_small / zk_o0_rows_fr fill those image extensions for the emails about to be expanded; zk_expand3_o0
// (zkwg_kernels_expand3.hip) is zk_expand3 with each lane's code coming from its wire's descriptor.
#pragma once
#include "zkwg_sched.h"
#include "zkwg_r1cs.h"

#define ZK_ROW_LONG 16u   // terms from which a row of its own gets a wavefront instead of a thread
enum ZkDescKind : u32 {
  ZK_D_IMM = 0,       // b = the code itself (a constant, or a reference known at build time)
  ZK_D_BIT64 = 1,     // bit (a & 63) of bits[b]
  ZK_D_BIT8 = 2,      // bit (a & 7) of rec[b]
  ZK_D_BYTE = 3,      // rec[b]
  ZK_D_SMALLRAW = 4,  // small[b], raw 32-bit value
  ZK_D_SMALLS = 5,    // (small[b], small[b+1]) as a signed 64-bit integer v (results of small rows): 0 <= v < 2^31 -> v, else the field element v mod r
  ZK_D_GENERIC = 6,   // slot b of kept-v1 segment (a & 0xffffff): decoded by the segment's own arithmetic
  ZK_D_SMALLN = 9,    // small[b] as a signed integer of 31 bits (results of small rows whose range is that narrow): bit 31 clear -> the value, set -> r - |v|
  ZK_D_BITRUN = 10,   // row terms only: bits (a & 63) .. (a & 63) + ((a >> 6) & 31) of bits[b] as an integer (a run of a Bits2Num sum)
  ZK_D_CODEW = 8,     // small[b] holds the wire's code itself (left there by zk_o0_generic for the wires of kind GENERIC)
  ZK_D_AFF = 11,      // b = entry of the affine table {source descriptor (2 words), c0 | c1 << 16 (two int16), 0}: the wire is c0 + c1 * source -- a row with ONE
                      // source (the `b - 1` of every bit constraint's B side, `1 - x`, `c - byte` ...), evaluated in line by the streaming kernel
  ZK_D_DFA = 7        // BodyHashRegex DFA array element: a = kind << 28 | ZkDfaKind << 24 | q << 20 | position << 9 | param b, b = param c
};

#if !defined(ZKWG_O0_DEVICE_ONLY)
#include <algorithm>
#include <string>
#include <vector>
#include <unordered_map>
#include "zkwg_full.h"

struct ZkO0Tables {
  std::vector<u32> desc;       // 2 words per wire
  u32 small_base = 0, fr_base = 0;   // first row result inside the (extended) small / fr image arrays
  // Rows are evaluated in GROUPS by one thread each: a group is one row, or a CHAIN of consecutive rows in which every
  // row's terms include all the terms of the row before it (the running sums of MultiOR / CalculateTotal / IsZero
  // popcount chains, which the elimination of zkwg_full.h flattened to O(n^2) terms): a chained row only lists the
  // terms it adds and continues from its predecessor's result.
  // small rows: result j -> small[small_base + 2 j .. +2) as a signed 64-bit integer
  std::vector<u64> s_ptr; std::vector<u32> s_term; std::vector<int32_t> s_coef; std::vector<u8> s_chain; std::vector<u32> s_group;
  std::vector<u32> s_out;     // per small row: word of `small` holding its result | bit 31: two words (64-bit integer)
  u64 n_narrow = 0;
  // field rows: result j -> fr[fr_base + j]
  std::vector<u64> f_ptr; std::vector<u32> f_term; std::vector<Fr> f_coef, f_coefm; std::vector<u8> f_kind; std::vector<u8> f_chain; std::vector<u32> f_group;
  u64 n_small() const { return s_ptr.empty() ? 0 : s_ptr.size() - 1; }
  u64 n_fr() const { return f_ptr.empty() ? 0 : f_ptr.size() - 1; }
  u64 n_alias = 0, n_const = 0, terms_before_chaining = 0;
  std::vector<u32> aff;        // 4 words per entry (ZK_D_AFF)
  u64 n_aff() const { return aff.size() / 4; }
  // wires of kind GENERIC (decoded by their segment's own arithmetic: selectors, comparators ...): a pre-pass kernel leaves
  // their codes in small[gen_base + g], so that the streaming kernel itself carries no segment decoder
  u32 gen_base = 0;
  std::vector<u32> gen_seg, gen_r;
};

// static value range of slot r of a kept-v1 segment; false = a field element (no small bound)
static inline bool zk_slot_range(const ZkSeg& g, u32 r, long long& lo, long long& hi) {
  lo = 0; hi = 1;
  switch (g.type) {
    case ZSEG_BITS: case ZSEG_SHA_SP: case ZSEG_SHA_T1: case ZSEG_SHA_T2: case ZSEG_IN8BITS: case ZSEG_LTBITS: case ZSEG_B64BITS:
      return true;
    case ZSEG_IN8: hi = 255; return true;
    case ZSEG_IN8MASK: hi = 255 * 255; return true;
    case ZSEG_SMALL: case ZSEG_VSHIFT: hi = 0xffffffffll; return true;
    case ZSEG_ISZ: return !(r & 1u);
    case ZSEG_SEL: { const u32 per = 3u * g.a, q = r % per; if (q < g.a) return true; return !((q - g.a) & 1u); }
    case ZSEG_REGSEL: {
      const u32 per = 6u + g.a + 1u;
      u32 q;
      if (r < per - 2u) q = r < 4u ? r : r + 2u; else q = (r - (per - 2u)) % per;
      return q >= 6u || !(q & 1u);
    }
    case ZSEG_B64: { const u32 q = r % 68u; if (q < 8u) { hi = 255; return true; } if (q < 62u) return true; return !((q - 62u) & 1u); }
    case ZSEG_DFA: return g.a != ZDFA_EQ || !(r & 1u);
    case ZSEG_RSLB: if (g.a == ZRS_EQ) return !(r & 1u); if (g.a == ZRS_PROC) hi = 255; return true;
    default: return false;   // FR, LIMB, NET, HOLE
  }
}

static inline void zk_o0_slot_desc(const std::vector<ZkSeg>& segs, u64 slot, u32 out[2], u32* seg_index = nullptr, u32* seg_r = nullptr) {
  // the segment holding kept-v1 slot `slot`
  size_t lo = 0, hi = segs.size();
  while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (segs[mid].slot <= slot) lo = mid; else hi = mid; }
  const ZkSeg& g = segs[lo];
  const u32 r = (u32)(slot - g.slot) + g.r0;
  if (seg_index) *seg_index = (u32)lo;
  if (seg_r) *seg_r = r;
  if (slot == 0) { out[0] = ZK_D_IMM << 28; out[1] = 1; return; }   // wire 0 of every circom witness: the constant 1 (the constant term of a combination)
  auto bit64 = [&](u32 word, u32 sh) { out[0] = (ZK_D_BIT64 << 28) | sh; out[1] = word; };
  switch (g.type) {
    case ZSEG_BITS: { const u32 grp = r / g.a, bit = r % g.a; bit64(g.src + grp * g.b + (bit >> 6), bit & 63u); return; }
    case ZSEG_SHA_SP: { const u32 i = r / ZK_SP_SLOTS, q = r % ZK_SP_SLOTS, sub = std::min(q >> 5, 4u); bit64(g.src + i * 5 + sub, q - sub * 32); return; }
    case ZSEG_SHA_T1: { const u32 i = r / ZK_T1_SLOTS, q = r % ZK_T1_SLOTS, sub = std::min(q >> 5, 3u); bit64(g.src + i * 4 + sub, q - sub * 32); return; }
    case ZSEG_SHA_T2: { const u32 i = r / ZK_T2_SLOTS, q = r % ZK_T2_SLOTS, sub = std::min(q >> 5, 4u); bit64(g.src + i * 5 + sub, q - sub * 32); return; }
    case ZSEG_IN8BITS: out[0] = (ZK_D_BIT8 << 28) | (r & 7u); out[1] = g.src + (r >> 3); return;
    case ZSEG_IN8: out[0] = ZK_D_BYTE << 28; out[1] = g.src + r; return;
    case ZSEG_SMALL: out[0] = ZK_D_SMALLRAW << 28; out[1] = g.src + r; return;
    case ZSEG_FR: out[0] = ZK_D_IMM << 28; out[1] = 0x80000000u | (0u << 28) | (g.src + r); return;          // ZK_REF_FRV
    case ZSEG_LIMB: out[0] = ZK_D_IMM << 28; out[1] = 0x80000000u | (2u << 28) | (g.src + 16u * r); return;  // ZK_REF_LIMB
    case ZSEG_DFA: {
      // the bulk of the non-bit wires of EmailVerifier: pre-resolved so that zk_expand3_o0 decodes them in line
      const u32 kind = g.a;
      u32 i, q;
      if (kind == ZDFA_LT) { i = r / 9u; q = r % 9u; } else if (kind == ZDFA_RNG || kind == ZDFA_AND) { i = r; q = 0; } else { i = r >> 1; q = r & 1u; }
      if (i < 2048u && g.b < 512u && kind < 16u) { out[0] = (ZK_D_DFA << 28) | (kind << 24) | (q << 20) | (i << 9) | g.b; out[1] = g.c; return; }
      out[0] = (ZK_D_GENERIC << 28) | (u32)lo; out[1] = r; return;
    }
    default: out[0] = (ZK_D_GENERIC << 28) | (u32)lo; out[1] = r; return;
  }
}
// desc_slot[w]: kept-v1 slot wire w copies, or 0xfffffffe when w is the destination of a row of `P` that is not a plain
// alias; term_slot[t]: kept-v1 slot of term t's source.  Extends s.img_small / s.img_fr by the row results.
static inline bool zk_o0_build(ZkSched& s, const std::vector<ZkSeg>& segs, const ZkLinPlan& P, const std::vector<u32>& desc_slot,
                               const std::vector<u32>& term_slot, ZkO0Tables& T, std::string& err) {
  const u64 W = desc_slot.size();
  if (segs.size() >= (1u << 24)) { err = "too many segments for the O0 descriptor table"; return false; }
  T.desc.assign(2 * W, 0);
  T.fr_base = s.img_fr;
  T.s_ptr.assign(1, 0); T.f_ptr.assign(1, 0);
  // slots whose value comes from their segment's own arithmetic (kind GENERIC), as wires or as terms of rows: numbered
  // first -- zk_o0_generic leaves their codes in small[gen_base + g], every later reader sees kind CODEW
  T.gen_base = s.img_small;
  std::unordered_map<u32, u32> gen_of;
  auto note_generic = [&](u32 slot) {
    u32 d[2], si, sr;
    zk_o0_slot_desc(segs, slot, d, &si, &sr);
    if ((d[0] >> 28) == ZK_D_GENERIC && gen_of.emplace(slot, (u32)T.gen_seg.size()).second) { T.gen_seg.push_back(si); T.gen_r.push_back(sr); }
  };
  for (u64 w = 0; w < W; ++w) if (desc_slot[w] != 0xfffffffeu) note_generic(desc_slot[w]);
  for (u64 r = 0; r < P.n_rows(); ++r)
    if (desc_slot[P.dst[r]] == 0xfffffffeu) for (u64 t = P.row_ptr[r]; t < P.row_ptr[r + 1]; ++t) note_generic(term_slot[t]);
  T.small_base = (T.gen_base + (u32)T.gen_seg.size() + 1u) & ~1u;
  auto describe = [&](u32 slot, u32 d[2], u32* si, u32* sr) {
    zk_o0_slot_desc(segs, slot, d, si, sr);
    if ((d[0] >> 28) == ZK_D_GENERIC) { d[0] = (u32)ZK_D_CODEW << 28; d[1] = T.gen_base + gen_of.at(slot); }
  };
  for (u64 w = 0; w < W; ++w)
    if (desc_slot[w] != 0xfffffffeu) describe(desc_slot[w], &T.desc[2 * w], nullptr, nullptr);
  const Fr p = fr_p();
  // the previous row of each class in full (chain detection)
  std::vector<u32> prev_s_t, prev_f_t; std::vector<int32_t> prev_s_c; std::vector<Fr> prev_f_c;
  std::vector<u32> td; std::vector<int32_t> cs;
  std::vector<u32> s_dst;   // wire written by each small row
  const size_t CHAIN_MIN = 4;   // shorter shared prefixes are cheaper to recompute than to serialise
  for (u64 r = 0; r < P.n_rows(); ++r) {
    const u64 a = P.row_ptr[r], b = P.row_ptr[r + 1];
    const u32 dst = P.dst[r];
    if (desc_slot[dst] != 0xfffffffeu) { ++T.n_alias; continue; }   // plain alias: already described through its source
    if (a == b) { T.desc[2 * dst] = ZK_D_IMM << 28; T.desc[2 * dst + 1] = 0; ++T.n_const; continue; }
    // interval analysis: small iff every source is small-ranged, every coefficient a small signed integer and the sum stays below 2^62
    bool small = true;
    __int128 lo = 0, hi = 0;
    td.clear(); cs.clear();
    for (u64 t = a; t < b; ++t) {
      u32 d[2], si, sr;
      describe(term_slot[t], d, &si, &sr);
      td.push_back(d[0]); td.push_back(d[1]);
      long long l, h;
      if (term_slot[t] == 0) l = h = 1;   // the constant term
      else if (!small || !zk_slot_range(segs[si], sr, l, h)) { small = false; continue; }
      if (!small) continue;
      const Fr& c = P.coef[t];
      long long k;
      if (!(c.l[1] | c.l[2] | c.l[3]) && c.l[0] < (1ull << 30)) k = (long long)c.l[0];
      else {
        u64 bw; const Fr n = fr_sub_raw(p, c, bw);
        if (!(n.l[1] | n.l[2] | n.l[3]) && n.l[0] < (1ull << 30)) k = -(long long)n.l[0]; else { small = false; continue; }
      }
      const __int128 x = (__int128)k * l, y = (__int128)k * h;
      lo += x < y ? x : y; hi += x < y ? y : x;
      if (lo < -((__int128)1 << 62) || hi > ((__int128)1 << 62)) small = false;
      cs.push_back((int32_t)k);
    }
    if (small && b - a <= 2 && !getenv("ZKWG_O0_NO_AFF")) {
      // one source (+ a constant): no row -- the streaming kernel computes c0 + c1 * source from the source's own descriptor
      int src = -1;
      long long c0 = 0;
      bool ok = true;
      for (u64 t = a; t < b; ++t) {
        if (term_slot[t] == 0) c0 += cs[t - a];
        else if (src < 0) src = (int)(t - a);
        else ok = false;
      }
      if (ok && src >= 0) {
        u32 d[2], si, sr;
        describe(term_slot[a + src], d, &si, &sr);
        long long l, h;
        const long long c1 = cs[src];
        if (zk_slot_range(segs[si], sr, l, h) && l >= 0 && h < (1ll << 30) && c0 > -32768 && c0 < 32768 && c1 > -32768 && c1 < 32768 &&
            lo > -((__int128)1 << 27) && hi < ((__int128)1 << 27) && T.aff.size() / 4 < (1u << 28)) {
          T.desc[2 * dst] = (u32)ZK_D_AFF << 28; T.desc[2 * dst + 1] = (u32)(T.aff.size() / 4);
          T.aff.push_back(d[0]); T.aff.push_back(d[1]); T.aff.push_back((u32)(uint16_t)(int16_t)c0 | ((u32)(uint16_t)(int16_t)c1 << 16)); T.aff.push_back(0u);
          continue;
        }
      }
    }
    T.terms_before_chaining += b - a;
    std::vector<Fr> fc;   // field rows: coefficients in standard form, like zk_linear_row; kinds
    std::vector<u8> fk;
    if (!small) { fc.assign(P.coef.begin() + a, P.coef.begin() + b); fk.assign(P.kind.begin() + a, P.kind.begin() + b); }
    {
      // runs: consecutive terms that read consecutive bits of one word of `bits` with doubling coefficients -- the
      // Sum 2^k b_k of Bits2Num / Num2Bits, most of the terms of a constraint system -- become one term: the bit field as an
      // integer (up to 31 bits) times the first coefficient
      const size_t n0 = b - a;
      size_t o = 0;
      for (size_t i = 0; i < n0;) {
        size_t j = i;
        if ((td[2 * i] >> 28) == ZK_D_BIT64)
          while (j + 1 < n0 && j + 1 - i < 31 && td[2 * (j + 1)] == td[2 * j] + 1u && td[2 * (j + 1) + 1] == td[2 * i + 1] &&
                 (small ? (cs[j + 1] == 2 * cs[j]) : fr_eq(fr_add(fc[j], fc[j]), fc[j + 1]))) ++j;
        td[2 * o] = j > i ? (((u32)ZK_D_BITRUN << 28) | ((u32)(j - i) << 6) | (td[2 * i] & 63u)) : td[2 * i];
        td[2 * o + 1] = td[2 * i + 1];
        if (small) cs[o] = cs[i]; else { fc[o] = fc[i]; fk[o] = fk[i]; }
        ++o; i = j + 1;
      }
      td.resize(2 * o);
      if (small) cs.resize(o); else { fc.resize(o); fk.resize(o); }
    }
    const size_t nt = td.size() / 2;
    // chained: every term of the previous row of the class (same source, same coefficient) is also a term of this one, in
    // the same order -- the row then only lists the others and continues from its predecessor's result.  `rest` = the
    // indices of this row's own terms.
    std::vector<u32> rest;
    auto subseq = [&](const std::vector<u32>& pt, size_t np, auto same_coef) {
      rest.clear();
      if (np < CHAIN_MIN || nt <= np) return false;
      size_t q = 0;
      for (size_t i = 0; i < nt; ++i) {
        if (q < np && pt[2 * q] == td[2 * i] && pt[2 * q + 1] == td[2 * i + 1] && same_coef(q, i)) ++q;
        else { if (rest.size() >= nt - np) return false; rest.push_back((u32)i); }
      }
      return q == np;
    };
    if (small && getenv("ZKWG_DEBUG_FORMS")) {
      // histogram of the small rows' shapes: "<terms>: kind*coef ..." (debug aid for choosing in-line descriptor kinds)
      static std::unordered_map<std::string, u64> forms;
      static u64 seen = 0;
      std::string key = std::to_string(nt) + ":";
      if (nt <= 3) for (size_t i = 0; i < nt; ++i) key += " k" + std::to_string(td[2 * i] >> 28) + ((td[2 * i] >> 28) == ZK_D_IMM ? "(" + std::to_string(td[2 * i + 1]) + ")" : "") + "*" + std::to_string(cs[i]);
      ++forms[key];
      if (++seen % 100000 == 0 || r + 1 == P.n_rows()) {
        std::vector<std::pair<u64, std::string>> top;
        for (auto& kv : forms) top.emplace_back(kv.second, kv.first);
        std::sort(top.rbegin(), top.rend());
        fprintf(stderr, "[zkwg] small-row forms after %llu rows:\n", (unsigned long long)seen);
        for (size_t i = 0; i < top.size() && i < 24; ++i) fprintf(stderr, "[zkwg]   %8llu  %s\n", (unsigned long long)top[i].first, top[i].second.c_str());
      }
    }
    if (small) {
      const bool ch = subseq(prev_s_t, prev_s_c.size(), [&](size_t q, size_t i) { return prev_s_c[q] == cs[i]; });
      if (!ch) { rest.resize(nt); for (size_t i = 0; i < nt; ++i) rest[i] = (u32)i; }
      for (u32 i : rest) { T.s_term.push_back(td[2 * i]); T.s_term.push_back(td[2 * i + 1]); T.s_coef.push_back(cs[i]); }
      T.s_chain.push_back(ch ? 1 : 0);
      if (!ch) T.s_group.push_back((u32)T.n_small());
      T.s_ptr.push_back(T.s_coef.size());
      // one word when the row's range allows it (most rows: sums of a few bits) -- half the traffic of the row results;
      // the offsets are known once all rows are counted (patched below)
      const bool narrow = lo >= -((__int128)1 << 30) && hi < ((__int128)1 << 30);
      T.s_out.push_back(narrow ? 0u : 0x80000000u); s_dst.push_back(dst);
      if (narrow) ++T.n_narrow;
      prev_s_t = td; prev_s_c = cs;
    } else {
      // (field rows are not chained: none of EmailVerifier's continues its predecessor, and zk_o0_rows_fr then needs no
      // running value -- its registers go to the emails it handles at once)
      const bool ch = false;
      if (!ch) { rest.resize(nt); for (size_t i = 0; i < nt; ++i) rest[i] = (u32)i; }
      for (u32 i : rest) { T.f_term.push_back(td[2 * i]); T.f_term.push_back(td[2 * i + 1]); T.f_coef.push_back(fc[i]); T.f_coefm.push_back(fr_to_mont(fc[i])); T.f_kind.push_back(fk[i]); }
      T.f_chain.push_back(ch ? 1 : 0);
      if (!ch) T.f_group.push_back((u32)T.n_fr());
      T.f_ptr.push_back(T.f_kind.size());
      T.desc[2 * dst] = ZK_D_IMM << 28; T.desc[2 * dst + 1] = 0x80000000u | (T.fr_base + (u32)(T.f_ptr.size() - 2));   // ZK_REF_FRV
      prev_f_t = td; prev_f_c.swap(fc);
    }
  }
  T.s_group.push_back((u32)T.n_small());
  T.f_group.push_back((u32)T.n_fr());
  {
    // [narrow results, one word each | wide results, two words each (8-byte aligned)]
    u32 nn = 0, nw = 0;
    const u32 wide_base = T.small_base + (u32)((T.n_narrow + 1u) & ~1ull);
    for (size_t j = 0; j < s_dst.size(); ++j) {
      const bool wide = T.s_out[j] >> 31;
      const u32 off = wide ? wide_base + 2u * nw++ : T.small_base + nn++;
      T.s_out[j] = (wide ? 0x80000000u : 0u) | off;
      T.desc[2 * (u64)s_dst[j]] = (u32)(wide ? ZK_D_SMALLS : ZK_D_SMALLN) << 28; T.desc[2 * (u64)s_dst[j] + 1] = off;
    }
  }
  const u64 small_words = ((T.n_narrow + 1u) & ~1ull) + 2 * (T.n_small() - T.n_narrow);
  if (getenv("ZKWG_DEBUG_PLAN")) {
    // chains: rows per chain, terms of the head row, terms the other rows add
    u64 nch = 0, max_rows = 0, max_head = 0, tot_rows = 0, tot_head = 0, tot_rest = 0, max_rest = 0, hist[6] = {0};
    for (size_t g = 0; g + 1 < T.s_group.size(); ++g) {
      const u32 a = T.s_group[g], b = T.s_group[g + 1];
      if (b - a == 1) continue;
      ++nch; tot_rows += b - a; max_rows = std::max<u64>(max_rows, b - a);
      const u64 head = T.s_ptr[a + 1] - T.s_ptr[a];
      tot_head += head; max_head = std::max(max_head, head);
      for (u32 j = a + 1; j < b; ++j) { const u64 n = T.s_ptr[j + 1] - T.s_ptr[j]; tot_rest += n; max_rest = std::max(max_rest, n); }
      ++hist[b - a <= 4 ? 0 : b - a <= 16 ? 1 : b - a <= 64 ? 2 : b - a <= 256 ? 3 : b - a <= 1024 ? 4 : 5];
    }
    fprintf(stderr, "[zkwg] small chains: %llu (rows %llu, longest %llu; head terms %llu, longest %llu; added terms %llu, most per row %llu); chains of <=4/16/64/256/1024/more rows: %llu %llu %llu %llu %llu %llu; narrow results %llu of %llu\n",
            (unsigned long long)nch, (unsigned long long)tot_rows, (unsigned long long)max_rows, (unsigned long long)tot_head, (unsigned long long)max_head,
            (unsigned long long)tot_rest, (unsigned long long)max_rest, (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2],
            (unsigned long long)hist[3], (unsigned long long)hist[4], (unsigned long long)hist[5], (unsigned long long)T.n_narrow, (unsigned long long)T.n_small());
    u64 fh[6] = {0}, fone = 0, fmone = 0, fgen = 0;
    for (size_t j = 0; j + 1 < T.f_ptr.size(); ++j) { const u64 n = T.f_ptr[j + 1] - T.f_ptr[j]; ++fh[n <= 4 ? 0 : n <= 16 ? 1 : n <= 64 ? 2 : n <= 256 ? 3 : n <= 1024 ? 4 : 5]; }
    for (u8 k : T.f_kind) (k == ZK_COEF_ONE ? fone : k == ZK_COEF_MINUS_ONE ? fmone : fgen)++;
    fprintf(stderr, "[zkwg] field rows of <=4/16/64/256/1024/more terms: %llu %llu %llu %llu %llu %llu; coefficients +1 %llu, -1 %llu, other %llu\n",
            (unsigned long long)fh[0], (unsigned long long)fh[1], (unsigned long long)fh[2], (unsigned long long)fh[3], (unsigned long long)fh[4], (unsigned long long)fh[5],
            (unsigned long long)fone, (unsigned long long)fmone, (unsigned long long)fgen);
    u64 kinds[16] = {0}, by_type[ZSEG_NTYPES + 1] = {0};
    for (u64 w = 0; w < W; ++w) { const u32 k = T.desc[2 * w] >> 28; ++kinds[k == ZK_D_CODEW ? 6 : (k == ZK_D_SMALLN ? 5 : k)]; }
    for (u32 sg : T.gen_seg) ++by_type[segs[sg].type];
    fprintf(stderr, "[zkwg] O0 descriptors: imm %llu bit64 %llu bit8 %llu byte %llu smallraw %llu smalls %llu dfa %llu generic %llu; generic by segment type:",
            (unsigned long long)kinds[0], (unsigned long long)kinds[1], (unsigned long long)kinds[2], (unsigned long long)kinds[3], (unsigned long long)kinds[4], (unsigned long long)kinds[5], (unsigned long long)kinds[7], (unsigned long long)kinds[6]);
    for (u32 t = 0; t <= ZSEG_NTYPES; ++t) if (by_type[t]) fprintf(stderr, " %u:%llu", t, (unsigned long long)by_type[t]);
    fprintf(stderr, "\n");
  }
  if ((u64)T.small_base + small_words >= (1u << 28) || (u64)T.fr_base + T.n_fr() >= (1u << 28)) { err = "circuit too large for the O0 row tables"; return false; }
  s.img_small = (u32)((T.small_base + small_words + 3u) & ~3ull);
  s.img_fr = (u32)(T.fr_base + T.n_fr());
  return true;
}
#endif  // host tables

#if defined(__HIPCC__)
// device-side tables of a numbered circuit (kernel argument, by value)
struct ZkO0Dev {
  const uint2* desc;       // per wire
  const uint4* aff;        // ZK_D_AFF entries
  u64 W;                   // wires
  u32 nportions;           // pieces of 256 K wires
  u32 emails_per_wg;       // a workgroup expands its piece for this many emails (the descriptors are loaded once)
  u32 small_base, fr_base;
  const u64* s_ptr; const uint2* s_term; const int* s_coef; const u32* s_out;
  const u32* s_single; u32 n_small_single;       // small rows that are groups of their own (one thread each)
  const u32* s_long; u32 n_small_long;           // ... those of more than ZK_ROW_LONG terms (one wavefront each)
  const uint2* s_chains; u32 n_small_chains;     // (first row, rows) of the chains (one wavefront each: prefix sum over the rows)
  const u32* gen_seg; const u32* gen_r; u32 n_gen, gen_base;   // wires decoded by zk_o0_generic into small[gen_base ..]
  const u64* f_ptr; const uint2* f_term; const Fr* f_coef; const Fr* f_coefm; const u8* f_kind; u32 n_fr_groups;   // (n_fr_groups = field rows: each a group of its own)
};
#endif
