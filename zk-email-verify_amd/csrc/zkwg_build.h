// Host-side schedule construction: zkwg_config -> ZkSched + segment table.
#pragma once
#include <chrono>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <string>
#include <unordered_map>
#include "../../include/zkwg.h"
#include "zkwg_layout.h"
#include "zkwg_par.h"

// Derived from the final segment list: the per-segment reciprocal of the type's period (so that zk_expand divides with
// one multiply; only set when exact over the segment's range).
static inline void zk_finish_tables(std::vector<ZkSeg>& segs) {
  for (ZkSeg& g : segs) {
    g.pad = 0;
    const u64 d = zk_seg_period(g);
    if (d < 2) continue;
    const u64 rmax = (u64)g.r0 + g.nslots;
    if ((d & (d - 1)) == 0) { g.pad = (u32)((1ull << 32) / d); continue; }
    const u64 m = (1ull << 32) / d + 1, e = m * d - (1ull << 32);
    if (rmax * e < (1ull << 32)) g.pad = (u32)m;
  }
}

// one entry per `piece`-slot piece of the witness (zkwg_sched.h ZkPortionEntry)
static inline void zk_build_entries(u64 W, const std::vector<ZkSeg>& segs, std::vector<ZkPortionEntry>& ent, u32 piece) {
  const u64 np = (W + piece - 1) / piece;
  ent.assign(np, ZkPortionEntry{});
  size_t si = 0;
  for (u64 p = 0; p < np; ++p) {
    const u64 slot0 = p * piece, slot1 = std::min<u64>(W, slot0 + piece);
    while (si + 1 < segs.size() && segs[si].slot + segs[si].nslots <= slot0) ++si;
    const ZkSeg& g = segs[si];
    ZkPortionEntry& e = ent[p];
    e.first_seg = (u32)si;
    if (g.slot <= slot0 && slot1 <= g.slot + g.nslots) {
      e.type = g.type; e.src = g.src; e.a = g.a; e.b = g.b; e.c = g.c; e.magic = g.pad;
      e.r_start = (u32)(slot0 - g.slot) + g.r0;
    } else e.type = ZSEG_NTYPES;
  }
}
static bool build_sched(const zkwg_config& cfg, ZkSched& s, std::vector<ZkSeg>& segs, const zkc::Net* net = nullptr) {
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
  s.in_off[ZKWG_IN_RANGE_FLAGS] = off; off += 4;   // generic input path: per-field "did not fit" bits
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
  w.net = net;
  if (net && (cfg.main_kind != ZKWG_MAIN_EMAIL_VERIFIER || cfg.ignore_body_hash_check || net->n_in != cfg.max_header)) return false;
  u64 max_small = 256;  // largest |d| whose inverse zk_expand looks up
  if (net) max_small = std::max<u64>(max_small, (u64)net->inv_need + 1);
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
      // SelectRegexReveal's IsEqual(i, startIndex): every startIndex its range checks admit
      // (startIndex + 43 < 2^sel_bits) must find |startIndex - i| in the table, also for a
      // maxHeadersLength that is not a power of two
      if (s.body) max_small = std::max<u64>(max_small, 1ull << s.sel_bits);
      max_small = std::max<u64>(max_small, std::max(s.fr[0].nblocks, s.fr[1].nblocks) + 2);
      break;
    case ZKWG_MAIN_RSA_VERIFIER:
      if (cfg.n != 121 || cfg.k != 17) return false;
      s.nframes = 0;
      zk_walk_main_rsa(w, s);
      max_small = std::max<u64>(max_small, 2100);
      break;
    case ZKWG_MAIN_FP_MUL:
      // generic parameters, as long as the numbers fit machine words (zkwg_fpmul_core.h); (121, 17) lives in the RSA path
      if (cfg.n < 1 || cfg.k < 2 || cfg.k > 17 || (u64)cfg.n * cfg.k > 62 || cfg.max_header || cfg.max_body) return false;
      s.nframes = 0;
      zk_walk_main_fpmul(w, s, cfg.n, cfg.k);
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
  zk_finish_tables(segs);
  return true;
}


// ------------------------------------------------------------------ `.sym`-driven layout
// A circom `.sym` file lists every signal of the compiled circuit, one per line:
//     labelIdx,witnessIdx,componentIdx,qualified.name        (witnessIdx = -1: eliminated by the optimiser)
// (circom_tester `loadSymbols`, packages/circuits/tests/email-verifier.test.ts:204-206 via assertOut).
// zk_sym_layout() turns such a file into the witness order zk_expand produces: every kept-v1 slot whose
// name the file maps to a witness index is emitted at that index, kept-v1 slots the compiler eliminated
// are dropped, and the file must not keep a signal the schedule cannot produce.  `alias` holds optional
// rename rules "ours=theirs" (one per line, applied as substring replacements to OUR names first) for the
// compiler-generated names of anonymous components.
struct ZkSymLayout {
  std::vector<u32> dst;            // kept-v1 slot -> witness index (0xffffffff = dropped)
  std::vector<std::string> names;  // witness index -> name (as in the file)
  u64 W = 0;
  std::string err;
  bool allow_holes = false;        // true: signals the schedule does not produce become holes (linear completion)
  std::vector<u8> hole;            // witness index -> 1 if not produced by the schedule
  u64 n_holes = 0;
};
// circom writes multi-dimensional signals as name[i][j]; this schedule names every signal array by its flattened
// index.  Pre-pass over the file: the extent of every dimension of every signal array (all elements are listed,
// eliminated ones with index -1), so that the second pass can flatten.  Only the LAST path component (the signal)
// is flattened; component arrays keep their indices.
struct ZkSymDims {
  std::unordered_map<std::string, std::vector<u32>> ext;   // "path.signal" -> extents (only for >= 2 dims)
  static bool split(const char* p, const char* e, const char*& base_end, u32* idx, int& nd) {
    // name = path '.' signal ( '[' n ']' )*   -- parse the trailing index groups of the last component
    nd = 0;
    const char* q = e;
    u32 tmp[8];
    while (q > p && q[-1] == ']') {
      const char* r = q - 1;
      while (r > p && r[-1] != '[') --r;
      if (r <= p) break;
      if (nd >= 8) return false;
      tmp[nd++] = (u32)strtoul(r, nullptr, 10);
      q = r - 1;
    }
    // the groups must belong to the last component: no '.' after q
    base_end = q;
    for (int i = 0; i < nd; ++i) idx[i] = tmp[nd - 1 - i];
    return true;
  }
};
static inline void zk_collect_names(ZkSched tmp, std::vector<std::string>& names, const zkc::Net* net = nullptr) {
  ZkWalker w;
  w.net = net;
  w.names = true;
  names.assign(tmp.W, std::string());
  w.sink = [&](u64 slot, const std::string& name) { if (slot < names.size()) names[slot] = name; };
  switch (tmp.main_kind) {
    case ZKWG_MAIN_SHA256_BYTES: zk_walk_main_sha(w, tmp); break;
    case ZKWG_MAIN_RSA_VERIFIER: zk_walk_main_rsa(w, tmp); break;
    default: zk_walk_main_ev(w, tmp); break;
  }
}
static bool zk_sym_layout(const ZkSched& s, const char* text, u64 len, const char* alias, u64 alias_len, ZkSymLayout& L,
                          const zkc::Net* net = nullptr) {
  const bool dbg_t = getenv("ZKWG_DEBUG_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!dbg_t) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[zkwg]   .sym layout: %-34s %6.2f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  std::vector<std::string> ours;
  zk_collect_names(s, ours, net);
  lap("our names");
  // rename rules
  std::vector<std::pair<std::string, std::string>> rules;
  for (u64 i = 0; alias && i < alias_len;) {
    u64 j = i;
    while (j < alias_len && alias[j] != '\n') ++j;
    std::string line(alias + i, alias + j);
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    size_t eq = line.find('=');
    if (eq != std::string::npos && eq > 0) rules.emplace_back(line.substr(0, eq), line.substr(eq + 1));
    i = j + 1;
  }
  const unsigned TN = ours.size() > (1u << 16) ? zk_host_threads() : 1u;
  if (!rules.empty())
    zk_parallel_chunks(TN, [&](unsigned ci, unsigned nc) {
      for (size_t k = ours.size() * ci / nc; k < ours.size() * (ci + 1) / nc; ++k) {
        std::string& nm = ours[k];
        for (auto& r : rules)
          for (size_t pos = nm.find(r.first); pos != std::string::npos; pos = nm.find(r.first, pos + r.second.size()))
            nm.replace(pos, r.first.size(), r.second);
      }
    });
  // name -> kept-v1 slot: an open-addressing table of slot indices (0 = empty; the names themselves stay in `ours`), filled by all
  // threads with compare-and-swap -- 1.8 M std::string keys in an unordered_map took 1.8 s on one core.  Of two slots with the
  // same name the lower one wins, as with emplace in slot order.
  struct NameTable {
    std::vector<u32> cell;
    u64 mask = 0;
    const std::vector<std::string>* names = nullptr;
    static u64 hash(const char* p, size_t n) {
      u64 h = 0xcbf29ce484222325ull;
      size_t i = 0;
      for (; i + 8 <= n; i += 8) { u64 w; memcpy(&w, p + i, 8); h = (h ^ w) * 0x100000001b3ull; h ^= h >> 29; }
      for (; i < n; ++i) h = (h ^ (u8)p[i]) * 0x100000001b3ull;
      h ^= h >> 32; h *= 0x9e3779b97f4a7c15ull; h ^= h >> 29;
      return h;
    }
    void init(const std::vector<std::string>& nm) {
      names = &nm;
      u64 cap = 1024;
      while (cap < nm.size() * 2 + 2) cap <<= 1;
      cell.assign(cap, 0u);
      mask = cap - 1;
    }
    void insert(u32 slot) {
      const std::string& k = (*names)[slot];
      for (u64 at = hash(k.data(), k.size()) & mask;; at = (at + 1) & mask) {
        u32 cur = __atomic_load_n(&cell[at], __ATOMIC_RELAXED);
        if (cur == 0) {
          if (__atomic_compare_exchange_n(&cell[at], &cur, slot, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return;
        }
        if ((*names)[cur] == k) {       // the same name twice: keep the lower slot
          while (cur > slot && !__atomic_compare_exchange_n(&cell[at], &cur, slot, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
          return;
        }
      }
    }
    u32 find(const char* p, size_t n) const {       // -> slot, or 0xffffffff
      for (u64 at = hash(p, n) & mask;; at = (at + 1) & mask) {
        const u32 cur = cell[at];
        if (cur == 0) return 0xffffffffu;
        const std::string& k = (*names)[cur];
        if (k.size() == n && memcmp(k.data(), p, n) == 0) return cur;
      }
    }
  } slot_of;
  slot_of.init(ours);
  zk_parallel_chunks(TN, [&](unsigned ci, unsigned nc) {
    for (size_t k = std::max<size_t>(1, ours.size() * ci / nc); k < ours.size() * (ci + 1) / nc; ++k)
      if (!ours[k].empty()) slot_of.insert((u32)k);
  });
  lap("rename + name -> slot map");
  L.dst.assign(ours.size(), 0xffffffffu);
  L.dst[0] = 0;
  L.names.clear();
  L.names.push_back("one");
  u64 unmatched = 0, dup = 0, maxw = 0;
  std::string first_unmatched;
  // The file has one line per signal of the compiled circuit (9.3 M lines, 0.86 GB for EmailVerifier(1024,1536) at --O0):
  // both passes run on threads over chunks cut at line boundaries; the per-chunk results are merged in file order.
  const unsigned T = len > (8u << 20) ? zk_host_threads() : 1u;
  std::vector<u64> cut(T + 1, len);
  cut[0] = 0;
  for (unsigned i = 1; i < T; ++i) {
    u64 at = len * i / T;
    while (at < len && text[at] != '\n') ++at;
    cut[i] = at < len ? at + 1 : len;
  }
  for (unsigned i = 1; i <= T; ++i) cut[i] = std::max(cut[i], cut[i - 1]);
  // pass 1: extents of multi-dimensional signal arrays
  ZkSymDims D;
  {
    std::vector<std::unordered_map<std::string, std::vector<u32>>> part(T);
    zk_parallel_chunks(T, [&](unsigned ci, unsigned) {
      auto& ext = part[ci];
      for (u64 i = cut[ci]; i < cut[ci + 1];) {
        u64 j = i;
        while (j < cut[ci + 1] && text[j] != '\n') ++j;
        const char* p = text + i;
        const char* e = text + j;
        while (e > p && (e[-1] == '\r' || e[-1] == ' ')) --e;
        i = j + 1;
        if (e - p < 4 || e[-1] != ']') continue;                 // (only names that end in an index can be multi-dimensional)
        const char* c1 = (const char*)memchr(p, ',', e - p);
        const char* c2 = c1 ? (const char*)memchr(c1 + 1, ',', e - c1 - 1) : nullptr;
        const char* c3 = c2 ? (const char*)memchr(c2 + 1, ',', e - c2 - 1) : nullptr;
        if (!c3) continue;
        const char* be; u32 idx[8]; int nd;
        if (!ZkSymDims::split(c3 + 1, e, be, idx, nd) || nd < 2) continue;
        auto& ex = ext[std::string(c3 + 1, be)];
        if (ex.size() < (size_t)nd) ex.resize(nd, 0);
        for (int k = 0; k < nd; ++k) ex[k] = std::max(ex[k], idx[k] + 1);
      }
    });
    for (auto& m : part)
      for (auto& kv : m) {
        auto& ex = D.ext[kv.first];
        if (ex.size() < kv.second.size()) ex.resize(kv.second.size(), 0);
        for (size_t k = 0; k < kv.second.size(); ++k) ex[k] = std::max(ex[k], kv.second[k]);
      }
  }
  lap("pass 1 (array extents)");
  // pass 2: every listed signal -> (witness index, kept-v1 slot or none, name)
  struct Hit { u32 widx, slot; std::string name; };
  std::vector<std::vector<Hit>> hits(T);
  std::vector<std::string> errs(T);
  zk_parallel_chunks(T, [&](unsigned ci, unsigned) {
    auto& out = hits[ci];
    out.reserve((cut[ci + 1] - cut[ci]) / 64);
    for (u64 i = cut[ci]; i < cut[ci + 1];) {
      u64 j = i;
      while (j < cut[ci + 1] && text[j] != '\n') ++j;
      // labelIdx,witnessIdx,componentIdx,name
      const char* p = text + i;
      const char* e = text + j;
      while (e > p && (e[-1] == '\r' || e[-1] == ' ')) --e;
      const u64 line_at = i;
      i = j + 1;
      if (p == e) continue;
      const char* c1 = (const char*)memchr(p, ',', e - p);
      const char* c2 = c1 ? (const char*)memchr(c1 + 1, ',', e - c1 - 1) : nullptr;
      const char* c3 = c2 ? (const char*)memchr(c2 + 1, ',', e - c2 - 1) : nullptr;
      if (!c3) { errs[ci] = "malformed .sym line at byte " + std::to_string(line_at); return; }
      const long long widx = strtoll(c1 + 1, nullptr, 10);
      if (widx <= 0) continue;   // eliminated signal (-1) / the constant-one wire (never listed by circom; tolerated)
      if ((u64)widx >= (1ull << 31)) { errs[ci] = "witness index out of range"; return; }
      std::string name(c3 + 1, e);
      bool flattened = false;
      if (e[-1] == ']') {
        // flatten name[i][j].. of a multi-dimensional signal to name[flat]
        const char* be; u32 idx[8]; int nd;
        if (ZkSymDims::split(c3 + 1, e, be, idx, nd) && nd >= 2) {
          auto dit = D.ext.find(std::string(c3 + 1, be));
          if (dit != D.ext.end() && dit->second.size() == (size_t)nd) {
            u64 flat = 0;
            for (int k = 0; k < nd; ++k) flat = flat * dit->second[k] + idx[k];
            name = std::string(c3 + 1, be) + "[" + std::to_string(flat) + "]";
            flattened = true;
          }
        }
      }
      const u32 hit = name.empty() ? 0xffffffffu : slot_of.find(name.data(), name.size());
      if (hit == 0xffffffffu) {
        if (flattened) name.assign(c3 + 1, e);      // a signal the schedule does not produce keeps the file's spelling
        out.push_back(Hit{(u32)widx, 0xffffffffu, std::move(name)});
      } else out.push_back(Hit{(u32)widx, hit, std::move(name)});
    }
  });
  for (auto& er : errs) if (!er.empty()) { L.err = er; return false; }
  lap("pass 2 (lookup)");
  // merge, in file order: a light serial pass decides what every listed signal is (only indices and slots are touched), then the
  // 9 M names move to their witness index on all threads -- every index is written by exactly one hit by then
  {
    std::vector<u64> mx(T, 0);
    zk_parallel_chunks(T, [&](unsigned ci, unsigned) { u64 m = 0; for (const Hit& h : hits[ci]) m = std::max<u64>(m, h.widx); mx[ci] = m; });
    for (u64 m : mx) maxw = std::max(maxw, m);
  }
  L.W = maxw + 1;
  L.names.resize(L.W);
  L.hole.assign(L.W, 0);
  const u32 SKIP = 0xfffffffeu;              // (marks a hit whose name is not stored)
  u64 n_holes = 0, hole_twice = 0xffffffffffffffffull;
  for (auto& part : hits)
    for (Hit& h : part) {
      if (h.slot == 0xffffffffu) {
        if (L.allow_holes) {
          if (L.hole[h.widx]) { if (hole_twice == 0xffffffffffffffffull) hole_twice = h.widx; h.slot = SKIP; continue; }
          L.hole[h.widx] = 1; ++n_holes;
          continue;
        }
        if (!unmatched++) first_unmatched = h.name;
        h.slot = SKIP;
        continue;
      }
      // a layout can only re-order / drop this schedule's own signals: indices beyond its length are bogus
      if (!L.allow_holes && (u64)h.widx >= ours.size()) { L.err = "witness index " + std::to_string(h.widx) + " exceeds the schedule's witness length"; return false; }
      if (L.dst[h.slot] != 0xffffffffu) { if (L.dst[h.slot] != h.widx) ++dup; h.slot = SKIP; continue; }   // (listed twice: the first stands)
      L.dst[h.slot] = h.widx;
    }
  lap("merge (indices)");
  if (unmatched) {
    L.err = std::to_string(unmatched) + " signal(s) kept by the .sym file are not produced by this schedule (first: " + first_unmatched + ")";
    return false;
  }
  if (dup) { L.err = std::to_string(dup) + " name(s) listed with two different witness indices"; return false; }
  if (hole_twice != 0xffffffffffffffffull) { L.err = "witness index " + std::to_string(hole_twice) + " assigned to two signals"; return false; }
  // the kept indices (and the holes) must tile [0, W) exactly once -- checked BEFORE the names move, so that no two hits share an index
  {
    std::vector<u8> seen(L.hole);
    for (u32 d : L.dst) {
      if (d == 0xffffffffu) continue;
      if (d >= L.W || seen[d]) { L.err = "witness index " + std::to_string(d) + " assigned to two signals"; return false; }
      seen[d] = 1;
    }
    for (u64 i = 0; i < L.W; ++i)
      if (!seen[i]) { L.err = "witness index " + std::to_string(i) + " is not covered by the .sym file"; return false; }
  }
  zk_parallel_chunks(T, [&](unsigned ci, unsigned) {
    for (Hit& h : hits[ci]) if (h.slot != SKIP) L.names[h.widx] = std::move(h.name);
    std::vector<Hit>().swap(hits[ci]);
  });
  L.names[0] = "one";
  L.n_holes = n_holes;
  lap("merge (names) + tiling check");
  return true;
}
// Re-target the segment table: split every kept-v1 segment into maximal runs whose destinations are
// consecutive, carry the run's offset inside the logical array in ZkSeg::r0, sort by destination.
static bool zk_remap_segments(ZkSched& s, std::vector<ZkSeg>& segs, const ZkSymLayout& L) {
  std::vector<ZkSeg> out;
  for (const ZkSeg& g : segs) {
    u32 i = 0;
    while (i < g.nslots) {
      const u32 d = L.dst[g.slot + i];
      if (d == 0xffffffffu) { ++i; continue; }
      u32 j = i + 1;
      while (j < g.nslots && L.dst[g.slot + j] == d + (j - i)) ++j;
      out.push_back(ZkSeg{d, j - i, g.type, g.src, g.a, g.b, g.c, g.r0 + i, 0});
      i = j;
    }
  }
  if (L.n_holes) {
    for (u64 i = 0; i < L.W;) {
      if (!L.hole[i]) { ++i; continue; }
      u64 j = i + 1;
      while (j < L.W && L.hole[j] && j - i < 0x7fffffffull) ++j;
      out.push_back(ZkSeg{i, (u32)(j - i), ZSEG_HOLE, 0, 0, 0, 0, 0, 0});
      i = j;
    }
  }
  std::sort(out.begin(), out.end(), [](const ZkSeg& a, const ZkSeg& b) { return a.slot < b.slot; });
  u64 cur = 0;
  for (const ZkSeg& g : out) { if (g.slot != cur) return false; cur += g.nslots; }
  if (cur != L.W || out.size() >= 0xffffffffull) return false;
  segs.swap(out);
  s.W = L.W;
  s.nsegs = (u32)segs.size();
  zk_finish_tables(segs);
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
