// Host-side evaluation of a loaded regex template (zkwg_circom.h zkc::Net) with the SAME inline code the kernels run -- zk_net_scan's
// table walk, zk_net_eval's records, zk_expand's run decoders (zkwg_net_core.h) -- and the loader's self-check built on it.
//
// Why the product carries this: the scan tables of chain_pass replace two recurrences of the template by tables the loader derives;
// a wrong table writes a wrong witness without any error.  tests/test_regex_template.py compares tables and plain gate list for
// the templates the repository knows; a template a user supplies has had no such test, so zkwg_circuit_create_regex evaluates a few
// messages both ways once per handle (1.9 s + 0.3 s for the stand-in template at 1,024 bytes; ZKWG_NET_SELFCHECK=0 skips it).
// Nothing here is a fallback: the values are compared and dropped.
#pragma once
#include "zkwg_circom.h"
#include "zkwg_net_core.h"

namespace zkc {

// kept-signal words [n_kept], `out` (match) and reveal[n_in] of one message; 1 = every assertion of the template holds, 0 = one
// fails, -1 = the runs do not tile the region (internal error)
static inline int eval_host(const Net& N, const u8* msg, u32* words, u32* match, u32* reveal) {
  std::vector<int> lds(N.lds_words, 0x55555555);
  for (u32 i = 0; i < N.n_in; ++i) lds[N.n_pins + i] = msg[i];
  lds[N.n_pins + N.n_in] = 0;
  // zk_net_scan: the chain states entering every position (chain_pass)
  ZkNetChains K;
  K.n_in = N.n_in;
  K.f_end = N.chain.end; K.f_smax = N.chain.smax; K.f_mw = N.chain.mask_words;
  K.b_end = N.bchain.end; K.b_smax = N.bchain.smax; K.b_mw = N.bchain.mask_words; K.b_fdim = N.bchain.fdim;
  K.f_cls = N.chain.cls.data(); K.f_delta = N.chain.delta.data(); K.f_mask = N.chain.mask.data();
  K.b_cls = N.bchain.cls.data(); K.b_delta = N.bchain.delta.data(); K.b_mask = N.bchain.mask.data();
  std::vector<u32> fwords(N.n_in / 4 + 2, 0xa5a5a5a5u), bwords(N.n_in / 4 + 2, 0xa5a5a5a5u);   // (as in the image: stale bytes where no chain wrote)
  zk_net_scan_email(K, msg, fwords.data(), bwords.data());
  const u8* fstate = (const u8*)fwords.data(); const u8* bstate = (const u8*)bwords.data();
  const u32 MS = N.mask_words + K.f_mw + K.b_mw;
  for (u32 i = 0; i < N.n_in; ++i)          // the evaluator's prologue: per byte its mask words (byte-local frontier bits, then the chains')
    zk_net_mask_words(K, N.mask_words, N.mask_tab.data(), i, msg, fstate, bstate, &lds[N.lds_masks + i * MS]);
  std::vector<u32> img(N.n_kept + N.n_temp, 0xdeadbeefu);
  bool ok = true;
  size_t g = 0;
  std::vector<int> snap;
  for (u32 st = 0; st < N.n_steps; ++st) {
    snap = lds;   // the lanes of a step read before any of them writes
    const bool general = N.step_count[st] & 0x8000u;
    for (u32 lane = 0; lane < (N.step_count[st] & 0x7fu); ++lane, ++g) {
      if (general) ok &= zk_net_record(&N.records[g * 16], snap.data(), lds.data(), img.data(), match, reveal, (long long)std::max<u32>(N.inv_need + 1, 256));
      else zk_net_record32(&N.records[g * 16], snap.data(), lds.data(), img.data());
    }
  }
  // byte-local and chain kept signals are not gates of the list: zk_expand decodes them from the position words the evaluator's
  // prologue leaves (zkwg_expand_dec.h ZkDecNetP / ZkDecNetQ) -- the same code, run over the region's runs
  std::vector<u32> small(N.n_kept + N.n_in, 0);
  memcpy(small.data(), img.data(), (size_t)N.n_kept * 4);
  for (u32 i = 0; i < N.n_in; ++i) small[N.n_kept + i] = zk_net_pos_word(K, i, msg, fstate, bstate);
  ZkNetDec D;
  D.pd = N.pd.data(); D.tab = N.tabs.data();
  D.offF = N.offF; D.offB = N.offB; D.nL = N.nL; D.nF = N.nF; D.nB = N.nB; D.b_fdim = N.bchain.fdim;
  D.m_net = 0; D.m_net_pw = N.n_kept; D.n_in = N.n_in;
  u32 covered = 0;
  for (const Net::Run& R : N.runs) {
    if (R.start != covered) return -1;      // the runs tile the region
    for (u32 r = 0; r < R.nslots; ++r) {
      const u32 i = r / R.period, q = r % R.period;
      if (R.dense != 0xffffffffu) { words[R.start + r] = zk_netq_word(D, R.dense, R.period, R.pos0, i, q, small.data()); continue; }
      words[R.start + r] = zk_netp_word(D, N.pd[2 * (R.pd0 + q)], N.pd[2 * (R.pd0 + q) + 1], R.pos0, i, R.start + r, small.data());
    }
    covered += R.nslots;
  }
  if (covered != N.n_kept) return -1;
  return ok ? 1 : 0;
}

// `net` (loaded with scan tables) against the same template with every gate in the list, on `n_msgs` messages: zeros, bytes drawn
// from the template's own comparison constants + CR LF, uniform bytes.  false + err on the first difference.
static inline bool self_check(const std::string& path, const std::string& include_dirs, const std::string& tname, const std::vector<i64>& args,
                              const Net& net, std::string& err, u32 n_msgs = 3) {
  if (!net.chain.end && !net.bchain.end) return true;      // nothing was replaced by tables
  Net plain;
  if (!load(path, include_dirs, tname, args, plain, err, /*chain_limit=*/0)) { err = "self-check: " + err; return false; }
  if (plain.n_kept != net.n_kept || plain.n_in != net.n_in) { err = "self-check: the plain gate list keeps other signals than the tables"; return false; }
  std::vector<u8> alphabet = {13, 10, 32, 59, 61, 58};
  for (u32 v = 97; v < 123; ++v) alphabet.push_back((u8)v);
  for (u32 v = 48; v < 58; ++v) alphabet.push_back((u8)v);
  u64 seed = 0x9e3779b97f4a7c15ull ^ net.n_in;
  auto next = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
  std::vector<u8> msg(net.n_in);
  std::vector<u32> wa(net.n_kept), wb(net.n_kept), ra(net.n_in), rb(net.n_in);
  for (u32 m = 0; m < n_msgs; ++m) {
    for (u32 i = 0; i < net.n_in; ++i) msg[i] = m == 0 ? 0 : (m % 2 ? alphabet[next() % alphabet.size()] : (u8)next());
    u32 ma = 0xffffffffu, mb = 0xffffffffu;
    std::fill(ra.begin(), ra.end(), 0u); std::fill(rb.begin(), rb.end(), 0u);
    const int sa = eval_host(net, msg.data(), wa.data(), &ma, ra.data());
    const int sb = eval_host(plain, msg.data(), wb.data(), &mb, rb.data());
    if (sa < 0 || sb < 0) { err = "self-check: the periodic runs do not tile the region"; return false; }
    if (sa != sb || ma != mb || ra != rb) { err = "self-check: scan tables and gate list disagree on the template's outputs (message " + std::to_string(m) + ")"; return false; }
    for (u32 i = 0; i < net.n_kept; ++i)
      if (wa[i] != wb[i]) { err = "self-check: scan tables and gate list disagree at kept signal " + std::to_string(i) + " (message " + std::to_string(m) + ")"; return false; }
  }
  return true;
}

}  // namespace zkc
