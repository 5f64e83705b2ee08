// Complete witnesses for a compiled circuit (`--O0` / `--O1` artefacts): linear completion plan.
//
// The schedule of this library produces the signals that carry information (inputs, hints, quadratic
// definitions: the kept-v1 layout).  A circuit compiled without full simplification also numbers every
// alias, constant and linear combination (2.4 M of the 3.1 M signals of EmailVerifier(576,192) at O0).
// Their definitions are exactly the LINEAR constraints of the circuit's own `.r1cs`: given the `.sym`
// (which names are ours) and the `.r1cs`, this file derives, once per circuit, for every signal the
// schedule does not produce a row   w[dst] = c0 + sum_k coef_k * w[src_k]   whose sources are signals the
// schedule does produce -- by triangular elimination over the linear constraints (a constraint with one
// unknown wire defines it), then substitution down to produced wires.  zkwg_o0.h lowers the rows to the
// descriptor / row tables that the row kernels and zk_expand3_o0 evaluate on the device.  Reference: the compile line the reference documents is `circom ... --O0`
// (docs/zk-email-docs/UsageGuide/README.md:56-64); `.sym` / `.r1cs` are what `circom_tester` loads
// (packages/circuits/tests/email-verifier.test.ts:21-31,44,204-206).
#pragma once
#include <algorithm>
#include <chrono>
#include <queue>
#include <string>
#include <vector>
#include "zkwg_r1cs.h"

struct ZkLinPlan {
  std::vector<u64> row_ptr;   // rows + 1
  std::vector<u32> dst;       // witness index written by the row
  std::vector<u32> src;       // term sources (witness indices produced by the schedule; 0 = the constant 1)
  std::vector<Fr> coef;       // term coefficients, STANDARD form (most operands are 0 or 1: no product needed)
  std::vector<u8> kind;       // ZK_COEF_ONE / MINUS_ONE / GENERIC
  u64 n_rows() const { return dst.size(); }
};

// produced[w] = 1 for the wires zk_expand writes (wire 0 included).  Fails (err) if a wire cannot be derived.
static inline bool zk_linear_plan(const ZkR1csHost& R, const std::vector<u8>& produced, ZkLinPlan& P, std::string& err) {
  const u64 nw = R.n_wires, m = R.n_constraints;
  const bool dbg_t = getenv("ZKWG_DEBUG_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) {
    if (!dbg_t) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[zkwg]   linear plan: %-24s %7.2f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  if (produced.size() < nw) { err = "the .sym file lists fewer signals than the .r1cs has wires"; return false; }
  auto row = [&](u64 lc, u64& a, u64& b) { a = R.row_ptr[lc]; b = R.row_ptr[lc + 1]; };
  const Fr zero = fr_zero();
  const Fr unit_m = fr_R(), neg_unit_m = fr_neg(fr_R());
  // linear constraints: A or B empty -> C = 0;  A (or B) a pure constant k -> k * B - C = 0.
  // Flat storage throughout (9.3 M constraints, 7.5 M derived wires for EmailVerifier(1024,1536) at --O0: one vector per
  // constraint / per definition cost more time in the allocator than the elimination itself):
  // lin_ptr / lin_w / lin_c = the combinations  sum c_i w_i = 0  (Montgomery coefficients)
  std::vector<u64> lin_ptr(1, 0);
  std::vector<u32> lin_w;
  std::vector<Fr> lin_c;
  lin_ptr.reserve(m / 2 + 1);
  for (int pass = 0; pass < 2; ++pass) {
    u64 terms = 0;
    for (u64 i = 0; i < m; ++i) {
      u64 a0, a1, b0, b1, c0, c1;
      row(3 * i, a0, a1); row(3 * i + 1, b0, b1); row(3 * i + 2, c0, c1);
      const bool a_empty = a0 == a1, b_empty = b0 == b1;
      const bool a_const = a1 - a0 == 1 && R.wire[a0] == 0, b_const = b1 - b0 == 1 && R.wire[b0] == 0;
      if (!(a_empty || b_empty || a_const || b_const)) continue;
      const bool both = !a_empty && !b_empty;
      const bool use_b = a_const;           // A constant: the other side is B
      const u64 o0 = both ? (use_b ? b0 : a0) : 0, o1 = both ? (use_b ? b1 : a1) : 0;
      const u64 n = (c1 - c0) + (o1 - o0);
      if (!n) continue;
      if (pass == 0) { terms += n; continue; }
      for (u64 t = c0; t < c1; ++t) { lin_w.push_back(R.wire[t]); lin_c.push_back(R.kind[t] == ZK_COEF_ONE ? neg_unit_m : (R.kind[t] == ZK_COEF_MINUS_ONE ? unit_m : fr_neg(R.coef[t]))); }
      if (both) {
        // k * other - C = 0
        const Fr k = use_b ? R.coef[a0] : R.coef[b0];
        const bool k_one = fr_eq(k, unit_m);
        for (u64 t = o0; t < o1; ++t) { lin_w.push_back(R.wire[t]); lin_c.push_back(k_one ? R.coef[t] : fr_mont_mul(k, R.coef[t])); }
      }
      lin_ptr.push_back(lin_w.size());
    }
    if (pass == 0) { lin_w.reserve(terms); lin_c.reserve(terms); }
  }
  const u32 n_lin = (u32)(lin_ptr.size() - 1);
  phase("collect linear constraints");
  // wire -> linear constraints containing it (only unknown wires matter)
  std::vector<u8> known(produced.begin(), produced.begin() + nw);
  known[0] = 1;
  std::vector<u32> deg(nw + 1, 0);
  for (u64 t = 0; t < lin_w.size(); ++t) if (!known[lin_w[t]]) ++deg[lin_w[t] + 1];
  for (u64 w = 0; w < nw; ++w) deg[w + 1] += deg[w];
  std::vector<u32> occ(deg[nw]);
  std::vector<u32> unk(n_lin, 0);
  {
    std::vector<u32> fill(deg.begin(), deg.end() - 1);
    for (u32 li = 0; li < n_lin; ++li) {
      u32 n = 0;
      for (u64 t = lin_ptr[li]; t < lin_ptr[li + 1]; ++t) if (!known[lin_w[t]]) { occ[fill[lin_w[t]]++] = li; ++n; }
      unk[li] = n;
    }
  }
  phase("occurrence lists");
  // flattened definition of every derived wire over produced wires (terms sorted by wire, merged): an append-only arena
  std::vector<u64> def_at(nw, 0);
  std::vector<u32> def_n(nw, 0);
  std::vector<u32> ar_w;
  std::vector<Fr> ar_c;
  ar_w.reserve(nw + nw / 4); ar_c.reserve(nw + nw / 4);
  std::vector<u32> order;
  order.reserve(nw);
  // shortest constraint first: a wire that is both an alias of a produced signal and a member of a long sum
  // (the bits under a Num2Bits / BinSum closing constraint) must be defined by the alias, not by solving the sum for it
  typedef std::pair<u32, u32> QE;   // (terms of the constraint, index)
  std::priority_queue<QE, std::vector<QE>, std::greater<QE>> queue;
  auto lin_len = [&](u32 li) { return (u32)(lin_ptr[li + 1] - lin_ptr[li]); };
  for (u32 li = 0; li < n_lin; ++li) if (unk[li] == 1) queue.emplace(lin_len(li), li);
  std::vector<std::pair<u32, Fr>> acc;
  while (!queue.empty()) {
    const u32 li = queue.top().second;
    queue.pop();
    if (unk[li] != 1) continue;
    const u64 l0 = lin_ptr[li], l1 = lin_ptr[li + 1];
    // the single unknown (it may appear more than once in the combination)
    u32 u = 0xffffffffu;
    Fr cu = zero;
    for (u64 t = l0; t < l1; ++t)
      if (!known[lin_w[t]]) { u = lin_w[t]; cu = fr_add(cu, lin_c[t]); }
    if (u == 0xffffffffu) continue;
    if (fr_is_zero(cu)) { unk[li] = 0; continue; }           // cancels out: not a definition
    // u = -(1 / cu) * sum_{others} c_w * w, with derived wires substituted by their definitions
    // -(1 / cu), Montgomery form; the coefficient of an alias / sum member is almost always +-1
    const bool f_one = fr_eq(cu, neg_unit_m), f_minus = fr_eq(cu, unit_m);
    const Fr f = f_minus ? neg_unit_m : (f_one ? unit_m : fr_neg(fr_mont_inv(cu)));
    auto times_f = [&](const Fr& x) { return f_one ? x : (f_minus ? fr_neg(x) : fr_mont_mul(f, x)); };
    acc.clear();
    for (u64 t = l0; t < l1; ++t) {
      const u32 w = lin_w[t];
      if (w == u) continue;
      const Fr k = times_f(lin_c[t]);
      if (produced[w]) acc.emplace_back(w, k);
      else {
        const bool k_one = fr_eq(k, unit_m), k_minus = fr_eq(k, neg_unit_m);
        for (u64 q = def_at[w], qe = def_at[w] + def_n[w]; q < qe; ++q)
          acc.emplace_back(ar_w[q], k_one ? ar_c[q] : (k_minus ? fr_neg(ar_c[q]) : fr_mont_mul(k, ar_c[q])));
      }
    }
    if (acc.size() > 1) std::sort(acc.begin(), acc.end(), [](const std::pair<u32, Fr>& a, const std::pair<u32, Fr>& b) { return a.first < b.first; });
    def_at[u] = ar_w.size();
    for (size_t t = 0; t < acc.size();) {
      Fr sacc = acc[t].second;
      size_t q = t + 1;
      while (q < acc.size() && acc[q].first == acc[t].first) { sacc = fr_add(sacc, acc[q].second); ++q; }
      if (!fr_is_zero(sacc)) { ar_w.push_back(acc[t].first); ar_c.push_back(sacc); }
      t = q;
    }
    def_n[u] = (u32)(ar_w.size() - def_at[u]);
    known[u] = 1;
    order.push_back(u);
    for (u32 k = deg[u]; k < deg[u + 1]; ++k) {
      const u32 lj = occ[k];
      if (unk[lj] > 0 && --unk[lj] == 1) queue.emplace(lin_len(lj), lj);
    }
  }
  phase("elimination");
  for (u64 w = 0; w < nw; ++w)
    if (!known[w]) { err = "signal with witness index " + std::to_string(w) + " is neither produced by this schedule nor defined by a linear constraint of the .r1cs"; return false; }
  // rows sorted by destination (coalesced writes)
  std::sort(order.begin(), order.end());
  const Fr one_m = fr_R(), minus_one_m = fr_neg(fr_R());
  const Fr one_s = fr_from_u64(1), minus_one_s = fr_neg(fr_from_u64(1));
  P.row_ptr.assign(1, 0);
  P.dst.clear(); P.src.clear(); P.coef.clear(); P.kind.clear();
  P.row_ptr.reserve(order.size() + 1); P.dst.reserve(order.size());
  P.src.reserve(ar_w.size()); P.coef.reserve(ar_w.size()); P.kind.reserve(ar_w.size());
  for (u32 u : order) {
    for (u64 q = def_at[u], qe = def_at[u] + def_n[u]; q < qe; ++q) {
      const Fr& c = ar_c[q];
      const u8 kd = fr_eq(c, one_m) ? ZK_COEF_ONE : (fr_eq(c, minus_one_m) ? ZK_COEF_MINUS_ONE : ZK_COEF_GENERIC);
      P.src.push_back(ar_w[q]);
      P.coef.push_back(kd == ZK_COEF_ONE ? one_s : (kd == ZK_COEF_MINUS_ONE ? minus_one_s : fr_from_mont(c)));
      P.kind.push_back(kd);
    }
    P.dst.push_back(u);
    P.row_ptr.push_back(P.src.size());
  }
  phase("rows");
  return true;
}

// one row for one witness (standard-form values); used by the host evaluation of the CPU tests
ZK_HD Fr zk_linear_row(const u64* __restrict__ row_ptr, const u32* __restrict__ src, const Fr* __restrict__ coef,
                       const u8* __restrict__ kind, u64 r, const Fr* __restrict__ w) {
  Fr acc = fr_zero();
  for (u64 t = row_ptr[r]; t < row_ptr[r + 1]; ++t) {
    const Fr x = w[src[t]];
    const u8 k = kind[t];
    if (k == ZK_COEF_ONE) acc = fr_add(acc, x);
    else if (k == ZK_COEF_MINUS_ONE) acc = fr_sub(acc, x);
    else if (!fr_is_zero(x)) {
      // generic coefficient (powers of two of Bits2Num ...): the operand is almost always a bit
      const bool one = x.l[0] == 1 && (x.l[1] | x.l[2] | x.l[3]) == 0;
      acc = fr_add(acc, one ? coef[t] : fr_mont_mul(fr_to_mont(x), coef[t]));
    }
  }
  return acc;
}
