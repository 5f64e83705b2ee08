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
  if (produced.size() < nw) { err = "the .sym file lists fewer signals than the .r1cs has wires"; return false; }
  auto row = [&](u64 lc, u64& a, u64& b) { a = R.row_ptr[lc]; b = R.row_ptr[lc + 1]; };
  const Fr zero = fr_zero();
  const Fr unit_m = fr_R(), neg_unit_m = fr_neg(fr_R());
  // linear constraints: A or B empty -> C = 0;  A (or B) a pure constant k -> k * B - C = 0
  struct Lin { std::vector<u32> w; std::vector<Fr> c; };   // sum c_i w_i = 0 (Montgomery coefficients)
  std::vector<Lin> lin;
  lin.reserve(m / 2);
  for (u64 i = 0; i < m; ++i) {
    u64 a0, a1, b0, b1, c0, c1;
    row(3 * i, a0, a1); row(3 * i + 1, b0, b1); row(3 * i + 2, c0, c1);
    const bool a_empty = a0 == a1, b_empty = b0 == b1;
    const bool a_const = a1 - a0 == 1 && R.wire[a0] == 0, b_const = b1 - b0 == 1 && R.wire[b0] == 0;
    if (!(a_empty || b_empty || a_const || b_const)) continue;
    Lin L;
    for (u64 t = c0; t < c1; ++t) { L.w.push_back(R.wire[t]); L.c.push_back(fr_neg(R.coef[t])); }
    if (!a_empty && !b_empty) {
      // k * other - C = 0
      const bool use_b = a_const;           // A constant: the other side is B
      const Fr k = use_b ? R.coef[a0] : R.coef[b0];
      const u64 o0 = use_b ? b0 : a0, o1 = use_b ? b1 : a1;
      for (u64 t = o0; t < o1; ++t) { L.w.push_back(R.wire[t]); L.c.push_back(fr_mont_mul(k, R.coef[t])); }
    }
    if (!L.w.empty()) lin.push_back(std::move(L));
  }
  // wire -> linear constraints containing it (only unknown wires matter)
  std::vector<u8> known(produced.begin(), produced.begin() + nw);
  known[0] = 1;
  std::vector<u32> deg(nw + 1, 0);
  for (const Lin& L : lin) for (u32 w : L.w) if (!known[w]) ++deg[w + 1];
  for (u64 w = 0; w < nw; ++w) deg[w + 1] += deg[w];
  std::vector<u32> occ(deg[nw]);
  {
    std::vector<u32> fill(deg.begin(), deg.end() - 1);
    for (u32 li = 0; li < lin.size(); ++li) for (u32 w : lin[li].w) if (!known[w]) occ[fill[w]++] = li;
  }
  std::vector<u32> unk(lin.size(), 0);
  for (u32 li = 0; li < lin.size(); ++li) { u32 n = 0; for (u32 w : lin[li].w) if (!known[w]) ++n; unk[li] = n; }
  // flattened definition of every derived wire over produced wires: terms sorted by wire, merged
  struct Def { std::vector<u32> w; std::vector<Fr> c; };
  std::vector<Def> def(nw);
  std::vector<u32> order;
  // shortest constraint first: a wire that is both an alias of a produced signal and a member of a long sum
  // (the bits under a Num2Bits / BinSum closing constraint) must be defined by the alias, not by solving the sum for it
  typedef std::pair<u32, u32> QE;   // (terms of the constraint, index)
  std::priority_queue<QE, std::vector<QE>, std::greater<QE>> queue;
  for (u32 li = 0; li < lin.size(); ++li) if (unk[li] == 1) queue.emplace((u32)lin[li].w.size(), li);
  std::vector<std::pair<u32, Fr>> acc;
  while (!queue.empty()) {
    const u32 li = queue.top().second;
    queue.pop();
    if (unk[li] != 1) continue;
    const Lin& L = lin[li];
    // the single unknown (it may appear more than once in the combination)
    u32 u = 0xffffffffu;
    Fr cu = zero;
    for (size_t t = 0; t < L.w.size(); ++t)
      if (!known[L.w[t]]) { u = L.w[t]; cu = fr_add(cu, L.c[t]); }
    if (u == 0xffffffffu) continue;
    if (fr_is_zero(cu)) { unk[li] = 0; continue; }           // cancels out: not a definition
    // u = -(1 / cu) * sum_{others} c_w * w, with derived wires substituted by their definitions
    // -(1 / cu), Montgomery form; the coefficient of an alias / sum member is almost always +-1
    const Fr f = fr_eq(cu, unit_m) ? neg_unit_m : (fr_eq(cu, neg_unit_m) ? unit_m : fr_neg(fr_mont_inv(cu)));
    acc.clear();
    for (size_t t = 0; t < L.w.size(); ++t) {
      const u32 w = L.w[t];
      if (w == u) continue;
      const Fr k = fr_mont_mul(f, L.c[t]);
      if (produced[w]) acc.emplace_back(w, k);
      else for (size_t q = 0; q < def[w].w.size(); ++q) acc.emplace_back(def[w].w[q], fr_mont_mul(k, def[w].c[q]));
    }
    std::sort(acc.begin(), acc.end(), [](const std::pair<u32, Fr>& a, const std::pair<u32, Fr>& b) { return a.first < b.first; });
    Def& D = def[u];
    for (size_t t = 0; t < acc.size();) {
      Fr s = acc[t].second;
      size_t q = t + 1;
      while (q < acc.size() && acc[q].first == acc[t].first) { s = fr_add(s, acc[q].second); ++q; }
      if (!fr_is_zero(s)) { D.w.push_back(acc[t].first); D.c.push_back(s); }
      t = q;
    }
    known[u] = 1;
    order.push_back(u);
    for (u32 k = deg[u]; k < deg[u + 1]; ++k) {
      const u32 lj = occ[k];
      if (unk[lj] > 0 && --unk[lj] == 1) queue.emplace((u32)lin[lj].w.size(), lj);
    }
  }
  for (u64 w = 0; w < nw; ++w)
    if (!known[w]) { err = "signal with witness index " + std::to_string(w) + " is neither produced by this schedule nor defined by a linear constraint of the .r1cs"; return false; }
  // rows sorted by destination (coalesced writes)
  std::sort(order.begin(), order.end());
  const Fr one_m = fr_R(), minus_one_m = fr_neg(fr_R());
  P.row_ptr.assign(1, 0);
  P.dst.clear(); P.src.clear(); P.coef.clear(); P.kind.clear();
  for (u32 u : order) {
    const Def& D = def[u];
    for (size_t t = 0; t < D.w.size(); ++t) {
      P.src.push_back(D.w[t]);
      P.coef.push_back(fr_from_mont(D.c[t]));
      P.kind.push_back(fr_eq(D.c[t], one_m) ? ZK_COEF_ONE : (fr_eq(D.c[t], minus_one_m) ? ZK_COEF_MINUS_ONE : ZK_COEF_GENERIC));
    }
    P.dst.push_back(u);
    P.row_ptr.push_back(P.src.size());
  }
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
