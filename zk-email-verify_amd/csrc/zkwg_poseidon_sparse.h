// Poseidon(t-1) for PoseidonModular (packages/circuits/utils/hash.circom:49-82; circomlib
// poseidon.circom [EXT]): x^5 S-box over BN254 Fr, 8 full + R_P partial rounds, one permutation per
// LANE (not per wave): RemoveSoftLineBreaks hashes 2*maxBody/16 independent 16-element chunks per
// email, so the batch supplies hundreds of thousands of independent permutations.
//
// The textbook partial round costs a dense t x t product; here every partial round is rewritten to a
// sparse one (2t-1 products).  With M = [[m00, v], [w, M^]] and B_k = diag(1, B^_k), B_0 = I:
//     z_k = B_k u_k,   M B_k = B_{k+1} S_{k+1},   S = [[n00, v'], [N^{-1} w', I]]   (N = M B_k)
//     u_{k+1} = S_{k+1} sigma(u_k + B_k^{-1} c_k)
// sigma (the S-box on element 0) commutes with B_k, so the S-box sees exactly the textbook values --
// the kept witness signals (Sigma.out, .in2, .in4) are unchanged; after the last partial round the
// dense B_{R_P} is applied once.  The state stays in STANDARD form; multiplicative constants are
// stored in Montgomery form (mont_mul(x_std, c_mont) = x*c in standard form), so every emitted value
// is already a witness value.
//
// Table layout (Fr elements) for a given t, built by zk_build_poseidon_sparse():
//   c_first[4][t] | mt[t][t] (mt[j][i] = M[i][j]) | c_part[rp][t] | s_part[rp][2t-1] = (n00, v[1..t), w^[1..t))
//   | bt[t][t] (bt[j][i] = B[i][j]) | c_last[4][t]
#pragma once
#include "zkwg_fr.h"

static const u32 ZK_POS_RP_TAB[16] = {56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68};  // t = 2..17
ZK_HD u32 zk_pos_tab_size(u32 t, u32 rp) { return 4 * t + t * t + rp * t + rp * (2 * t - 1) + t * t + 4 * t; }
// kept signals of one Poseidon(t-1) instance: (8 t + rp) S-boxes x (out, in2, in4)
ZK_HD u32 zk_pos_kept(u32 t, u32 rp) { return 3 * (8 * t + rp); }

// x (standard form) -> x^5, x^2, x^4 (standard form), 5 Montgomery products
ZK_HD Fr zk_sbox5(const Fr& x, Fr* emit) {
  const Fr xm = fr_mont_mul(x, fr_R2());
  const Fr in2 = fr_mont_mul(xm, x);
  const Fr in2m = fr_mont_mul(in2, fr_R2());
  const Fr in4 = fr_mont_mul(in2m, in2);
  const Fr out = fr_mont_mul(xm, in4);
  emit[0] = out; emit[1] = in2; emit[2] = in4;
  return out;
}

// One permutation.  `st`: T state elements with stride `ss` (standard form, element 0 = capacity);
// `emit`: 3 * (8T + rp) Fr, Sigma signals in component order (sigmaF[8][T], sigmaP[rp]);
// `tmp`: T Fr of scratch with stride `ts`.  Returns out[0].
// st[i] = sum_j y_j * mat[j*T + i] for all i, y = the current st (dense product, lazily reduced); `tmp` = T Fr of
// per-lane scratch with stride `ts`
template <int T>
ZK_HD void zk_pos_dense(Fr* st, const u32 ss, const Fr* __restrict__ mat, Fr* tmp, const u32 ts) {
  for (u32 i = 0; i < (u32)T; ++i) {
    FrWide w;
    fr_wide_zero(w);
    for (u32 j = 0; j < (u32)T; ++j) fr_wide_mac(w, st[j * ss], mat[j * T + i]);
    tmp[i * ts] = fr_wide_redc(w);
  }
  for (u32 i = 0; i < (u32)T; ++i) st[i * ss] = tmp[i * ts];
}

template <int T>
ZK_HD Fr zk_poseidon_sparse(Fr* st, const u32 ss, const Fr* __restrict__ tab, const u32 rp, Fr* emit, Fr* tmp, const u32 ts) {
  const Fr* c_first = tab;
  const Fr* mt = c_first + 4 * T;
  const Fr* c_part = mt + T * T;
  const Fr* s_part = c_part + rp * T;
  const Fr* bt = s_part + rp * (2 * T - 1);
  const Fr* c_last = bt + T * T;
  for (u32 half = 0; half < 2; ++half) {
    if (half == 1) {
      // partial rounds on u (u_0 lives in a register); the first-row dot product is reduced once
      Fr u0 = st[0];
      for (u32 k = 0; k < rp; ++k) {
        const Fr* ck = c_part + k * T;
        const Fr* sk = s_part + k * (2 * T - 1);
        const Fr y0 = zk_sbox5(fr_add(u0, ck[0]), emit + 3 * (8 * T + k));
        FrWide w;
        fr_wide_zero(w);
        fr_wide_mac(w, y0, sk[0]);
        for (u32 j = 1; j < (u32)T; ++j) {
          const Fr uj = fr_add(st[j * ss], ck[j]);
          fr_wide_mac(w, uj, sk[j]);
          st[j * ss] = fr_add(uj, fr_mont_mul(y0, sk[T - 1 + j]));
        }
        u0 = fr_wide_redc(w);
      }
      st[0] = u0;
      zk_pos_dense<T>(st, ss, bt, tmp, ts);   // z = B u
    }
    // full rounds: ark + S-box in place, then the dense mix
    for (u32 r = 0; r < 4; ++r) {
      const Fr* c = (half ? c_last : c_first) + r * T;
      for (u32 j = 0; j < (u32)T; ++j)
        st[j * ss] = zk_sbox5(fr_add(st[j * ss], c[j]), emit + 3 * ((half * 4 + r) * T + j));
      zk_pos_dense<T>(st, ss, mt, tmp, ts);
    }
  }
  return st[0];
}

#include <vector>
#include <utility>
// Host: the table above from the textbook constants C ((8+rp) x t) and M (t x t), both in Montgomery form
// (build_poseidon_constants).  Returns false if a sub-matrix is singular (cannot happen for an MDS M).
static inline bool zk_build_poseidon_sparse(u32 t, u32 rp, const std::vector<Fr>& C, const std::vector<Fr>& M,
                                            std::vector<Fr>& tab) {
  const u32 n = t - 1;
  auto inv_mat = [&](std::vector<Fr> a, std::vector<Fr>& out) -> bool {  // n x n, Montgomery form
    out.assign((size_t)n * n, fr_zero());
    for (u32 i = 0; i < n; ++i) out[i * n + i] = fr_R();
    for (u32 col = 0; col < n; ++col) {
      u32 piv = col;
      while (piv < n && fr_is_zero(a[piv * n + col])) ++piv;
      if (piv == n) return false;
      if (piv != col)
        for (u32 j = 0; j < n; ++j) { std::swap(a[piv * n + j], a[col * n + j]); std::swap(out[piv * n + j], out[col * n + j]); }
      const Fr pi = fr_mont_inv(a[col * n + col]);
      for (u32 j = 0; j < n; ++j) { a[col * n + j] = fr_mont_mul(a[col * n + j], pi); out[col * n + j] = fr_mont_mul(out[col * n + j], pi); }
      for (u32 r = 0; r < n; ++r) {
        if (r == col || fr_is_zero(a[r * n + col])) continue;
        const Fr f = a[r * n + col];
        for (u32 j = 0; j < n; ++j) {
          a[r * n + j] = fr_sub(a[r * n + j], fr_mont_mul(f, a[col * n + j]));
          out[r * n + j] = fr_sub(out[r * n + j], fr_mont_mul(f, out[col * n + j]));
        }
      }
    }
    return true;
  };
  tab.assign(zk_pos_tab_size(t, rp), fr_zero());
  Fr* c_first = tab.data();
  Fr* mt = c_first + 4 * t;
  Fr* c_part = mt + t * t;
  Fr* s_part = c_part + rp * t;
  Fr* bt = s_part + rp * (2 * t - 1);
  Fr* c_last = bt + t * t;
  for (u32 i = 0; i < 4 * t; ++i) { c_first[i] = fr_from_mont(C[i]); c_last[i] = fr_from_mont(C[(4 + rp) * t + i]); }
  for (u32 i = 0; i < t; ++i) for (u32 j = 0; j < t; ++j) mt[j * t + i] = M[i * t + j];
  std::vector<Fr> Bh((size_t)n * n, fr_zero()), Bi((size_t)n * n, fr_zero());  // B^_k and its inverse
  for (u32 i = 0; i < n; ++i) Bh[i * n + i] = Bi[i * n + i] = fr_R();
  std::vector<Fr> N((size_t)t * t), Nh((size_t)n * n), Nhi;
  for (u32 k = 0; k < rp; ++k) {
    // c'_k = B_k^{-1} c_{4+k}
    const Fr* ck = &C[(4 + k) * t];
    c_part[k * t] = fr_from_mont(ck[0]);
    for (u32 i = 0; i < n; ++i) {
      Fr a = fr_zero();
      for (u32 j = 0; j < n; ++j) a = fr_add(a, fr_mont_mul(Bi[i * n + j], ck[1 + j]));
      c_part[k * t + 1 + i] = fr_from_mont(a);
    }
    // N = M B_k
    for (u32 i = 0; i < t; ++i) {
      N[i * t] = M[i * t];
      for (u32 j = 0; j < n; ++j) {
        Fr a = fr_zero();
        for (u32 l = 0; l < n; ++l) a = fr_add(a, fr_mont_mul(M[i * t + 1 + l], Bh[l * n + j]));
        N[i * t + 1 + j] = a;
      }
    }
    for (u32 i = 0; i < n; ++i) for (u32 j = 0; j < n; ++j) Nh[i * n + j] = N[(1 + i) * t + 1 + j];
    if (!inv_mat(Nh, Nhi)) return false;
    Fr* sk = s_part + k * (2 * t - 1);
    sk[0] = N[0];
    for (u32 j = 0; j < n; ++j) sk[1 + j] = N[1 + j];
    for (u32 i = 0; i < n; ++i) {
      Fr a = fr_zero();
      for (u32 j = 0; j < n; ++j) a = fr_add(a, fr_mont_mul(Nhi[i * n + j], N[(1 + j) * t]));
      sk[t + i] = a;
    }
    Bh = Nh;
    Bi = Nhi;
  }
  bt[0] = fr_R();
  for (u32 i = 0; i < n; ++i) for (u32 j = 0; j < n; ++j) bt[(1 + j) * t + (1 + i)] = Bh[i * n + j];
  return true;
}
