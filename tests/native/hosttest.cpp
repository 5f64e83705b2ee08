// Test-only host build of the wave-collective RSA core (zkwg_rsa_core.h) and of the
// schedule builder.  NOT part of the product: it lets the CPU test-suite check the
// lane-parallel RSA code and the segment table against the oracle without a GPU.
#include <stdlib.h>
#include <vector>
#include "zkwg_build.h"
#include "zkwg_rsa_core.h"
#include "zkwg_fpmul_core.h"
#include "zkwg_poseidon_core.h"
#include "zkwg_poseidon_sparse.h"
#define ZKWG_P29_CHECK 1     // count violations of the limb-form evaluator's range argument (zkwg_poseidon29.h)
#include "zkwg_poseidon29.h"
#include "zkwg_r1cs.h"
#include "zkwg_regex_core.h"

struct HT {
  ZkSched s;
  std::vector<ZkSeg> segs;
  std::vector<u32> first;
};

extern "C" {
void* ht_create(const zkwg_config* cfg) {
  HT* h = new HT();
  if (!build_sched(*cfg, h->s, h->segs)) { delete h; return nullptr; }
  return h;
}
// `.sym`-ordered variant of the same schedule (zkwg_build.h zk_sym_layout + zk_remap_segments)
void* ht_create_sym(const zkwg_config* cfg, const char* sym, uint64_t len) {
  HT* h = new HT();
  ZkSymLayout L;
  if (!build_sched(*cfg, h->s, h->segs) || !zk_sym_layout(h->s, sym, len, nullptr, 0, L) ||
      !zk_remap_segments(h->s, h->segs, L)) { delete h; return nullptr; }
  return h;
}
void ht_destroy(void* p) { delete (HT*)p; }
uint64_t ht_W(void* p) { return ((HT*)p)->s.W; }
uint32_t ht_nsegs(void* p) { return ((HT*)p)->s.nsegs; }
const void* ht_segs(void* p) { return ((HT*)p)->segs.data(); }
uint32_t ht_img_bits(void* p) { return ((HT*)p)->s.img_bits; }
uint32_t ht_img_small(void* p) { return ((HT*)p)->s.img_small; }
uint32_t ht_img_fr(void* p) { return ((HT*)p)->s.img_fr; }
uint32_t ht_in_stride(void* p) { return ((HT*)p)->s.in_stride; }
uint32_t ht_in_off(void* p, int f) { return ((HT*)p)->s.in_off[f]; }
uint32_t ht_inv_half(void* p) { return ((HT*)p)->s.inv_half; }
uint32_t ht_m_one(void* p) { return ((HT*)p)->s.m_one; }
// run the RSA block of one email on the host; digest = 8 state words or NULL
int ht_run_rsa(void* p, const uint8_t* rec, const uint32_t* digest, uint64_t* bits, uint32_t* small, void* frv) {
  HT* h = (HT*)p;
  ZkRsaLds* S = new ZkRsaLds();
  u32 lt_eq[40];
  zk_rsa_email(*S, h->s.rsa, rec, digest, bits, small, (Fr*)frv, lt_eq);
  int ok = (int)S->ok;
  delete S;
  return ok;
}
// main = FpMul(n, k) with small parameters: the core zk_fpmul_small runs (zkwg_fpmul_core.h); returns the status
int ht_run_fpmul(void* p, const uint8_t* rec, uint64_t* bits, uint32_t* small, void* frv) {
  HT* h = (HT*)p;
  if (!h->s.fpg.present) return -1;
  return zk_fpmul_small_core(h->s.fpg, h->s.m_one, rec, bits, small, (Fr*)frv);
}
// PoseidonLarge(121,17) of 17 x 16-byte limbs: out420 = S-box signals, hash = pubkeyHash
void ht_poseidon(const uint8_t* limbs, void* out420, void* hash) {
  std::vector<Fr> C, M;
  build_poseidon_constants(10, 8, 60, C, M);
  ZkPosLds* S = new ZkPosLds();
  u64 l[17][2];
  memcpy(l, limbs, sizeof(l));
  zk_poseidon_large(*S, l, C.data(), M.data(), (Fr*)out420, (Fr*)hash);
  delete S;
}
// Poseidon(t-1) through the product's sparse-partial-round evaluator (zkwg_poseidon_sparse.h):
// inputs = t-1 standard-form Fr, emit = 3*(8t+rp) Fr, returns 0 on success
int ht_poseidon_sparse(uint32_t t, const void* inputs, void* emit, void* hash) {
  const u32 rp = ZK_POS_RP_TAB[t - 2];
  std::vector<Fr> C, M, tab;
  build_poseidon_constants(t, 8, rp, C, M);
  if (!zk_build_poseidon_sparse(t, rp, C, M, tab)) return 1;
  std::vector<Fr> st(t, fr_zero());
  memcpy(&st[1], inputs, (t - 1) * sizeof(Fr));
  Fr h;
  Fr tmp[17];
  if (t == 3) h = zk_poseidon_sparse<3>(st.data(), 1, tab.data(), rp, (Fr*)emit, tmp, 1);
  else if (t == 17) h = zk_poseidon_sparse<17>(st.data(), 1, tab.data(), rp, (Fr*)emit, tmp, 1);
  else return 2;
  *(Fr*)hash = h;
  return 0;
}
unsigned long long ht_p29_violations() { return zk_p29_violations; }
// the same permutation through the 29-bit-limb evaluator zk_rslb_chunks runs (zkwg_poseidon29.h); state laid out limb-major as in LDS
int ht_poseidon29(uint32_t t, uint32_t variant, const void* inputs, void* emit, void* hash) {
  const u32 rp = ZK_POS_RP_TAB[t - 2];
  std::vector<Fr> C, M, tab;
  std::vector<u32> tab29;
  build_poseidon_constants(t, 8, rp, C, M);
  if (!zk_build_poseidon_sparse(t, rp, C, M, tab)) return 1;
  zk_build_poseidon29(t, rp, tab, tab29);
  std::vector<u32> st(9 * t, 0u);
  for (u32 j = 1; j < t; ++j) {
    u32 l[9];
    zk_l29_from_fr(((const Fr*)inputs)[j - 1], l);
    for (u32 i = 0; i < 9; ++i) st[i * t + j] = l[i];
  }
  Fr h;
  std::vector<u32> stage(9 * 17, 0xdeadbeefu);
#define HT_P29(TT, VV) if (t == TT && variant == VV) h = zk_poseidon29<TT, VV>(st.data(), 1, t, tab29.data(), rp, (Fr*)emit, stage.data(), 1)
  HT_P29(3, 0); else HT_P29(3, 3); else HT_P29(3, 7);
  else HT_P29(17, 0); else HT_P29(17, 1); else HT_P29(17, 2); else HT_P29(17, 3);
  else HT_P29(17, 4); else HT_P29(17, 5); else HT_P29(17, 6); else HT_P29(17, 7);
#undef HT_P29
  else return 2;
  *(Fr*)hash = h;
  return 0;
}
// `.r1cs` reader + check core (zkwg_r1cs.h) on the host: index of the first violated constraint, -1 if none,
// -2 on a parse error
long long ht_r1cs_first_bad(const uint8_t* data, uint64_t len, const void* witness) {
  ZkR1csHost R;
  if (!zk_r1cs_parse(data, len, R)) return -2;
  for (u64 i = 0; i < R.n_constraints; ++i)
    if (!zk_r1cs_check_one(R.row_ptr.data(), R.wire.data(), R.coef.data(), R.kind.data(), i, (const Fr*)witness)) return (long long)i;
  return -1;
}
// the product's DFA scan (zkwg_regex_core.h) on the host: rev[n], own[2(n+1) + NP n + n]; returns accept count
uint32_t ht_regex_scan(const uint8_t* msg, uint32_t n, uint32_t* rev, uint32_t* own) {
  std::vector<u8> st(n + 4), live(n + 4);
  return zk_bh_dfa_scan(msg, n, ZK_DFA_DELTA, st.data(), live.data(), own, rev);
}
// Fr helpers for unit tests
void ht_fr_mul(const void* a, const void* b, void* out) { *(Fr*)out = fr_mul_std(*(const Fr*)a, *(const Fr*)b); }
void ht_fr_inv(const void* a, void* out) { *(Fr*)out = fr_inv_std(*(const Fr*)a); }
}
#include "zkwg_fr_inv.h"
extern "C" void ht_fr_inv_by(const void* a, void* out) { *(Fr*)out = fr_inv_by(*(const Fr*)a); }
// Loaded regex template (zkwg_circom.h) + the host evaluation shared with the kernels and the loader's self-check (zkwg_net_host.h)
#include "zkwg_net_host.h"
struct HTNet { zkc::Net net; std::string err; std::string names; };
extern "C" {
void* ht_net_load(const char* path, const char* include_dirs, const char* tname, uint32_t n) {
  HTNet* h = new HTNet();
  if (!zkc::load(path, include_dirs ? include_dirs : "", tname, {(zkc::i64)n}, h->net, h->err)) h->net.n_kept = 0xffffffffu;
  else for (auto& nm : h->net.names) { h->names += nm; h->names += '\n'; }
  return h;
}
void ht_net_destroy(void* p) { delete (HTNet*)p; }
const char* ht_net_error(void* p) { return ((HTNet*)p)->net.n_kept == 0xffffffffu ? ((HTNet*)p)->err.c_str() : nullptr; }
uint32_t ht_net_kept(void* p) { return ((HTNet*)p)->net.n_kept; }
uint32_t ht_net_inv_need(void* p) { return ((HTNet*)p)->net.inv_need; }
// what the loader did with the recurrences: positions covered by the forward / backward chain tables, steps left in the gate list
void ht_net_chain_info(void* p, uint32_t out[4]) {
  const zkc::Net& N = ((HTNet*)p)->net;
  out[0] = N.chain.end; out[1] = N.bchain.end; out[2] = N.n_steps; out[3] = N.chain.classes | (N.bchain.classes << 16);
}
const char* ht_net_names(void* p) { return ((HTNet*)p)->names.c_str(); }
// evaluates the template exactly as the kernels do (zkwg_net_host.h: the code the loader's self-check runs); words[n_kept], reveal[n]
int ht_net_eval(void* p, const uint8_t* msg, uint32_t* words, uint32_t* match, uint32_t* reveal) {
  return zkc::eval_host(((HTNet*)p)->net, msg, words, match, reveal);
}
// the loader's self-check on this template: 1 = tables and gate list agree, 0 = not (ht_net_check_error has the message)
int ht_net_self_check(void* p, const char* path, const char* include_dirs, const char* tname, uint32_t n) {
  HTNet* h = (HTNet*)p;
  std::string err;
  if (zkc::self_check(path, include_dirs ? include_dirs : "", tname, {(zkc::i64)n}, h->net, err, 3)) return 1;
  h->err = err;
  return 0;
}
const char* ht_net_check_error(void* p) { return ((HTNet*)p)->err.c_str(); }
// damages the forward chain's transition table (every entry that is not already 0 becomes 0): what the self-check exists to catch
void ht_net_damage_chain(void* p) {
  for (auto& d : ((HTNet*)p)->net.chain.delta) d = 0;
}
}

// BN254 G1 building blocks of the multi-exponentiation row (zkwg_fq.h, zkwg_g1.h): every value crosses this boundary in STANDARD form,
// 32-byte little-endian; points are x | y, infinity = all zeros
#include "zkwg_g1.h"
static G1Affine ht_pt_in(const uint8_t* p) {
  G1Affine a;
  memcpy(&a.x, p, 32); memcpy(&a.y, p + 32, 32);
  if (g1_is_inf(a)) return a;
  return G1Affine{fq_to_mont(a.x), fq_to_mont(a.y)};
}
static void ht_pt_out(const G1Xyzz& r, uint8_t* out) {
  const G1Affine a = g1_to_affine(r);
  const Fq x = g1_is_inf(a) ? a.x : fq_from_mont(a.x), y = g1_is_inf(a) ? a.y : fq_from_mont(a.y);
  memcpy(out, &x, 32); memcpy(out + 32, &y, 32);
}
extern "C" {
// Montgomery product on raw limbs through the 32-bit-limb path (the device's) or the 64-bit one
void ht_fq_mont_mul(const void* a, const void* b, void* out, int path32) {
  *(Fq*)out = path32 ? fq_mont_mul_32(*(const Fq*)a, *(const Fq*)b) : fq_mont_mul_64(*(const Fq*)a, *(const Fq*)b);
}
void ht_fq_op(int op, const void* a, const void* b, void* out) {   // standard form in and out: 0 add, 1 sub, 2 mul, 3 inv, 4 neg
  const Fq x = fq_to_mont(*(const Fq*)a), y = fq_to_mont(*(const Fq*)b);
  Fq r = op == 0 ? fq_add(x, y) : op == 1 ? fq_sub(x, y) : op == 2 ? fq_mont_mul(x, y) : op == 3 ? fq_mont_inv(x) : fq_neg(x);
  *(Fq*)out = fq_from_mont(r);
}
// op 0: (a as accumulator) + b mixed; 1: a + b with both in XYZZ, b scaled to a non-trivial ZZ first; 2: 2 a (affine); 3: 2 a (XYZZ, scaled)
void ht_g1_op(int op, const uint8_t* a, const uint8_t* b, const uint8_t* scale, uint8_t* out) {
  const G1Affine A = ht_pt_in(a), B = ht_pt_in(b);
  Fq s; memcpy(&s, scale, 32); s = fq_to_mont(s);
  // the same point with ZZ = s^2, ZZZ = s^3: X = x s^2, Y = y s^3
  auto scaled = [&](const G1Affine& p) {
    if (g1_is_inf(p)) return g1_xyzz_inf();
    const Fq s2 = fq_mont_sqr(s), s3 = fq_mont_mul(s2, s);
    return G1Xyzz{fq_mont_mul(p.x, s2), fq_mont_mul(p.y, s3), s2, s3};
  };
  G1Xyzz r;
  if (op == 0) r = g1_add_mixed(scaled(A), B);
  else if (op == 1) r = g1_add(scaled(A), g1_add_mixed(g1_xyzz_inf(), B));
  else if (op == 2) r = g1_dbl_affine(A);
  else if (op == 4) r = g1_add(scaled(A), scaled(B));
  else r = g1_dbl(scaled(A));
  ht_pt_out(r, out);
}
int ht_g1_on_curve(const uint8_t* a) { return g1_on_curve(ht_pt_in(a)) ? 1 : 0; }
// digits of one scalar: out[K]; returns the final carry (must be 0)
uint32_t ht_msm_digits(const uint64_t* k, uint32_t c, int32_t* out) {
  u32 carry = 0;
  for (u32 w = 0; w < zk_msm_windows(c); ++w) out[w] = zk_msm_digit(k, w, c, carry);
  return carry;
}
uint32_t ht_msm_windows(uint32_t c) { return zk_msm_windows(c); }
void ht_msm(const uint8_t* points, const uint64_t* scalars, uint64_t n, uint32_t c, uint8_t* out) {
  std::vector<G1Affine> P(n);
  for (uint64_t i = 0; i < n; ++i) P[i] = ht_pt_in(points + 64 * i);
  ht_pt_out(zk_msm_host(P.data(), scalars, n, c), out);
}
}

// The device multi-exponentiation (zkwg_msm_core.h: the per-thread bodies of the kernels of zkwg_kernels_msm.hip), executed here
// thread by thread in the launch order of zk_msm_launch; `shuffle` permutes the thread order of the two atomic passes, as the
// hardware may.  Scalars in standard form or (mont = 1) Montgomery form.
#include "zkwg_msm_core.h"
extern "C" void ht_msm_device_mirror(const uint8_t* points, const uint64_t* scalars, uint64_t n, uint32_t c, int mont, uint32_t shuffle, int ones_apart, uint8_t* out) {
  // ones_apart bit 1: the precomputed-windows layout (K copies of the bases, one bucket set), as zkwg_msm_create builds it by default
  const bool precomp = (ones_apart & 2) != 0, planes = (ones_apart & 4) != 0, lds_sort = (ones_apart & 8) != 0;
  ones_apart &= 1;
  std::vector<G1Affine> P(n);
  for (uint64_t i = 0; i < n; ++i) P[i] = ht_pt_in(points + 64 * i);
  std::vector<G1Affine> EXT;
  if (precomp) {
    EXT.resize((size_t)n * zk_msm_windows(c));
    for (u32 i = 0; i < n; ++i) zk_msm_shift_thread<ZkCurveG1>(P.data(), EXT.data(), (u32)n, c, zk_msm_windows(c), i, [](const G1Xyzz& a) { return g1_to_affine(a); });
  }
  std::vector<Fr> S(n);
  for (uint64_t i = 0; i < n; ++i) { memcpy(&S[i], scalars + 4 * i, 32); if (mont) S[i] = fr_to_mont(S[i]); }
  ZkMsmArgs A;
  A.bases = precomp ? EXT.data() : P.data(); A.scalars = S.data(); A.n = (u32)n; A.c = c; A.K = zk_msm_windows(c); A.nb = 1u << (c - 1); A.scalars_mont = mont ? 1u : 0u;
  A.KS = precomp ? 1u : A.K; A.stride = precomp ? (u32)n : 0u;
  const u32 total = A.KS * A.nb, half = A.KS * ((A.nb + ZK_MSM_FAN - 1) / ZK_MSM_FAN);
  std::vector<u32> count(total + 1, 0), cursor(total, 0), entry((size_t)n * A.K + 1, 0xdeadbeefu);
  const size_t pn0 = zk_msm_plane_n0(A.nb), pn1 = (pn0 + ZK_MSM_PFAN - 1) / ZK_MSM_PFAN;
  std::vector<G1Xyzz> bucket(total), ns(std::max<size_t>(2 * (size_t)half, (size_t)A.KS * c * pn0) + 1), na(std::max<size_t>(2 * (size_t)half, (size_t)A.KS * c * pn1) + 1), window(A.KS), res(1);
  A.plane_sums = planes ? 1u : 0u;
  A.count = count.data(); A.cursor = cursor.data(); A.entry = entry.data(); A.bucket = bucket.data();
  A.node_s = ns.data(); A.node_a = na.data(); A.window = window.data(); A.out = res.data();
  const u32 half1 = (u32)((n + ZK_MSM_ONES - 1) / ZK_MSM_ONES);
  std::vector<G1Xyzz> ones(2 * (size_t)half1 + 1);
  A.ones_apart = ones_apart ? 1u : 0u; A.ones = ones.data();
  std::vector<u32> order(n);
  for (u32 i = 0; i < n; ++i) order[i] = i;
  u64 x = 0x9e3779b97f4a7c15ull * (shuffle + 1);
  if (shuffle) for (u64 i = n; i > 1; --i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(order[i - 1], order[x % i]); }
  // the counting sort: one global atomic per digit, or (lds_sort, when the counters fit LDS) workgroup-local histograms -- as zk_msm_launch
  const bool wg_sort = lds_sort && total <= ZK_MSM_LDS_BUCKETS;
  const u32 per_wg = zk_msm_sort_per_wg(A.n) > 64 && shuffle ? 37u : zk_msm_sort_per_wg(A.n);      // (small odd workgroups under `shuffle`: more of them, ragged ends)
  const u32 n_wg = (A.n + per_wg - 1) / per_wg;
  std::vector<u32> hist(ZK_MSM_LDS_BUCKETS);
  auto sort_wg = [&](bool scatter) {
    for (u32 wgi = 0; wgi < n_wg; ++wgi) {
      const u32 wg = shuffle ? n_wg - 1 - wgi : wgi;
      for (int phase = 0; phase < (scatter ? 4 : 3); ++phase)
        for (u32 t = 0; t < 64; ++t) zk_msm_sort_wg_thread(A, wg, per_wg, shuffle ? 63 - t : t, 64u, hist.data(), phase, scatter);
    }
  };
  if (wg_sort) sort_wg(false);
  else for (u32 i : order) zk_msm_count_thread(A, i);
  std::vector<u32> partial(1025);
  for (int phase = 0; phase < 2; ++phase) for (u32 t = 0; t < 1024; ++t) zk_msm_scan_thread(A, t, 1024u, partial.data(), phase);
  if (wg_sort) sort_wg(true);
  else for (u32 i : order) zk_msm_scatter_thread(A, i);
  // the buckets' runs in slices (three levels), then one join per bucket -- as zk_msm_launch
  std::vector<std::vector<u32>> soff(3, std::vector<u32>(total + 1, 0xdeadbeefu));
  std::vector<std::vector<G1Xyzz>> part(3);
  {
    u64 items = (u64)n * A.K;
    for (int l = 0; l < 3; ++l) { const u64 cap = items / zk_msm_slice_size(l) + total + 1; A.cap[l] = (u32)cap; part[l].resize(cap); A.soff[l] = soff[l].data(); A.part[l] = part[l].data(); items = cap; }
  }
  for (int level = 0; level < 3; ++level) {
    for (int phase = 0; phase < 2; ++phase) for (u32 t = 0; t < 1024; ++t) zk_msm_slice_scan_thread(A, level, t, 1024u, partial.data(), phase);
    for (u32 t = 0; t < A.cap[level]; ++t) zk_msm_slice_sum_thread(A, level, t);
  }
  for (u32 b = 0; b < total; ++b) zk_msm_bucket_join_thread(A, b);
  const G1Xyzz* in_s = A.bucket; const G1Xyzz* in_a = nullptr;
  if (A.plane_sums) {     // as zk_msm_launch does
    const u32 rows = A.KS * A.c;
    u32 n_in = zk_msm_plane_n0(A.nb);
    for (u32 g = 0; g < rows * n_in; ++g) zk_msm_plane0_thread(A, g, A.node_s);
    G1Xyzz* cur = A.node_s;
    while (n_in > 1) {
      const u32 n_out = (n_in + ZK_MSM_PFAN - 1) / ZK_MSM_PFAN;
      G1Xyzz* nxt = cur == A.node_s ? A.node_a : A.node_s;
      for (u32 g = 0; g < rows * n_out; ++g) zk_msm_plane_join_thread<ZkCurveG1>(cur, rows, n_in, nxt, g);
      cur = nxt; n_in = n_out;
    }
    for (u32 w = 0; w < A.KS; ++w) zk_msm_plane_window_thread(A, cur, w);
  } else {
    u32 n_in = A.nb, span = 1, flip = 0;
    for (;;) {
      const u32 n_out = (n_in + ZK_MSM_FAN - 1) / ZK_MSM_FAN;
      G1Xyzz* out_s = A.node_s + (size_t)flip * half;
      G1Xyzz* out_a = A.node_a + (size_t)flip * half;
      for (u32 g = 0; g < A.KS * n_out; ++g) zk_msm_reduce_thread(A, g, in_s, in_a, n_in, span, out_s, out_a);
      if (n_out == 1) break;
      in_s = out_s; in_a = out_a; n_in = n_out; span *= ZK_MSM_FAN; flip ^= 1;
    }
  }
  if (A.ones_apart) {     // as zk_msm_launch does
    u32 m = half1, levels = 0;
    for (u32 q = m; q > 1; q = (q + ZK_MSM_JOIN - 1) / ZK_MSM_JOIN) ++levels;
    G1Xyzz* cur = A.ones + ((levels & 1u) ? half1 : 0);
    { ZkMsmArgs B = A; B.ones = cur; for (u32 t = 0; t < half1; ++t) zk_msm_ones_thread(B, t); }
    while (m > 1) {
      const u32 m2 = (m + ZK_MSM_JOIN - 1) / ZK_MSM_JOIN;
      G1Xyzz* nxt = cur == A.ones ? A.ones + half1 : A.ones;
      for (u32 t = 0; t < m2; ++t) zk_msm_tree_thread(cur, m, nxt, t);
      cur = nxt; m = m2;
    }
  }
  zk_msm_combine_thread(A);
  ht_pt_out(res[0], out);
}

// BN254 G2 building blocks (zkwg_g2.h): standard form across this boundary; a point is x.c0 | x.c1 | y.c0 | y.c1, zeros = infinity
#include "zkwg_g2.h"
static Fq2 ht_f2_in(const uint8_t* p) { Fq2 a; memcpy(&a.c0, p, 32); memcpy(&a.c1, p + 32, 32); return Fq2{fq_to_mont(a.c0), fq_to_mont(a.c1)}; }
static void ht_f2_out(const Fq2& a, uint8_t* out) { const Fq x = fq_from_mont(a.c0), y = fq_from_mont(a.c1); memcpy(out, &x, 32); memcpy(out + 32, &y, 32); }
static G2Affine ht_p2_in(const uint8_t* p) {
  bool zero = true;
  for (int i = 0; i < 128; ++i) zero = zero && p[i] == 0;
  if (zero) return G2Affine{fq2_zero(), fq2_zero()};
  return G2Affine{ht_f2_in(p), ht_f2_in(p + 64)};
}
extern "C" {
void ht_fq2_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {   // 0 add, 1 sub, 2 mul, 3 sqr, 4 inv
  const Fq2 x = ht_f2_in(a), y = ht_f2_in(b);
  ht_f2_out(op == 0 ? fq2_add(x, y) : op == 1 ? fq2_sub(x, y) : op == 2 ? fq2_mul(x, y) : op == 3 ? fq2_sqr(x) : fq2_inv(x), out);
}
// op 0: scaled(a) + b mixed; 1: scaled(a) + scaled(b); 2: 2 a (affine); 3: 2 scaled(a)
void ht_g2_op(int op, const uint8_t* a, const uint8_t* b, const uint8_t* scale, uint8_t* out) {
  const G2Affine A = ht_p2_in(a), B = ht_p2_in(b);
  const Fq2 s = ht_f2_in(scale);
  auto scaled = [&](const G2Affine& p) {
    if (g2_is_inf(p)) return g2_xyzz_inf();
    const Fq2 s2 = fq2_sqr(s), s3 = fq2_mul(s2, s);
    return G2Xyzz{fq2_mul(p.x, s2), fq2_mul(p.y, s3), s2, s3};
  };
  const G2Xyzz r = op == 0 ? g2_add_mixed(scaled(A), B) : op == 1 ? g2_add(scaled(A), scaled(B)) : op == 2 ? g2_dbl_affine(A) : g2_dbl(scaled(A));
  const G2Affine q = g2_to_affine(r);
  if (g2_is_inf(q)) { memset(out, 0, 128); return; }
  ht_f2_out(q.x, out); ht_f2_out(q.y, out + 64);
}
int ht_g2_on_curve(const uint8_t* a) { return g2_on_curve(ht_p2_in(a)) ? 1 : 0; }
}
