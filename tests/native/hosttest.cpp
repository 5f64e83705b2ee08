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
  HT_P29(3, 0); else HT_P29(3, 2); else HT_P29(3, 3); else HT_P29(3, 7);      // (3, 2): what zk_rslb_merge1 runs
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
// the DEVICE's product since round 5 (zkwg_comba29.h: 9 x 29-bit product scanning behind the 4 x 64-bit interface) on raw limbs, for Fq
// (field 0) and Fr (field 1): ADVICE r5 -- the host mirror ran the 64-bit CIOS only, so no CPU test executed the kernels' product
void ht_comba_mont_mul(int field, const void* a, const void* b, void* out) {
  if (field == 0) *(Fq*)out = fq_mont_mul_comba(*(const Fq*)a, *(const Fq*)b);
  else *(Fr*)out = fr_mont_mul_comba(*(const Fr*)a, *(const Fr*)b);
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

// The device multi-exponentiation (zkwg_msm_core.h: the per-thread bodies of the kernels of zkwg_kernels_msm.hip), executed here
// thread by thread in the launch order of zk_msm_launch_t, for E emails at once; `shuffle` permutes the thread order of the atomic
// passes, as the hardware may.  Scalars in standard form or (mont = 1) Montgomery form.  The point arithmetic is the lazy 29-bit limb
// form the device runs (zkwg_ec29.h; G2: both halves of a lane pair computed here), with every precondition of zkwg_fq29.h counted.
#define ZKWG_FQ29_CHECK 1
#include "zkwg_msm_core.h"
template <class C, class Aff, class Xyzz, class In, class Out, class Add, class Dbl, class ToAff, class ToTab, class IsInf>
static void ht_msm_mirror_t(const uint8_t* points, size_t pt_bytes, const uint64_t* scalars, uint64_t n, uint32_t E, uint32_t c, int mont, uint32_t shuffle, int layout,
                            uint32_t s0, uint8_t* out, In pt_in, Out pt_out, Xyzz inf, Add add_mixed, Dbl dbl, ToAff to_affine, ToTab to_table, IsInf is_inf) {
  typedef Xyzz29<typename C::F> X;
  // layout bit 0: ones apart (classification into index lists); bit 1: precomputed windows; bit 3: workgroup-local sort
  const bool ones_apart = (layout & 1) != 0, precomp = (layout & 2) != 0, lds_sort = (layout & 8) != 0;
  const u32 K = zk_msm_windows(c);
  std::vector<Aff> P(n), T((size_t)n * (precomp ? K : 1));
  for (uint64_t i = 0; i < n; ++i) P[i] = pt_in(points + pt_bytes * i);
  for (u32 i = 0; i < n; ++i) zk_msm_table_thread(P.data(), T.data(), (u32)n, c, precomp ? K : 1u, i, inf, add_mixed, dbl, to_affine, to_table);
  std::vector<u32> infb((n + 31) / 32 + 1, 0);
  for (u32 i = 0; i < n; ++i) if (is_inf(P[i])) infb[i >> 5] |= 1u << (i & 31);
  std::vector<Fr> S((size_t)n * E);
  for (uint64_t i = 0; i < n * E; ++i) { memcpy(&S[i], scalars + 4 * i, 32); if (mont) S[i] = fr_to_mont(S[i]); }
  ZkMsmArgsT<C> A;
  A.table = T.data(); A.inf = infb.data(); A.scalars = S.data(); A.scalar_stride = n;
  A.n = (u32)n; A.c = c; A.K = K; A.nb = 1u << (c - 1); A.KS = precomp ? 1u : K; A.stride = precomp ? (u32)n : 0u; A.E = E; A.scalars_mont = mont ? 1u : 0u;
  A.lds_sort = lds_sort ? 1u : 0u; A.s0 = s0;
  const u32 total = A.KS * A.nb;
  // one email's arrays, as zkwg_msm_api.hip lays them out (sizes in accumulators of this build: sizeof(X) per point)
  auto al = [](u64 x) { return (x + 255) & ~255ull; };
  const u64 xs = sizeof(X) * C::LANES;
  ZkMsmOff& W = A.off;
  {
    u64 off = 0;
    W.count = off; off += al((total + 1) * 4);
    W.cursor = off; off += al((u64)total * 4);
    W.entry = off; off += al((u64)n * K * 4);
    W.bucket = off; off += al((u64)total * xs);
    const u64 n0 = zk_msm_plane_n0(A.nb), n1 = (n0 + ZK_MSM_PFAN - 1) / ZK_MSM_PFAN;
    W.node_s = off; off += al((u64)A.KS * c * n0 * xs);
    W.node_a = off; off += al((u64)A.KS * c * n1 * xs);
    W.window = off; off += al((u64)A.KS * xs);
    W.ones = off; off += al(2 * ((n + ZK_MSM_ONES - 1) / ZK_MSM_ONES) * xs);
    u64 items = (u64)n * K;
    for (int l = 0; l < 3; ++l) {
      const u64 cap = items / (l == 0 ? s0 : ZK_MSM_S1) + total + 1;
      W.cap[l] = (u32)cap; W.soff[l] = off; off += al((total + 1) * 4); W.part[l] = off; off += al(cap * xs); items = cap;
    }
    W.total = al(off);
  }
  std::vector<u64> workbuf((W.total * E + 15) / 8 + 2, 0xdeadbeefdeadbeefull);
  A.work = (u8*)(((uintptr_t)workbuf.data() + 15) & ~(uintptr_t)15); A.work_stride = W.total;
  std::vector<typename C::Out> res(E);
  A.out = res.data();
  // thread orders of the atomic passes
  std::vector<u32> order(n);
  for (u32 i = 0; i < n; ++i) order[i] = i;
  u64 x = 0x9e3779b97f4a7c15ull * (shuffle + 1);
  if (shuffle) for (u64 i = n; i > 1; --i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(order[i - 1], order[x % i]); }
  // classification (zk_msm_classify: one target)
  std::vector<u32> sel((size_t)n * E + 1, 0xdeadbeefu), ones((size_t)n * E + 1, 0xdeadbeefu), cnt(2 * E, 0);
  A.sel = A.n_sel = A.ones = A.n_ones = nullptr; A.list_stride = n;
  if (ones_apart) {
    ZkClassifyArgs Q;
    Q.scalars = S.data(); Q.scalar_stride = n; Q.n = (u32)n; Q.E = E; Q.scalars_mont = A.scalars_mont; Q.ones_apart = 1; Q.n_targets = 1;
    Q.t[0] = ZkClassifyTarget{infb.data(), 0u, (u32)n, sel.data(), cnt.data(), ones.data(), cnt.data() + E, n};
    for (u32 e = 0; e < E; ++e) for (u32 i : order) zk_msm_classify_thread(Q, e, i);
    A.sel = sel.data(); A.n_sel = cnt.data(); A.ones = ones.data(); A.n_ones = cnt.data() + E;
  }
  std::vector<u32> hist(ZK_MSM_LDS_BUCKETS), partial(1025);
  for (u32 e = 0; e < E; ++e) {
    memset(A.count(e), 0, ((size_t)total + 1) * 4);
    const u32 len = A.sel_count(e);
    const bool wg_sort = lds_sort && total <= ZK_MSM_LDS_BUCKETS;
    const u32 n_wg = shuffle ? 7u : 3u;
    auto sort_wg = [&](bool scatter) {
      for (u32 wgi = 0; wgi < n_wg; ++wgi) {
        const u32 wg = shuffle ? n_wg - 1 - wgi : wgi;
        for (int phase = 0; phase < (scatter ? 4 : 3); ++phase)
          for (u32 t = 0; t < 64; ++t) zk_msm_sort_wg_thread(A, e, wg, n_wg, shuffle ? 63 - t : t, 64u, hist.data(), phase, scatter);
      }
    };
    if (wg_sort) sort_wg(false);
    else for (u32 j : order) if (j < len) zk_msm_count_thread(A, e, j);
    for (int phase = 0; phase < 2; ++phase) for (u32 t = 0; t < 1024; ++t) zk_msm_scan_thread(A, e, t, 1024u, partial.data(), phase);
    if (wg_sort) sort_wg(true);
    else for (u32 j : order) if (j < len) zk_msm_scatter_thread(A, e, j);
    for (int level = 0; level < 3; ++level) {
      for (int phase = 0; phase < 2; ++phase) for (u32 t = 0; t < 1024; ++t) zk_msm_slice_scan_thread(A, e, level, t, 1024u, partial.data(), phase);
      for (u32 t = 0, m = zk_msm_slice_count(A, e, level); t < m; ++t) { if (level == 0) zk_msm_slice_sum_thread<C, true>(A, e, level, t, 0); else zk_msm_slice_sum_thread<C, false>(A, e, level, t, 0); }
    }
    for (u32 b = 0; b < total; ++b) zk_msm_bucket_join_thread(A, e, b, 0);
    const u32 rows = A.KS * A.c;
    u32 n_in = zk_msm_plane_n0(A.nb), flip = 0;
    for (u32 g = 0; g < rows * n_in; ++g) zk_msm_plane0_thread(A, e, g, 0);
    while (n_in > 1) {
      const u32 n_out = (n_in + ZK_MSM_PFAN - 1) / ZK_MSM_PFAN;
      for (u32 g = 0; g < rows * n_out; ++g) zk_msm_plane_join_thread<C>(flip ? A.node_a(e) : A.node_s(e), rows, n_in, flip ? A.node_s(e) : A.node_a(e), g, 0);
      flip ^= 1u; n_in = n_out;
    }
    for (u32 w = 0; w < A.KS; ++w) zk_msm_plane_window_thread(A, e, flip ? A.node_a(e) : A.node_s(e), w, 0);
    u32 half = 0, in_second = 0;
    if (A.ones) {
      half = (u32)((n + ZK_MSM_ONES - 1) / ZK_MSM_ONES);
      X* lo = A.ones_acc(e);
      X* hi = lo + (u64)half * C::LANES;
      for (u32 t = 0, m = zk_msm_ones_parts(A.n_ones[e]); t < m; ++t) zk_msm_ones_thread(A, e, t, 0, lo);
      u32 m = half, level = 0;
      while (m > 1) {
        const u32 m2 = (m + ZK_MSM_JOIN - 1) / ZK_MSM_JOIN;
        const u32 cntl = (zk_msm_ones_level_count(A.n_ones[e], level) + ZK_MSM_JOIN - 1u) / ZK_MSM_JOIN;
        for (u32 t = 0; t < cntl; ++t) zk_msm_tree_thread(A, e, level, t, 0, in_second ? hi : lo, in_second ? lo : hi);
        in_second ^= 1u; m = m2; ++level;
      }
    }
    zk_msm_combine_thread(A, e, 0, A.ones ? A.ones_acc(e) + (in_second ? (u64)half * C::LANES : 0) : nullptr);
    pt_out(res[e], out + pt_bytes * e);
  }
}

extern "C" {
// group 1 / 2; E emails (scalars: E x n x 32 bytes); out: E points in standard form
void ht_msm_device_mirror_batch(int group, const uint8_t* points, const uint64_t* scalars, uint64_t n, uint32_t E, uint32_t c, int mont, uint32_t shuffle, int layout,
                                uint32_t s0, uint8_t* out) {
  if (group == 1)
    ht_msm_mirror_t<ZkEcG1, G1Affine, G1Xyzz>(points, 64, scalars, n, E, c, mont, shuffle, layout, s0, out, ht_pt_in, [](const G1Xyzz& r, uint8_t* o) { ht_pt_out(r, o); },
        g1_xyzz_inf(), [](const G1Xyzz& a, const G1Affine& p) { return g1_add_mixed(a, p); }, [](const G1Xyzz& a) { return g1_dbl(a); },
        [](const G1Xyzz& a) { return g1_to_affine(a); }, [](const G1Affine& p) { return zk_g1_to_table_form(p); }, [](const G1Affine& p) { return g1_is_inf(p); });
  else
    ht_msm_mirror_t<ZkEcG2, G2Affine, G2Xyzz>(points, 128, scalars, n, E, c, mont, shuffle, layout, s0, out, ht_p2_in,
        [](const G2Xyzz& r, uint8_t* o) { const G2Affine q = g2_to_affine(r); if (g2_is_inf(q)) { memset(o, 0, 128); return; } ht_f2_out(q.x, o); ht_f2_out(q.y, o + 64); },
        g2_xyzz_inf(), [](const G2Xyzz& a, const G2Affine& p) { return g2_add_mixed(a, p); }, [](const G2Xyzz& a) { return g2_dbl(a); },
        [](const G2Xyzz& a) { return g2_to_affine(a); }, [](const G2Affine& p) { return zk_g2_to_table_form(p); }, [](const G2Affine& p) { return g2_is_inf(p); });
}
void ht_msm_device_mirror(const uint8_t* points, const uint64_t* scalars, uint64_t n, uint32_t c, int mont, uint32_t shuffle, int layout, uint8_t* out) {
  ht_msm_device_mirror_batch(1, points, scalars, n, 1, c, mont, shuffle, layout, 16, out);
}
unsigned long long ht_fq29_violations() { return zk_fq29_violations; }

// The lazy-limb point arithmetic by itself (zkwg_ec29.h), standard form across this boundary.  op 0: scaled(a) + b mixed; 1: scaled(a) +
// scaled(b); 2: 2 a (affine); 3: 2 scaled(a); `reps` > 1 repeats op 0 / 1 / 3 on the running result (accumulated bounds: acc = acc + b ...)
static Fq29 ht_q29(const Fq& canon_r256) { return fq29_from_fq(zk_fq_r256_to_r261(canon_r256)); }
void ht_ec29_g1_op(int op, const uint8_t* a, const uint8_t* b, const uint8_t* scale, uint32_t reps, uint8_t* out) {
  typedef ZkF1 F;
  const G1Affine A = ht_pt_in(a), B = ht_pt_in(b);
  Fq s; memcpy(&s, scale, 32); s = fq_to_mont(s);
  auto scaled = [&](const G1Affine& p) {
    if (g1_is_inf(p)) return ec29_inf<F>();
    const Fq s2 = fq_mont_sqr(s), s3 = fq_mont_mul(s2, s);
    return Xyzz29<F>{ht_q29(fq_mont_mul(p.x, s2)), ht_q29(fq_mont_mul(p.y, s3)), ht_q29(s2), ht_q29(s3)};
  };
  const G1Affine Bt = zk_g1_to_table_form(B), At = zk_g1_to_table_form(A);
  const Aff29<F> Bl = ZkEcG1::load(&Bt, 0, false), Al = ZkEcG1::load(&At, 0, false);
  Xyzz29<F> r = scaled(A);
  for (uint32_t i = 0; i < (reps ? reps : 1); ++i) {
    if (op == 0) r = ec29_add_mixed<F>(r, Bl);
    else if (op == 1) r = ec29_add<F>(r, scaled(B));
    else if (op == 2) r = ec29_dbl_affine<F>(Al);
    else r = ec29_dbl<F>(r);
  }
  G1Xyzz o;
  ZkEcG1::store_out(&o, r, 0);
  ht_pt_out(o, out);
}
void ht_ec29_g2_op(int op, const uint8_t* a, const uint8_t* b, const uint8_t* scale, uint32_t reps, uint8_t* out) {
  typedef ZkF2 F;
  const G2Affine A = ht_p2_in(a), B = ht_p2_in(b);
  const Fq2 s = ht_f2_in(scale);
  auto q2 = [](const Fq2& v) { return Fq29x2{{ht_q29(v.c0), ht_q29(v.c1)}}; };
  auto scaled = [&](const G2Affine& p) {
    if (g2_is_inf(p)) return ec29_inf<F>();
    const Fq2 s2 = fq2_sqr(s), s3 = fq2_mul(s2, s);
    return Xyzz29<F>{q2(fq2_mul(p.x, s2)), q2(fq2_mul(p.y, s3)), q2(s2), q2(s3)};
  };
  const G2Affine Bt = zk_g2_to_table_form(B), At = zk_g2_to_table_form(A);
  const Aff29<F> Bl = ZkEcG2::load(&Bt, 0, false), Al = ZkEcG2::load(&At, 0, false);
  Xyzz29<F> r = scaled(A);
  for (uint32_t i = 0; i < (reps ? reps : 1); ++i) {
    if (op == 0) r = ec29_add_mixed<F>(r, Bl);
    else if (op == 1) r = ec29_add<F>(r, scaled(B));
    else if (op == 2) r = ec29_dbl_affine<F>(Al);
    else r = ec29_dbl<F>(r);
  }
  G2Xyzz o;
  ZkEcG2::store_out(&o, r, 0);
  const G2Affine q = g2_to_affine(o);
  if (g2_is_inf(q)) { memset(out, 0, 128); return; }
  ht_f2_out(q.x, out); ht_f2_out(q.y, out + 64);
}
// the field layer: op 0 mul, 1 dot2 (a b + c d), 2 sub<12,1> then norm, 3 to_fq of a lazy sum a + b + c + d; standard form in / out (2^261 form inside)
void ht_fq29_op(int op, const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, uint8_t* out) {
  auto in = [](const uint8_t* p) { Fq v; memcpy(&v, p, 32); return ht_q29(fq_to_mont(v)); };
  const Fq29 x = in(a), y = in(b), z = in(c), w = in(d);
  Fq29 r;
  if (op == 0) r = fq29_mul(x, y);
  else if (op == 1) r = fq29_dot2(x, y, z, w);
  else if (op == 2) r = fq29_norm(fq29_sub<12, 1>(x, y));
  else r = fq29_add(fq29_add(x, y), fq29_add(z, w));
  // back: the value is v 2^261 (op 0, 1: v 2^261 with v the product) -> x 2^256 -> standard
  const Fq m = fq29_to_fq<16>(fq29_mul(r, fq29_r256()));
  const Fq st = fq_from_mont(m);
  memcpy(out, &st, 32);
}
}
