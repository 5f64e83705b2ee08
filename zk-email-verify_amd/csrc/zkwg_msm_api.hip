// C-ABI of the multi-exponentiations of the prover (include/zkwg.h "prover stage 3"): sums over BN254 G1 (pi_a, pib1, pi_c, the H sum)
// and G2 (pi_b) with resident bases, E emails per launch series (zkwg_msm_core.h), and the fixed-base multiples that turn a key with a
// known trapdoor into bases.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/zkwg.h"
#include "zkwg_msm_core.h"
#include "zkwg_fr_inv.h"

void zk_msm_launch_g1(const ZkMsmArgsT<ZkEcG1>& A, hipStream_t st);                          // zkwg_kernels_msm.hip
void zk_msm_launch_g2(const ZkMsmArgsT<ZkEcG2>& A, hipStream_t st);
void zk_msm_classify_launch(const ZkClassifyArgs& A, hipStream_t st);
void zk_msm_table_launch(int group, const void* bases, void* ext, u32* inf_bits, u32 n, u32 c, u32 K, hipStream_t st);
void zk_fixed_base_g1_launch(const G1Affine& gen, const Fr* k, G1Affine* out, u32 n, hipStream_t st);
void zk_fixed_base_g2_launch(const G2Affine& gen, const Fr* k, G2Affine* out, u32 n, hipStream_t st);

struct zkwg_msm {
  int device;
  int group;        // 1: G1 (64-byte bases), 2: G2 (128-byte bases)
  u64 n;
  u32 c, K, nb, s0;
  bool precomp;     // the table holds K shifted copies (one bucket set); false: the bases only (K bucket sets, Horner at the end)
  void* d_table;    // 2^261-Montgomery form (zkwg_ec29.h)
  u32* d_inf;       // one bit per base: the point at infinity
};
static inline u64 al256(u64 x) { return (x + 255) & ~255ull; }
// bytes of an accumulator in the kernels' form per point: 4 coordinates x 9 limbs (x 2 halves for G2)
static inline u64 msm_xs(const zkwg_msm* p) { return p->group == 2 ? 288 : 144; }
static ZkMsmOff msm_off(const zkwg_msm* p) {
  ZkMsmOff W;
  const u64 xs = msm_xs(p);
  const u64 sets = p->precomp ? 1 : p->K;
  const u64 total = sets * p->nb;
  u64 off = 0;
  W.count = off; off += al256((total + 1) * 4);
  W.cursor = off; off += al256(total * 4);
  W.entry = off; off += al256(p->n * p->K * 4);
  W.bucket = off; off += al256(total * xs);
  // the bit planes' two regions (zkwg_msm_core.h: rows = sets * c, n0 = ceil(nb / PFAN))
  const u64 n0 = zk_msm_plane_n0(p->nb), n1 = (n0 + ZK_MSM_PFAN - 1) / ZK_MSM_PFAN;
  W.node_s = off; off += al256(sets * p->c * n0 * xs);
  W.node_a = off; off += al256(sets * p->c * n1 * xs);
  W.window = off; off += al256(sets * xs);
  W.ones = off; off += al256(2 * ((p->n + ZK_MSM_ONES - 1) / ZK_MSM_ONES) * xs);
  u64 items = p->n * p->K;                       // entries: at most one per (scalar, window)
  for (int l = 0; l < 3; ++l) {
    const u64 cap = items / (l == 0 ? p->s0 : ZK_MSM_S1) + total + 1;     // sum_b ceil(len_b / S) <= items / S + buckets
    W.cap[l] = (u32)cap;
    W.soff[l] = off; off += al256((total + 1) * 4);
    W.part[l] = off; off += al256(cap * xs);
    items = cap;
  }
  W.total = al256(off);
  return W;
}
// the index lists of zk_msm_classify for E emails over this plan's bases: counters (n_sel[E] | n_ones[E]), then sel[E][n], ones[E][n]
struct ZkMsmLists { u64 counters, sel, ones, total; };
static ZkMsmLists msm_lists(const zkwg_msm* p, u64 E) {
  ZkMsmLists L;
  L.counters = 0;
  L.sel = al256(2 * E * 4);
  L.ones = L.sel + al256(E * p->n * 4);
  L.total = L.ones + al256(E * p->n * 4);
  return L;
}
static int msm_new(int device, int group, const void* bases, bool on_device, uint64_t n, int window_bits, int slice0, uint64_t table_budget, zkwg_msm_t** out) {
  if (!out || !bases || n == 0 || n >= (1ull << 31) || window_bits < 0 || window_bits > 20 || window_bits == 1 || slice0 < 0 || slice0 > 1024) return ZKWG_RC_BAD_ARG;
  if (device < 0) return ZKWG_RC_NO_DEVICE;
  zkwg_msm* p = new zkwg_msm();
  p->device = device; p->group = group; p->n = n; p->d_table = nullptr; p->d_inf = nullptr;
  // window: the bucket work (2^(c-1) buckets, c / 2 additions each in the bit planes) against K * n mixed additions
  p->c = window_bits ? (u32)window_bits : (n >= (1u << 20) ? 16u : n >= (1u << 16) ? 13u : n >= (1u << 12) ? 10u : n >= 256 ? 7u : 4u);
  p->K = zk_msm_windows(p->c); p->nb = 1u << (p->c - 1);
  p->s0 = slice0 ? (u32)slice0 : 16u;
  const size_t pt = group == 2 ? sizeof(G2Affine) : sizeof(G1Affine), bytes = n * pt;
  if (hipSetDevice(device) != hipSuccess) { delete p; return ZKWG_RC_HIP_ERROR; }
  void* d_src = (void*)bases;
  if (!on_device) {
    if (hipMalloc(&d_src, bytes) != hipSuccess) { delete p; (void)hipGetLastError(); return ZKWG_RC_OOM; }
    if (hipMemcpy(d_src, bases, bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d_src); delete p; return ZKWG_RC_HIP_ERROR; }
  }
  // Precomputed windows (default; ZKWG_MSM_PRECOMP=0 or a table_budget below K x the bases keeps the classic K bucket sets): K x the
  // bases' memory buys the Horner pass and K - 1 of the K bucket reductions.  table_budget = 0: whatever is free minus 4 GiB.
  bool pre = !(getenv("ZKWG_MSM_PRECOMP") && atoi(getenv("ZKWG_MSM_PRECOMP")) == 0) && (u64)p->K * n < (1ull << 31);
  if (pre) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { free_b = 0; (void)hipGetLastError(); }
    const u64 budget = table_budget ? table_budget : (free_b > (4ull << 30) ? free_b - (4ull << 30) : 0);
    if ((u64)bytes * p->K > budget) pre = false;
  }
  int rc = ZKWG_RC_OK;
  if (hipMalloc((void**)&p->d_inf, ((n + 31) / 32 + 1) * 4) != hipSuccess) rc = ZKWG_RC_OOM;
  if (rc == ZKWG_RC_OK && pre && hipMalloc(&p->d_table, bytes * p->K) != hipSuccess) { pre = false; p->d_table = nullptr; (void)hipGetLastError(); }
  if (rc == ZKWG_RC_OK && !pre && hipMalloc(&p->d_table, bytes) != hipSuccess) rc = ZKWG_RC_OOM;
  if (rc == ZKWG_RC_OK) {
    p->precomp = pre;
    zk_msm_table_launch(group, d_src, p->d_table, p->d_inf, (u32)n, p->c, pre ? p->K : 1u, 0);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = ZKWG_RC_HIP_ERROR;
  }
  if (!on_device) hipFree(d_src);
  if (rc != ZKWG_RC_OK) { (void)hipGetLastError(); if (p->d_table) hipFree(p->d_table); if (p->d_inf) hipFree(p->d_inf); delete p; return rc; }
  *out = p;
  return ZKWG_RC_OK;
}
template <class C>
static void msm_args(const zkwg_msm* p, const void* d_scalars, u64 scalar_stride_bytes, u32 E, int mont, void* d_work, const void* d_lists, bool with_ones,
                     void* d_out, ZkMsmArgsT<C>& A) {
  A.table = (const typename C::Affine*)p->d_table; A.inf = p->d_inf;
  A.scalars = (const Fr*)d_scalars; A.scalar_stride = scalar_stride_bytes / 32;
  A.n = (u32)p->n; A.c = p->c; A.K = p->K; A.nb = p->nb; A.KS = p->precomp ? 1u : p->K; A.stride = p->precomp ? (u32)p->n : 0u;
  A.E = E; A.scalars_mont = mont ? 1u : 0u; A.s0 = p->s0;
  { static const u32 ls = getenv("ZKWG_MSM_LDS_SORT") ? (u32)atoi(getenv("ZKWG_MSM_LDS_SORT")) : 1u; A.lds_sort = ls; }
  A.off = msm_off(p);
  A.work = (u8*)d_work; A.work_stride = A.off.total;
  A.sel = A.n_sel = A.ones = A.n_ones = nullptr; A.list_stride = p->n;
  if (d_lists) {
    const ZkMsmLists L = msm_lists(p, E);
    const u8* l = (const u8*)d_lists;
    A.n_sel = (const u32*)(l + L.counters); A.sel = (const u32*)(l + L.sel);
    if (with_ones) { A.n_ones = A.n_sel + E; A.ones = (const u32*)(l + L.ones); }
  }
  A.out = (typename C::Out*)d_out;
}
static void msm_launch(const zkwg_msm* p, const void* d_scalars, u64 stride_bytes, u32 E, int mont, void* d_work, const void* d_lists, bool with_ones, void* d_out,
                       hipStream_t st) {
  if (p->group == 1) { ZkMsmArgsT<ZkEcG1> A; msm_args<ZkEcG1>(p, d_scalars, stride_bytes, E, mont, d_work, d_lists, with_ones, d_out, A); zk_msm_launch_g1(A, st); }
  else { ZkMsmArgsT<ZkEcG2> A; msm_args<ZkEcG2>(p, d_scalars, stride_bytes, E, mont, d_work, d_lists, with_ones, d_out, A); zk_msm_launch_g2(A, st); }
}
static void classify_target(const zkwg_msm* p, u32 first, u32 E, void* d_lists, ZkClassifyTarget& T) {
  const ZkMsmLists L = msm_lists(p, E);
  u8* l = (u8*)d_lists;
  T.inf = p->d_inf; T.first = first; T.n = (u32)p->n;
  T.n_sel = (u32*)(l + L.counters); T.n_ones = T.n_sel + E;
  T.sel = (u32*)(l + L.sel); T.ones = (u32*)(l + L.ones);
  T.list_stride = p->n;
}

extern "C" {

int zkwg_msm_create(int device, const uint8_t* bases, uint64_t n, int window_bits, zkwg_msm_t** out) { return msm_new(device, 1, bases, false, n, window_bits, 0, 0, out); }
int zkwg_msm_create_g2(int device, const uint8_t* bases, uint64_t n, int window_bits, zkwg_msm_t** out) { return msm_new(device, 2, bases, false, n, window_bits, 0, 0, out); }
int zkwg_msm_create_device(int device, int group, const void* d_bases, uint64_t n, int window_bits, zkwg_msm_t** out) {
  if (group != 1 && group != 2) return ZKWG_RC_BAD_ARG;
  return msm_new(device, group, d_bases, true, n, window_bits, 0, 0, out);
}
int zkwg_msm_create_ex(int device, int group, const void* bases, int bases_on_device, uint64_t n, int window_bits, int slice0, uint64_t table_budget,
                       zkwg_msm_t** out) {
  if (group != 1 && group != 2) return ZKWG_RC_BAD_ARG;
  return msm_new(device, group, bases, bases_on_device != 0, n, window_bits, slice0, table_budget, out);
}
void zkwg_msm_destroy(zkwg_msm_t* p) {
  if (!p) return;
  if (p->device >= 0) { hipSetDevice(p->device); if (p->d_table) hipFree(p->d_table); if (p->d_inf) hipFree(p->d_inf); }
  delete p;
}
// scratch of one email's sum / of E emails' sums (the generic entry points keep their own index lists behind the sums' arrays)
uint64_t zkwg_msm_work_bytes_batch(const zkwg_msm_t* p, uint64_t n_emails) { return p && n_emails ? n_emails * msm_off(p).total + msm_lists(p, n_emails).total : 0; }
uint64_t zkwg_msm_work_bytes(const zkwg_msm_t* p) { return zkwg_msm_work_bytes_batch(p, 1); }
// the same figures before a plan exists (what a prover sets aside before it sizes its tables): per email, precomputed-windows layout
uint64_t zkwg_msm_estimate_work_bytes(int group, uint64_t n, int window_bits, int slice0) {
  if ((group != 1 && group != 2) || n == 0 || n >= (1ull << 31)) return 0;
  zkwg_msm t;
  t.group = group; t.n = n; t.precomp = true;
  t.c = window_bits ? (u32)window_bits : (n >= (1u << 20) ? 16u : n >= (1u << 16) ? 13u : n >= (1u << 12) ? 10u : n >= 256 ? 7u : 4u);
  t.K = zk_msm_windows(t.c); t.nb = 1u << (t.c - 1); t.s0 = slice0 ? (u32)slice0 : 16u;
  return msm_off(&t).total + msm_lists(&t, 1).total;
}
uint64_t zkwg_msm_lists_bytes(const zkwg_msm_t* p, uint64_t n_emails) { return p && n_emails ? msm_lists(p, n_emails).total : 0; }
uint64_t zkwg_msm_table_bytes(const zkwg_msm_t* p) { return p ? p->n * (p->group == 2 ? 128ull : 64ull) * (p->precomp ? p->K : 1u) : 0; }
int zkwg_msm_window_bits(const zkwg_msm_t* p) { return p ? (int)p->c : 0; }
int zkwg_msm_group(const zkwg_msm_t* p) { return p ? p->group : 0; }
int zkwg_msm_precomputed(const zkwg_msm_t* p) { return p && p->precomp ? 1 : 0; }

// One pass over E scalar vectors (n_scalars each, scalar_stride bytes apart) for up to three plans at once: plan t covers the scalars
// [first[t], first[t] + its n) and gets, in d_lists[t] (zkwg_msm_lists_bytes(plan, E) bytes), per email the list of its scalars that are 1
// (ones_apart; their bases are summed without buckets) and the list of the others that are not 0 -- bases at infinity left out.
int zkwg_msm_classify_device(zkwg_msm_t* const* plans, const uint64_t* first, uint32_t n_plans, const void* d_scalars, uint64_t scalar_stride, uint64_t n_scalars,
                             uint64_t n_emails, int scalars_montgomery, int ones_apart, void* const* d_lists, void* hip_stream) {
  if (!plans || !first || n_plans == 0 || n_plans > 3 || !d_scalars || !d_lists || n_emails == 0 || n_emails > 65535 || n_scalars == 0 || n_scalars >= (1ull << 31) ||
      (scalar_stride & 31) || ((uintptr_t)d_scalars & 15))
    return ZKWG_RC_BAD_ARG;
  ZkClassifyArgs A;
  A.scalars = (const Fr*)d_scalars; A.scalar_stride = scalar_stride / 32; A.n = (u32)n_scalars; A.E = (u32)n_emails; A.scalars_mont = scalars_montgomery ? 1u : 0u;
  A.ones_apart = ones_apart ? 1u : 0u; A.n_targets = n_plans;
  hipStream_t st = (hipStream_t)hip_stream;
  for (uint32_t t = 0; t < n_plans; ++t) {
    if (!plans[t] || !d_lists[t] || ((uintptr_t)d_lists[t] & 255) || first[t] + plans[t]->n > n_scalars || plans[t]->device != plans[0]->device) return ZKWG_RC_BAD_ARG;
    classify_target(plans[t], (u32)first[t], (u32)n_emails, d_lists[t], A.t[t]);
  }
  if (hipSetDevice(plans[0]->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  for (uint32_t t = 0; t < n_plans; ++t) hipMemsetAsync(A.t[t].n_sel, 0, 2 * n_emails * 4, st);
  zk_msm_classify_launch(A, st);
  return hipGetLastError() == hipSuccess ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}
// E sums over the plan's bases with the lists of zkwg_msm_classify_device: the accumulators (XYZZ words, 2^256-Montgomery form: 128 bytes
// per email for G1, 256 for G2) are left at d_out_xyzz, nothing is synchronised.  d_work: n_emails * zkwg_msm_work_bytes(plan) bytes.
int zkwg_msm_enqueue_lists_device(zkwg_msm_t* p, const void* d_scalars, uint64_t scalar_stride, uint64_t n_emails, int scalars_montgomery, const void* d_lists,
                                  int with_ones, void* d_work, void* d_out_xyzz, void* hip_stream) {
  if (!p || !d_scalars || !d_lists || !d_work || !d_out_xyzz || n_emails == 0 || n_emails > 65535 || (scalar_stride & 31) || ((uintptr_t)d_work & 255) ||
      ((uintptr_t)d_lists & 255) || ((uintptr_t)d_scalars & 15) || ((uintptr_t)d_out_xyzz & 15))
    return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  msm_launch(p, d_scalars, scalar_stride, (u32)n_emails, scalars_montgomery, d_work, d_lists, with_ones != 0, d_out_xyzz, (hipStream_t)hip_stream);
  return hipGetLastError() == hipSuccess ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}
// E sums of arbitrary scalars: classification (when ones_apart) into lists kept behind the sums' arrays, then the series.
// d_work: zkwg_msm_work_bytes_batch(plan, n_emails) bytes.
int zkwg_msm_enqueue_batch_device(zkwg_msm_t* p, const void* d_scalars, uint64_t scalar_stride, uint64_t n_emails, int scalars_montgomery, int ones_apart,
                                  void* d_work, void* d_out_xyzz, void* hip_stream) {
  if (!p || !d_scalars || !d_work || !d_out_xyzz || n_emails == 0 || n_emails > 65535 || (scalar_stride & 31) || ((uintptr_t)d_work & 255) || ((uintptr_t)d_scalars & 15) ||
      ((uintptr_t)d_out_xyzz & 15))
    return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  hipStream_t st = (hipStream_t)hip_stream;
  void* lists = nullptr;
  if (ones_apart) {
    lists = (u8*)d_work + n_emails * msm_off(p).total;
    zkwg_msm_t* plans[1] = {p};
    const uint64_t first[1] = {0};
    void* ls[1] = {lists};
    const int rc = zkwg_msm_classify_device(plans, first, 1, d_scalars, scalar_stride, p->n, n_emails, scalars_montgomery, 1, ls, hip_stream);
    if (rc != ZKWG_RC_OK) return rc;
  }
  msm_launch(p, d_scalars, scalar_stride, (u32)n_emails, scalars_montgomery, d_work, lists, ones_apart != 0, d_out_xyzz, st);
  return hipGetLastError() == hipSuccess ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}
int zkwg_msm_enqueue_device(zkwg_msm_t* p, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, void* d_out_xyzz, void* hip_stream) {
  return zkwg_msm_enqueue_batch_device(p, d_scalars, p ? p->n * 32 : 0, 1, scalars_montgomery, ones_apart, d_work, d_out_xyzz, hip_stream);
}
// one sum, synchronous, as the zkey would store the point (one inversion, on the host).  The accumulator is parked at the end of the
// email's window area (a spare slot: the layout rounds every array up to 256 bytes and the window of a precomputed plan is one point)
static int msm_one(zkwg_msm_t* p, int group, const void* d_scalars, int mont, int ones_apart, void* d_work, uint8_t* out_xy, void* hip_stream) {
  if (!p || p->group != group || !d_scalars || !d_work || !out_xy) return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  hipStream_t st = (hipStream_t)hip_stream;
  void* d_out = nullptr;
  if (hipMalloc(&d_out, 256) != hipSuccess) { (void)hipGetLastError(); return ZKWG_RC_OOM; }
  int rc = zkwg_msm_enqueue_device(p, d_scalars, mont, ones_apart, d_work, d_out, hip_stream);
  uint8_t raw[256];
  if (rc == ZKWG_RC_OK && (hipMemcpyAsync(raw, d_out, 256, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) rc = ZKWG_RC_HIP_ERROR;
  if (rc == ZKWG_RC_OK && hipGetLastError() != hipSuccess) rc = ZKWG_RC_HIP_ERROR;
  hipFree(d_out);
  if (rc != ZKWG_RC_OK) return rc;
  return zkwg_msm_finish_host(group, raw, 1, out_xy);
}
int zkwg_msm_g1_device(zkwg_msm_t* p, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, uint8_t* out_xy, void* hip_stream) {
  return msm_one(p, 1, d_scalars, scalars_montgomery, ones_apart, d_work, out_xy, hip_stream);
}
int zkwg_msm_g2_device(zkwg_msm_t* p, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, uint8_t* out_xy, void* hip_stream) {
  return msm_one(p, 2, d_scalars, scalars_montgomery, ones_apart, d_work, out_xy, hip_stream);
}

int zkwg_msm_finish_host(int group, const uint8_t* xyzz, uint64_t n, uint8_t* out_points) {
  if (!xyzz || !out_points || (group != 1 && group != 2)) return ZKWG_RC_BAD_ARG;
  for (uint64_t i = 0; i < n; ++i) {
    if (group == 1) { G1Xyzz r; memcpy(&r, xyzz + i * sizeof(r), sizeof(r)); const G1Affine a = g1_to_affine(r); memcpy(out_points + 64 * i, &a, 64); }
    else { G2Xyzz r; memcpy(&r, xyzz + i * sizeof(r), sizeof(r)); const G2Affine a = g2_to_affine(r); memcpy(out_points + 128 * i, &a, 128); }
  }
  return ZKWG_RC_OK;
}

// d_out[i] = k_i G (group 1: the generator (1, 2), 64-byte affine Montgomery points; group 2: the EIP-197 generator, 128 bytes)
// for n standard-form scalars at d_scalars: how the tests turn a key with a known trapdoor (oracle/pyref/groth16.py) into bases.
int zkwg_fixed_base_device(int device, int group, const void* d_scalars, uint64_t n, void* d_out, void* hip_stream) {
  if (!d_scalars || !d_out || n == 0 || n >= (1ull << 31) || (group != 1 && group != 2)) return ZKWG_RC_BAD_ARG;
  if (device < 0) return ZKWG_RC_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  hipStream_t st = (hipStream_t)hip_stream;
  if (group == 1) {
    const G1Affine g{fq_to_mont(Fq{{1, 0, 0, 0}}), fq_to_mont(Fq{{2, 0, 0, 0}})};
    zk_fixed_base_g1_launch(g, (const Fr*)d_scalars, (G1Affine*)d_out, (u32)n, st);
  } else {
    // EIP-197 generator of G2 (oracle/pyref/bn254_g2.py), standard form -> Montgomery
    const Fq x0{{0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL}};
    const Fq x1{{0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL}};
    const Fq y0{{0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL}};
    const Fq y1{{0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL}};
    const G2Affine g{Fq2{fq_to_mont(x0), fq_to_mont(x1)}, Fq2{fq_to_mont(y0), fq_to_mont(y1)}};
    if (!g2_on_curve(g)) return ZKWG_RC_BAD_CONFIG;
    zk_fixed_base_g2_launch(g, (const Fr*)d_scalars, (G2Affine*)d_out, (u32)n, st);
  }
  return hipGetLastError() == hipSuccess ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}

// pi_a, pi_b, pi_c from the five sums (groth16_prove.js):
//     pi_a = alpha1 + sum_a + r delta1        pi_b = beta2 + sum_b2 + s delta2        pib1 = beta1 + sum_b1 + s delta1
//     pi_c = sum_c + sum_h + s pi_a + r pib1 - (r s) delta1
// Points in: affine Montgomery form as the zkey stores them (64 / 128 bytes, zeros = infinity); r, s: 32-byte little-endian
// standard-form scalars (the prover's blinding; any value below the group order).  Points out: affine STANDARD form, little-endian
// x | y (64 bytes; G2: x.c0 | x.c1 | y.c0 | y.c1, 128 bytes) -- the integers snarkjs prints into proof.json.  A dozen group
// operations: host arithmetic (zkwg_g1.h / zkwg_g2.h, the functions the kernels run).
int zkwg_groth16_assemble(const uint8_t* sum_a, const uint8_t* sum_b1, const uint8_t* sum_b2, const uint8_t* sum_c, const uint8_t* sum_h,
                          const uint8_t* vk_alpha1, const uint8_t* vk_beta1, const uint8_t* vk_beta2, const uint8_t* vk_delta1, const uint8_t* vk_delta2,
                          const uint8_t* r32, const uint8_t* s32, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c) {
  if (!sum_a || !sum_b1 || !sum_b2 || !sum_c || !sum_h || !vk_alpha1 || !vk_beta1 || !vk_beta2 || !vk_delta1 || !vk_delta2 || !r32 || !s32 || !pi_a || !pi_b || !pi_c)
    return ZKWG_RC_BAD_ARG;
  auto g1 = [](const uint8_t* p) { G1Affine a; memcpy(&a, p, 64); return a; };
  auto g2 = [](const uint8_t* p) { G2Affine a; memcpy(&a, p, 128); return a; };
  Fr r, s;
  memcpy(&r, r32, 32); memcpy(&s, s32, 32);
  if (fr_geq(r, fr_p()) || fr_geq(s, fr_p())) return ZKWG_RC_BAD_ARG;
  for (const uint8_t* p : {sum_a, sum_b1, sum_c, sum_h, vk_alpha1, vk_beta1, vk_delta1}) if (!g1_on_curve(g1(p))) return ZKWG_RC_BAD_ARG;
  for (const uint8_t* p : {sum_b2, vk_beta2, vk_delta2}) if (!g2_on_curve(g2(p))) return ZKWG_RC_BAD_ARG;
  auto mul1 = [](const Fr& k, const G1Xyzz& p) {
    G1Xyzz acc = g1_xyzz_inf();
    for (int i = 255; i >= 0; --i) { acc = g1_dbl(acc); if ((k.l[i >> 6] >> (i & 63)) & 1) acc = g1_add(acc, p); }
    return acc;
  };
  auto mul2 = [](const Fr& k, const G2Xyzz& p) {
    G2Xyzz acc = g2_xyzz_inf();
    for (int i = 255; i >= 0; --i) { acc = g2_dbl(acc); if ((k.l[i >> 6] >> (i & 63)) & 1) acc = g2_add(acc, p); }
    return acc;
  };
  auto x1 = [](const G1Affine& a) { return g1_from_affine(a); };
  auto x2 = [](const G2Affine& a) { return g2_is_inf(a) ? g2_xyzz_inf() : G2Xyzz{a.x, a.y, fq2_one(), fq2_one()}; };
  const G1Xyzz d1 = x1(g1(vk_delta1));
  const G1Xyzz A = g1_add(g1_add(x1(g1(vk_alpha1)), x1(g1(sum_a))), mul1(r, d1));
  const G2Xyzz B = g2_add(g2_add(x2(g2(vk_beta2)), x2(g2(sum_b2))), mul2(s, x2(g2(vk_delta2))));
  const G1Xyzz B1 = g1_add(g1_add(x1(g1(vk_beta1)), x1(g1(sum_b1))), mul1(s, d1));
  const Fr rs = fr_mont_mul(fr_mont_mul(r, s), fr_R2());   // standard-form product
  const G1Affine rsd = g1_to_affine(mul1(rs, d1));
  G1Xyzz Cc = g1_add(x1(g1(sum_c)), x1(g1(sum_h)));
  Cc = g1_add(Cc, mul1(s, A));
  Cc = g1_add(Cc, mul1(r, B1));
  Cc = g1_add_mixed(Cc, g1_neg(rsd));
  auto out1 = [](const G1Xyzz& p, uint8_t* o) {
    const G1Affine a = g1_to_affine(p);
    const Fq x = g1_is_inf(a) ? a.x : fq_from_mont(a.x), y = g1_is_inf(a) ? a.y : fq_from_mont(a.y);
    memcpy(o, &x, 32); memcpy(o + 32, &y, 32);
  };
  out1(A, pi_a); out1(Cc, pi_c);
  {
    const G2Affine a = g2_to_affine(B);
    const bool inf = g2_is_inf(a);
    const Fq v[4] = {inf ? a.x.c0 : fq_from_mont(a.x.c0), inf ? a.x.c1 : fq_from_mont(a.x.c1), inf ? a.y.c0 : fq_from_mont(a.y.c0), inf ? a.y.c1 : fq_from_mont(a.y.c1)};
    memcpy(pi_b, v, 128);
  }
  return ZKWG_RC_OK;
}

}
