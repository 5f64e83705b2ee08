// C-ABI of the multi-exponentiations of the prover (include/zkwg.h "prover stage 3"): sums over BN254 G1 (pi_a, pib1, pi_c, the H sum)
// and G2 (pi_b) with resident bases, and the fixed-base multiples that turn a key with a known trapdoor into bases.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/zkwg.h"
#include "zkwg_msm_core.h"
#include "zkwg_fr_inv.h"

void zk_msm_launch(const ZkMsmArgs& A, hipStream_t st);                          // zkwg_kernels_msm.hip
void zk_msm_launch_g2(const ZkMsmArgsT<ZkCurveG2>& A, hipStream_t st);
void zk_msm_shift_launch(int group, const void* bases, void* ext, u32 n, u32 c, u32 K, hipStream_t st);
void zk_fixed_base_g1_launch(const G1Affine& gen, const Fr* k, G1Affine* out, u32 n, hipStream_t st);
void zk_fixed_base_g2_launch(const G2Affine& gen, const Fr* k, G2Affine* out, u32 n, hipStream_t st);

struct zkwg_msm {
  int device;
  int group;        // 1: G1 (64-byte bases), 2: G2 (128-byte bases)
  u64 n;
  u32 c, K, nb;
  void* d_bases;
  bool owns;        // false: the bases are the caller's device memory (zkwg_msm_create_device)
  void* d_ext;      // precomputed windows: K copies of the bases, copy w = 2^(c w) * base (NULL: classic layout, ZKWG_MSM_PRECOMP=0)
};
struct ZkMsmWork { u64 count, cursor, entry, bucket, node_s, node_a, window, out, ones, soff[3], part[3], total; u32 cap[3]; };
static ZkMsmWork msm_work(const zkwg_msm* p) {
  ZkMsmWork W;
  auto al = [](u64 x) { return (x + 255) & ~255ull; };
  const u64 xs = p->group == 2 ? sizeof(G2Xyzz) : sizeof(G1Xyzz);
  const u64 sets = p->d_ext ? 1 : p->K;
  const u64 total = sets * p->nb, half = sets * ((p->nb + ZK_MSM_FAN - 1) / ZK_MSM_FAN);
  u64 off = 0;
  W.count = off; off += al((total + 1) * 4);
  W.cursor = off; off += al(total * 4);
  W.entry = off; off += al(p->n * p->K * 4);
  W.bucket = off; off += al(total * xs);
  // the larger of the tree's ping-pong halves and the bit planes' two regions (zkwg_msm_core.h: rows = sets * c, n0 = ceil(nb / PFAN))
  const u64 n0 = zk_msm_plane_n0(p->nb), n1 = (n0 + ZK_MSM_PFAN - 1) / ZK_MSM_PFAN;
  W.node_s = off; off += al(std::max<u64>(2 * half, sets * p->c * n0) * xs);
  W.node_a = off; off += al(std::max<u64>(2 * half, sets * p->c * n1) * xs);
  W.window = off; off += al(sets * xs);
  W.out = off; off += al(xs);
  W.ones = off; off += al(2 * ((p->n + ZK_MSM_ONES - 1) / ZK_MSM_ONES) * xs);
  u64 items = p->n * p->K;                       // entries: at most one per (scalar, window)
  for (int l = 0; l < 3; ++l) {
    const u64 cap = items / zk_msm_slice_size(l) + total + 1;     // sum_b ceil(len_b / S) <= items / S + buckets
    W.cap[l] = (u32)cap;
    W.soff[l] = off; off += al((total + 1) * 4);
    W.part[l] = off; off += al(cap * xs);
    items = cap;
  }
  W.total = off;
  return W;
}
static int msm_new(int device, int group, const void* bases, bool on_device, uint64_t n, int window_bits, zkwg_msm_t** out) {
  if (!out || !bases || n == 0 || n >= (1ull << 31) || window_bits < 0 || window_bits > 20 || window_bits == 1) return ZKWG_RC_BAD_ARG;
  if (device < 0) return ZKWG_RC_NO_DEVICE;
  zkwg_msm* p = new zkwg_msm();
  p->device = device; p->group = group; p->n = n; p->d_bases = nullptr; p->owns = !on_device;
  // window: the bucket work (K * 2^(c-1) buckets, three additions each in the tree) against K * n mixed additions
  p->c = window_bits ? (u32)window_bits : (n >= (1u << 20) ? 16u : n >= (1u << 16) ? 13u : n >= (1u << 12) ? 10u : n >= 256 ? 7u : 4u);
  p->K = zk_msm_windows(p->c); p->nb = 1u << (p->c - 1);
  const size_t bytes = n * (group == 2 ? sizeof(G2Affine) : sizeof(G1Affine));
  if (hipSetDevice(device) != hipSuccess) { delete p; return ZKWG_RC_HIP_ERROR; }
  if (on_device) p->d_bases = (void*)bases;
  else {
    if (hipMalloc(&p->d_bases, bytes) != hipSuccess) { delete p; return ZKWG_RC_OOM; }
    if (hipMemcpy(p->d_bases, bases, bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(p->d_bases); delete p; return ZKWG_RC_HIP_ERROR; }
  }
  // precomputed windows (default; ZKWG_MSM_PRECOMP=0 keeps the classic K bucket sets): K x the bases' memory buys the Horner pass
  p->d_ext = nullptr;
  const bool pre = !(getenv("ZKWG_MSM_PRECOMP") && atoi(getenv("ZKWG_MSM_PRECOMP")) == 0) && (u64)p->K * n < (1ull << 31);
  if (pre) {
    size_t free_b = 0, total_b = 0;
    hipMemGetInfo(&free_b, &total_b);
    if (bytes * p->K + (4ull << 30) < free_b && hipMalloc(&p->d_ext, bytes * p->K) == hipSuccess) {
      zk_msm_shift_launch(group, p->d_bases, p->d_ext, (u32)n, p->c, p->K, 0);
      if (hipDeviceSynchronize() != hipSuccess) { hipFree(p->d_ext); p->d_ext = nullptr; (void)hipGetLastError(); }
    } else (void)hipGetLastError();
  }
  *out = p;
  return ZKWG_RC_OK;
}
template <class C>
static void msm_args(const zkwg_msm* p, const void* d_scalars, int mont, int ones_apart, void* d_work, ZkMsmArgsT<C>& A) {
  typedef typename C::Xyzz X;
  const ZkMsmWork W = msm_work(p);
  u8* w = (u8*)d_work;
  A.bases = (const typename C::Affine*)(p->d_ext ? p->d_ext : p->d_bases); A.KS = p->d_ext ? 1u : p->K; A.stride = p->d_ext ? (u32)p->n : 0u; A.scalars = (const Fr*)d_scalars; A.n = (u32)p->n; A.c = p->c; A.K = p->K; A.nb = p->nb;
  A.scalars_mont = mont ? 1u : 0u; A.ones_apart = ones_apart ? 1u : 0u; A.ones = (X*)(w + W.ones);
  { static const u32 ps = getenv("ZKWG_MSM_PLANES") ? (u32)atoi(getenv("ZKWG_MSM_PLANES")) : 1u; A.plane_sums = ps; }
  { static const u32 ls = getenv("ZKWG_MSM_LDS_SORT") ? (u32)atoi(getenv("ZKWG_MSM_LDS_SORT")) : 1u; A.lds_sort = ls; }
  A.count = (u32*)(w + W.count); A.cursor = (u32*)(w + W.cursor); A.entry = (u32*)(w + W.entry); A.bucket = (X*)(w + W.bucket);
  A.node_s = (X*)(w + W.node_s); A.node_a = (X*)(w + W.node_a); A.window = (X*)(w + W.window); A.out = (X*)(w + W.out);
  for (int l = 0; l < 3; ++l) { A.soff[l] = (u32*)(w + W.soff[l]); A.part[l] = (X*)(w + W.part[l]); A.cap[l] = W.cap[l]; }
}

extern "C" {

int zkwg_msm_create(int device, const uint8_t* bases, uint64_t n, int window_bits, zkwg_msm_t** out) { return msm_new(device, 1, bases, false, n, window_bits, out); }
int zkwg_msm_create_g2(int device, const uint8_t* bases, uint64_t n, int window_bits, zkwg_msm_t** out) { return msm_new(device, 2, bases, false, n, window_bits, out); }
int zkwg_msm_create_device(int device, int group, const void* d_bases, uint64_t n, int window_bits, zkwg_msm_t** out) {
  if (group != 1 && group != 2) return ZKWG_RC_BAD_ARG;
  return msm_new(device, group, d_bases, true, n, window_bits, out);
}
void zkwg_msm_destroy(zkwg_msm_t* p) {
  if (!p) return;
  if (p->device >= 0) { hipSetDevice(p->device); if (p->owns) hipFree(p->d_bases); if (p->d_ext) hipFree(p->d_ext); }
  delete p;
}
uint64_t zkwg_msm_work_bytes(const zkwg_msm_t* p) { return p ? msm_work(p).total : 0; }
int zkwg_msm_window_bits(const zkwg_msm_t* p) { return p ? (int)p->c : 0; }
int zkwg_msm_group(const zkwg_msm_t* p) { return p ? p->group : 0; }

// the sum as a point in XYZZ coordinates left on the device is not exposed: the result is small, the caller wants it on the host
int zkwg_msm_g1_device(zkwg_msm_t* p, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, uint8_t* out_xy, void* hip_stream) {
  if (!p || p->group != 1 || !d_scalars || !d_work || !out_xy || ((uintptr_t)d_work & 255) || ((uintptr_t)d_scalars & 15)) return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  ZkMsmArgs A;
  msm_args<ZkCurveG1>(p, d_scalars, scalars_montgomery, ones_apart, d_work, A);
  hipStream_t st = (hipStream_t)hip_stream;
  zk_msm_launch(A, st);
  G1Xyzz r;
  if (hipMemcpyAsync(&r, A.out, sizeof(r), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  if (hipGetLastError() != hipSuccess) return ZKWG_RC_HIP_ERROR;
  const G1Affine a = g1_to_affine(r);        // one inversion, on the host
  memcpy(out_xy, &a, 64);
  return ZKWG_RC_OK;
}
int zkwg_msm_g2_device(zkwg_msm_t* p, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, uint8_t* out_xy, void* hip_stream) {
  if (!p || p->group != 2 || !d_scalars || !d_work || !out_xy || ((uintptr_t)d_work & 255) || ((uintptr_t)d_scalars & 15)) return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  ZkMsmArgsT<ZkCurveG2> A;
  msm_args<ZkCurveG2>(p, d_scalars, scalars_montgomery, ones_apart, d_work, A);
  hipStream_t st = (hipStream_t)hip_stream;
  zk_msm_launch_g2(A, st);
  G2Xyzz r;
  if (hipMemcpyAsync(&r, A.out, sizeof(r), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  if (hipGetLastError() != hipSuccess) return ZKWG_RC_HIP_ERROR;
  const G2Affine a = g2_to_affine(r);
  memcpy(out_xy, &a, 128);
  return ZKWG_RC_OK;
}

// The same sums without a host round trip per sum: the accumulator (XYZZ coordinates, Montgomery form: 128 bytes for G1, 256 for G2)
// is left at d_out_xyzz and nothing is synchronised, so the sums of several proofs can be in flight on several streams -- a
// multi-exponentiation ends in a few hundred dependent group operations on a handful of lanes (the bucket tree, the Horner pass over
// the windows), which only other proofs' work can hide.  zkwg_msm_finish_host turns downloaded accumulators into the zkey's point form.
int zkwg_msm_enqueue_device(zkwg_msm_t* p, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, void* d_out_xyzz, void* hip_stream) {
  if (!p || !d_scalars || !d_work || !d_out_xyzz || ((uintptr_t)d_work & 255) || ((uintptr_t)d_scalars & 15) || ((uintptr_t)d_out_xyzz & 15)) return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  hipStream_t st = (hipStream_t)hip_stream;
  if (p->group == 1) {
    ZkMsmArgs A;
    msm_args<ZkCurveG1>(p, d_scalars, scalars_montgomery, ones_apart, d_work, A);
    A.out = (G1Xyzz*)d_out_xyzz;
    zk_msm_launch(A, st);
  } else {
    ZkMsmArgsT<ZkCurveG2> A;
    msm_args<ZkCurveG2>(p, d_scalars, scalars_montgomery, ones_apart, d_work, A);
    A.out = (G2Xyzz*)d_out_xyzz;
    zk_msm_launch_g2(A, st);
  }
  return hipGetLastError() == hipSuccess ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
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
