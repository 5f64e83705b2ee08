// C-ABI of the G1 multi-exponentiation (include/zkwg.h "prover stage 3").  DRAFT, branch next/msm: never run on a GPU yet.
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>
#include "../../include/zkwg.h"
#include "zkwg_msm_core.h"

void zk_msm_launch(const ZkMsmArgs& A, hipStream_t st);   // zkwg_kernels_msm.hip

struct zkwg_msm {
  int device;
  u64 n;
  u32 c, K, nb;
  G1Affine* d_bases;
};
struct ZkMsmWork { u64 count, cursor, entry, bucket, node_s, node_a, window, out, ones, total; };
static ZkMsmWork msm_work(const zkwg_msm* p) {
  ZkMsmWork W;
  auto al = [](u64 x) { return (x + 255) & ~255ull; };
  const u64 total = (u64)p->K * p->nb, half = (u64)p->K * ((p->nb + 31) / 32);
  u64 off = 0;
  W.count = off; off += al((total + 1) * 4);
  W.cursor = off; off += al(total * 4);
  W.entry = off; off += al(p->n * p->K * 4);
  W.bucket = off; off += al(total * sizeof(G1Xyzz));
  W.node_s = off; off += al(2 * half * sizeof(G1Xyzz));
  W.node_a = off; off += al(2 * half * sizeof(G1Xyzz));
  W.window = off; off += al((u64)p->K * sizeof(G1Xyzz));
  W.out = off; off += al(sizeof(G1Xyzz));
  W.ones = off; off += al(2 * ((p->n + 63) / 64) * sizeof(G1Xyzz));
  W.total = off;
  return W;
}

extern "C" {

int zkwg_msm_create(int device, const uint8_t* bases, uint64_t n, int window_bits, zkwg_msm_t** out) {
  if (!out || !bases || n == 0 || n >= (1ull << 31) || window_bits < 0 || window_bits > 20 || window_bits == 1) return ZKWG_RC_BAD_ARG;
  if (device < 0) return ZKWG_RC_NO_DEVICE;
  zkwg_msm* p = new zkwg_msm();
  p->device = device; p->n = n; p->d_bases = nullptr;
  // window: the bucket work (K * 2^(c-1) buckets, three additions each in the tree) against K * n mixed additions
  p->c = window_bits ? (u32)window_bits : (n >= (1u << 20) ? 16u : n >= (1u << 16) ? 13u : n >= (1u << 12) ? 10u : n >= 256 ? 7u : 4u);
  p->K = zk_msm_windows(p->c); p->nb = 1u << (p->c - 1);
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&p->d_bases, n * sizeof(G1Affine)) != hipSuccess) { delete p; return ZKWG_RC_OOM; }
  if (hipMemcpy(p->d_bases, bases, n * sizeof(G1Affine), hipMemcpyHostToDevice) != hipSuccess) { hipFree(p->d_bases); delete p; return ZKWG_RC_HIP_ERROR; }
  *out = p;
  return ZKWG_RC_OK;
}
void zkwg_msm_destroy(zkwg_msm_t* p) {
  if (!p) return;
  if (p->device >= 0) { hipSetDevice(p->device); hipFree(p->d_bases); }
  delete p;
}
uint64_t zkwg_msm_work_bytes(const zkwg_msm_t* p) { return p ? msm_work(p).total : 0; }
int zkwg_msm_window_bits(const zkwg_msm_t* p) { return p ? (int)p->c : 0; }

int zkwg_msm_g1_device(zkwg_msm_t* p, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, uint8_t* out_xy, void* hip_stream) {
  if (!p || !d_scalars || !d_work || !out_xy || ((uintptr_t)d_work & 255) || ((uintptr_t)d_scalars & 15)) return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  const ZkMsmWork W = msm_work(p);
  u8* w = (u8*)d_work;
  ZkMsmArgs A;
  A.bases = p->d_bases; A.scalars = (const Fr*)d_scalars; A.n = (u32)p->n; A.c = p->c; A.K = p->K; A.nb = p->nb; A.scalars_mont = scalars_montgomery ? 1u : 0u;
  A.ones_apart = ones_apart ? 1u : 0u; A.ones = (G1Xyzz*)(w + W.ones);
  A.count = (u32*)(w + W.count); A.cursor = (u32*)(w + W.cursor); A.entry = (u32*)(w + W.entry); A.bucket = (G1Xyzz*)(w + W.bucket);
  A.node_s = (G1Xyzz*)(w + W.node_s); A.node_a = (G1Xyzz*)(w + W.node_a); A.window = (G1Xyzz*)(w + W.window); A.out = (G1Xyzz*)(w + W.out);
  hipStream_t st = (hipStream_t)hip_stream;
  zk_msm_launch(A, st);
  G1Xyzz r;
  if (hipMemcpyAsync(&r, A.out, sizeof(r), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  if (hipGetLastError() != hipSuccess) return ZKWG_RC_HIP_ERROR;
  const G1Affine a = g1_to_affine(r);        // one inversion, on the host
  memcpy(out_xy, &a, 64);
  return ZKWG_RC_OK;
}

}
