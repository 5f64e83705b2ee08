// C-ABI of the transform stage (include/zkwg.h "prover stage 2"): plans (twiddle / coset tables built on the host once per
// domain size), stand-alone transforms, and the H-evaluation pipeline of groth16_prove.js over a batch of emails.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/zkwg.h"
#include "zkwg_fr.h"

extern "C" int zk_ntt_launch(int dit, const Fr* src, u64 src_es, u64 src_ps, u64 valid, int src_lazy, void* work, Fr* out, const Fr* tw, const Fr* scale,
                             const Fr* uni_host, u32 L, u32 n_polys, u32 inv, hipStream_t st);
extern "C" int zk_ntt_join_launch(const void* work, Fr* out, u64 n, u64 out_es, u32 n_emails, hipStream_t st);
extern "C" int zk_ntt_bitrev_launch(Fr* data, u32 L, u32 n_polys, hipStream_t st);

struct zkwg_ntt {
  int device;
  u32 L;
  u64 n;
  // The tables are in 2^261-Montgomery form (canonical words): the kernels keep values in 9 x 29-bit limb form (zkwg_fr29.h), whose
  // product divides by 2^261; the DATA stay in the callers' 2^256 form (one operand in 2^261 form is what a product needs).
  Fr* d_tw;       // w^k, k < n; w = Fr.w[L] of ffjavascript (nqr = 5)
  Fr* d_scale;    // position p (bit-reversed coefficient index): inc^bitrev(p) / n
  Fr ninv_m;      // 1 / n
};

static Fr pow_m(Fr base_m, const u64 e[4]) {   // Montgomery in / out
  Fr acc = fr_R();
  for (int i = 255; i >= 0; --i) {
    acc = fr_mont_mul(acc, acc);
    if ((e[i >> 6] >> (i & 63)) & 1) acc = fr_mont_mul(acc, base_m);
  }
  return acc;
}
static u32 bitrev_host(u32 x, u32 bits) { u32 r = 0; for (u32 i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1u - i); return r; }

extern "C" {

int zkwg_ntt_create(int device, uint32_t log2_n, zkwg_ntt_t** out) {
  if (!out || log2_n < 2 || log2_n > 26) return ZKWG_RC_BAD_ARG;
  try {
    zkwg_ntt* p = new zkwg_ntt();
    p->device = device; p->L = log2_n; p->n = 1ull << log2_n; p->d_tw = p->d_scale = nullptr;
    // ffjavascript F1Field: s = 28, t = (r - 1) >> 28, w[28] = 5^t, w[i] = w[i+1]^2, shift = 5^2
    const u64 r1[4] = {ZK_P0 - 1, ZK_P1, ZK_P2, ZK_P3};
    u64 t[4];
    for (int i = 0; i < 4; ++i) t[i] = (r1[i] >> 28) | (i < 3 ? r1[i + 1] << 36 : 0);
    const Fr five_m = fr_to_mont(fr_from_u64(5));
    Fr w = pow_m(five_m, t);                                   // w[28]
    Fr wL1 = w;                                                // w[L + 1] (L < 28)
    for (u32 i = 28; i > log2_n; --i) { if (i == log2_n + 1) wL1 = w; w = fr_mont_mul(w, w); }
    const Fr inc = log2_n == 28 ? fr_to_mont(fr_from_u64(25)) : wL1;
    std::vector<Fr> tw(p->n), sc(p->n);
    Fr acc = fr_R();
    for (u64 k = 0; k < p->n; ++k) { tw[k] = acc; acc = fr_mont_mul(acc, w); }
    if (!fr_eq(acc, fr_R()) || !fr_eq(tw[p->n / 2], fr_neg(fr_R()))) { delete p; return ZKWG_RC_BAD_CONFIG; }   // w^n = 1, w^(n/2) = -1
    const u64 e2[4] = {ZK_P0 - 2, ZK_P1, ZK_P2, ZK_P3};
    p->ninv_m = pow_m(fr_to_mont(fr_from_u64(p->n)), e2);
    acc = p->ninv_m;
    for (u64 i = 0; i < p->n; ++i) { sc[bitrev_host((u32)i, log2_n)] = acc; acc = fr_mont_mul(acc, inc); }
    // 2^256 form -> 2^261 form: times 32
    const Fr m32 = fr_to_mont(fr_from_u64(32));
    for (u64 k = 0; k < p->n; ++k) { tw[k] = fr_mont_mul(tw[k], m32); sc[k] = fr_mont_mul(sc[k], m32); }
    p->ninv_m = fr_mont_mul(p->ninv_m, m32);
    if (device >= 0) {
      if (hipSetDevice(device) != hipSuccess) { delete p; return ZKWG_RC_NO_DEVICE; }
      if (hipMalloc((void**)&p->d_tw, p->n * sizeof(Fr)) != hipSuccess || hipMalloc((void**)&p->d_scale, p->n * sizeof(Fr)) != hipSuccess ||
          hipMemcpy(p->d_tw, tw.data(), p->n * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(p->d_scale, sc.data(), p->n * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(p->d_tw); hipFree(p->d_scale); delete p;
        return ZKWG_RC_OOM;
      }
    }
    *out = p;
    return ZKWG_RC_OK;
  } catch (const std::bad_alloc&) {
    return ZKWG_RC_OOM;
  }
}
void zkwg_ntt_destroy(zkwg_ntt_t* p) {
  if (!p) return;
  if (p->device >= 0) { hipSetDevice(p->device); hipFree(p->d_tw); hipFree(p->d_scale); }
  delete p;
}
uint64_t zkwg_ntt_domain(const zkwg_ntt_t* p) { return p ? p->n : 0; }
// three polynomials per email in planar limb form: 16 + 16 + 4 bytes per element
uint64_t zkwg_ntt_work_bytes(const zkwg_ntt_t* p, uint64_t n_emails) { return p ? 3ull * p->n * 36ull * n_emails : 0; }

int zkwg_ntt_transform_device(zkwg_ntt_t* p, void* d_data, uint64_t n_polys, int inverse, void* hip_stream) {
  if (!p || !d_data) return ZKWG_RC_BAD_ARG;
  if (p->device < 0) return ZKWG_RC_NO_DEVICE;
  if (n_polys == 0) return ZKWG_RC_OK;
  if (n_polys > 65535) return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  hipStream_t st = (hipStream_t)hip_stream;
  Fr* d = (Fr*)d_data;
  // the passes run through a temporary limb-form buffer (this entry point is not the pipeline's: it allocates and synchronises)
  void* work = nullptr;
  if (hipMalloc(&work, 36ull * p->n * n_polys) != hipSuccess) { (void)hipGetLastError(); return ZKWG_RC_OOM; }
  int rc;
  if (inverse) {
    // natural -> (DIF, inverse roots, x 1 / n) -> bit-reversed -> permuted back to natural order
    rc = zk_ntt_launch(0, d, 3 * p->n, p->n, p->n, 0, work, d, p->d_tw, nullptr, &p->ninv_m, p->L, (u32)n_polys, 1u, st);
    if (rc == 0) rc = zk_ntt_bitrev_launch(d, p->L, (u32)n_polys, st);
  } else {
    rc = zk_ntt_bitrev_launch(d, p->L, (u32)n_polys, st);
    if (rc == 0) rc = zk_ntt_launch(1, d, 3 * p->n, p->n, p->n, 0, work, d, p->d_tw, nullptr, nullptr, p->L, (u32)n_polys, 0u, st);
  }
  if (hipStreamSynchronize(st) != hipSuccess) rc = -1;
  hipFree(work);
  return rc == 0 ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}

int zkwg_h_evaluations_device(zkwg_ntt_t* p, const void* d_abc, uint64_t abc_stride, uint64_t n_constraints, uint64_t n_emails,
                              void* d_work, void* d_out, uint64_t out_stride, void* hip_stream) {
  if (!p || !d_abc || !d_work || !d_out) return ZKWG_RC_BAD_ARG;
  if (p->device < 0) return ZKWG_RC_NO_DEVICE;
  if (n_emails == 0) return ZKWG_RC_OK;
  if (n_constraints > p->n || abc_stride < 96 * n_constraints || (abc_stride & 31) || out_stride < 32 * p->n || (out_stride & 31) || 3 * n_emails > 65535)
    return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  hipStream_t st = (hipStream_t)hip_stream;
  void* work = d_work;
  const u32 np = (u32)(3 * n_emails);
  // 3 inverse transforms per email straight from A.w | B.w | C.w (zero-padded to the domain), leaving the coset-shifted
  // coefficients inc^i a_i in bit-reversed order; 3 forward transforms from that order; a b - c
  int rc = zk_ntt_launch(0, (const Fr*)d_abc, abc_stride / 32, n_constraints, n_constraints, 0, work, nullptr, p->d_tw, p->d_scale, nullptr, p->L, np, 1u, st);
  if (rc == 0) rc = zk_ntt_launch(1, nullptr, 0, 0, p->n, 1, work, nullptr, p->d_tw, nullptr, nullptr, p->L, np, 0u, st);
  if (rc == 0) rc = zk_ntt_join_launch(work, (Fr*)d_out, p->n, out_stride / 32, (u32)n_emails, st);
  return rc == 0 ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}

}  // extern "C"
