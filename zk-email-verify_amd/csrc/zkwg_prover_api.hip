// `groth16.prove` for the emails of a prepared batch, behind one C entry point (include/zkwg.h "prover"): the second half of
// snarkjs.groth16.fullProve (reference call site: packages/helpers/src/chunked-zkey.ts:80-84) as a host such as the N-API addon binds it.
// It only ORCHESTRATES the stages that have their own entry points -- zkwg_expand_device (the witness as the sums' scalars),
// zkwg_expand_abc_device (buildABC1), zkwg_h_evaluations_device (ifft / coset shift / fft / joinABC), zkwg_msm_enqueue_device (the five
// multiExpAffine), zkwg_groth16_assemble -- with `slots` proofs in flight, each on its own stream with its own buffers, because a
// multi-exponentiation ends in a few hundred dependent group operations on a handful of lanes that only other proofs' work can hide.
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>
#include "../../include/zkwg.h"

struct ZkProveSlot {
  hipStream_t st = nullptr;
  uint8_t *wit = nullptr, *abc = nullptr, *h = nullptr, *ntt = nullptr, *work = nullptr, *sums = nullptr;
};
struct zkwg_prover {
  zkwg_circuit_t* c;
  int device;
  uint64_t n_rows, n_public, W;
  uint32_t power;
  zkwg_ntt_t* ntt = nullptr;
  zkwg_msm_t *ma = nullptr, *mb1 = nullptr, *mb2 = nullptr, *mc = nullptr, *mh = nullptr;
  uint8_t alpha1[64], beta1[64], beta2[128], delta1[64], delta2[128];
  uint64_t work_bytes = 0;
  std::vector<ZkProveSlot> slots;
};

static void prover_free(zkwg_prover* p) {
  if (!p) return;
  hipSetDevice(p->device);
  for (ZkProveSlot& s : p->slots) {
    if (s.st) hipStreamDestroy(s.st);
    hipFree(s.wit); hipFree(s.abc); hipFree(s.h); hipFree(s.ntt); hipFree(s.work); hipFree(s.sums);
  }
  zkwg_msm_destroy(p->ma); zkwg_msm_destroy(p->mb1); zkwg_msm_destroy(p->mb2); zkwg_msm_destroy(p->mc); zkwg_msm_destroy(p->mh);
  zkwg_ntt_destroy(p->ntt);
  delete p;
}

extern "C" {

int zkwg_prover_create(zkwg_circuit_t* c, int device, const uint8_t* r1cs, uint64_t r1cs_len, uint64_t n_rows, const zkwg_proving_key* key,
                       uint32_t slots, zkwg_prover_t** out) {
  if (!c || !key || !out || !key->a || !key->b1 || !key->b2 || !key->c || !key->h || slots == 0 || slots > 256) return ZKWG_RC_BAD_ARG;
  if (device < 0) return ZKWG_RC_NO_DEVICE;
  const uint64_t W = zkwg_witness_len(c), n = 1ull << key->log2_domain;
  if (key->n_wires != W || key->n_public + 1 >= W || n_rows > n || key->log2_domain > 28) return ZKWG_RC_BAD_CONFIG;
  if (hipSetDevice(device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  int rc = ZKWG_RC_OK;
  if (r1cs) rc = zkwg_circuit_attach_r1cs(c, r1cs, r1cs_len);     // (NULL: the caller attached the system already)
  if (rc != ZKWG_RC_OK) return rc;
  if (zkwg_abc_bytes(c) != 96 * n_rows) return ZKWG_RC_BAD_CONFIG;
  zkwg_prover* p = new zkwg_prover();
  p->c = c; p->device = device; p->n_rows = n_rows; p->n_public = key->n_public; p->W = W; p->power = (uint32_t)key->log2_domain;
  memcpy(p->alpha1, key->alpha1, 64); memcpy(p->beta1, key->beta1, 64); memcpy(p->beta2, key->beta2, 128);
  memcpy(p->delta1, key->delta1, 64); memcpy(p->delta2, key->delta2, 128);
  auto plan = [&](int group, const void* bases, uint64_t count, zkwg_msm_t** m) {
    if (rc != ZKWG_RC_OK) return;
    rc = key->bases_on_device ? zkwg_msm_create_device(device, group, bases, count, 0, m)
                              : (group == 1 ? zkwg_msm_create(device, (const uint8_t*)bases, count, 0, m) : zkwg_msm_create_g2(device, (const uint8_t*)bases, count, 0, m));
  };
  rc = zkwg_ntt_create(device, p->power, &p->ntt);
  plan(1, key->a, W, &p->ma); plan(1, key->b1, W, &p->mb1); plan(2, key->b2, W, &p->mb2);
  plan(1, key->c, W - key->n_public - 1, &p->mc); plan(1, key->h, n, &p->mh);
  if (rc != ZKWG_RC_OK) { prover_free(p); return rc; }
  for (zkwg_msm_t* m : {p->ma, p->mb1, p->mb2, p->mc, p->mh}) p->work_bytes = std::max<uint64_t>(p->work_bytes, zkwg_msm_work_bytes(m));
  p->slots.resize(slots);
  for (ZkProveSlot& s : p->slots) {
    bool ok = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void**)&s.wit, zkwg_witness_bytes(c)) == hipSuccess && hipMalloc((void**)&s.abc, zkwg_abc_bytes(c)) == hipSuccess &&
              hipMalloc((void**)&s.h, 32ull << p->power) == hipSuccess && hipMalloc((void**)&s.ntt, zkwg_ntt_work_bytes(p->ntt, 1)) == hipSuccess &&
              hipMalloc((void**)&s.work, p->work_bytes + 256) == hipSuccess && hipMalloc((void**)&s.sums, 5 * 256) == hipSuccess;
    if (!ok) { prover_free(p); (void)hipGetLastError(); return ZKWG_RC_OOM; }
  }
  *out = p;
  return ZKWG_RC_OK;
}
void zkwg_prover_destroy(zkwg_prover_t* p) { prover_free(p); }

int zkwg_prover_prove_prepared(zkwg_prover_t* p, const void* d_in, uint64_t n, const void* d_scratch, const uint64_t* indices, uint64_t n_idx,
                               const uint8_t* blinding, uint8_t* out_proofs) {
  if (!p || !d_in || !d_scratch || !indices || !blinding || !out_proofs) return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  zkwg_circuit_t* c = p->c;
  const uint64_t wb = zkwg_witness_bytes(c), ab = zkwg_abc_bytes(c);
  const size_t S = p->slots.size();
  for (uint64_t w0 = 0; w0 < n_idx; w0 += S) {
    const uint64_t cnt = std::min<uint64_t>(S, n_idx - w0);
    int rc = ZKWG_RC_OK;
    for (uint64_t j = 0; j < cnt && rc == ZKWG_RC_OK; ++j) {
      ZkProveSlot& s = p->slots[j];
      const uint64_t e = indices[w0 + j];
      if (e >= n) { rc = ZKWG_RC_BAD_ARG; break; }
      uint8_t* work = (uint8_t*)(((uintptr_t)s.work + 255) & ~(uintptr_t)255);
      rc = zkwg_expand_device(c, d_in, n, d_scratch, e, 1, s.wit, wb, s.st);
      if (rc == ZKWG_RC_OK) rc = zkwg_expand_abc_device(c, d_in, n, d_scratch, e, 1, 1, s.abc, ab, s.st);
      if (rc == ZKWG_RC_OK) rc = zkwg_h_evaluations_device(p->ntt, s.abc, ab, p->n_rows, 1, s.ntt, s.h, 32ull << p->power, s.st);
      if (rc == ZKWG_RC_OK) rc = zkwg_msm_enqueue_device(p->ma, s.wit, 0, 1, work, s.sums, s.st);
      if (rc == ZKWG_RC_OK) rc = zkwg_msm_enqueue_device(p->mb1, s.wit, 0, 1, work, s.sums + 256, s.st);
      if (rc == ZKWG_RC_OK) rc = zkwg_msm_enqueue_device(p->mb2, s.wit, 0, 1, work, s.sums + 512, s.st);
      if (rc == ZKWG_RC_OK) rc = zkwg_msm_enqueue_device(p->mc, s.wit + 32 * (p->n_public + 1), 0, 1, work, s.sums + 768, s.st);
      if (rc == ZKWG_RC_OK) rc = zkwg_msm_enqueue_device(p->mh, s.h, 1, 0, work, s.sums + 1024, s.st);
    }
    for (uint64_t j = 0; j < cnt; ++j) {
      ZkProveSlot& s = p->slots[j];
      uint8_t raw[5 * 256], pts[5][128];
      if (hipStreamSynchronize(s.st) != hipSuccess && rc == ZKWG_RC_OK) rc = ZKWG_RC_HIP_ERROR;
      if (rc != ZKWG_RC_OK) continue;
      if (hipMemcpy(raw, s.sums, sizeof raw, hipMemcpyDeviceToHost) != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; continue; }
      for (int k = 0; k < 5; ++k) zkwg_msm_finish_host(k == 2 ? 2 : 1, raw + 256 * k, 1, pts[k]);
      uint8_t* o = out_proofs + 256 * (w0 + j);
      const uint8_t* bl = blinding + 64 * (w0 + j);
      rc = zkwg_groth16_assemble(pts[0], pts[1], pts[2], pts[3], pts[4], p->alpha1, p->beta1, p->beta2, p->delta1, p->delta2, bl, bl + 32, o, o + 64, o + 192);
    }
    if (rc != ZKWG_RC_OK) return rc;
  }
  return ZKWG_RC_OK;
}

// inputs -> proofs: n packed input records (zkwg_pack_input) on the host -> status[n] (circom_runtime codes) and, for every email whose
// witness exists (status 0), its proof; the proof bytes of a failed email are zero
int zkwg_prover_prove_batch(zkwg_prover_t* p, const uint8_t* packed, uint64_t n, const uint8_t* blinding, int32_t* status, uint8_t* out_proofs) {
  if (!p || !packed || !blinding || !status || !out_proofs) return ZKWG_RC_BAD_ARG;
  if (n == 0) return ZKWG_RC_OK;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  zkwg_circuit_t* c = p->c;
  const uint64_t is = zkwg_input_stride(c);
  uint8_t *d_in = nullptr, *d_scr = nullptr; int32_t* d_st = nullptr;
  int rc = ZKWG_RC_OK;
  if (hipMalloc((void**)&d_in, n * is) != hipSuccess || hipMalloc((void**)&d_scr, zkwg_scratch_bytes(c, n)) != hipSuccess || hipMalloc((void**)&d_st, n * 4) != hipSuccess) rc = ZKWG_RC_OOM;
  if (rc == ZKWG_RC_OK && hipMemcpy(d_in, packed, n * is, hipMemcpyHostToDevice) != hipSuccess) rc = ZKWG_RC_HIP_ERROR;
  if (rc == ZKWG_RC_OK) rc = zkwg_prepare_device(c, d_in, n, d_st, d_scr, nullptr);
  // (zkwg_expand_device orders behind the prepare on the null stream only if it runs on that stream: settle the batch first)
  if (rc == ZKWG_RC_OK && hipDeviceSynchronize() != hipSuccess) rc = ZKWG_RC_HIP_ERROR;
  if (rc == ZKWG_RC_OK && hipMemcpy(status, d_st, n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = ZKWG_RC_HIP_ERROR;
  if (rc == ZKWG_RC_OK) {
    std::vector<uint64_t> idx;
    std::vector<uint8_t> bl;
    for (uint64_t i = 0; i < n; ++i) if (status[i] == 0) { idx.push_back(i); bl.insert(bl.end(), blinding + 64 * i, blinding + 64 * i + 64); }
    std::vector<uint8_t> proofs(256 * idx.size());
    if (!idx.empty()) rc = zkwg_prover_prove_prepared(p, d_in, n, d_scr, idx.data(), idx.size(), bl.data(), proofs.data());
    memset(out_proofs, 0, 256 * n);
    if (rc == ZKWG_RC_OK) for (size_t k = 0; k < idx.size(); ++k) memcpy(out_proofs + 256 * idx[k], proofs.data() + 256 * k, 256);
  }
  hipFree(d_in); hipFree(d_scr); hipFree(d_st);
  return rc;
}

}
