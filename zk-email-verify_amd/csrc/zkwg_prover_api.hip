// `groth16.prove` for the emails of a prepared batch, behind one C entry point (include/zkwg.h "prover"): the second half of
// snarkjs.groth16.fullProve (reference call site: packages/helpers/src/chunked-zkey.ts:80-84) as a host such as the N-API addon binds it.
// It only ORCHESTRATES the stages that have their own entry points -- zkwg_expand_device (the witness as the sums' scalars),
// zkwg_expand_abc_device (buildABC1), zkwg_h_evaluations_device (ifft / coset shift / fft / joinABC), zkwg_msm_* (the five
// multiExpAffine), zkwg_groth16_assemble.
//
// Round 6: proofs are made E AT A TIME.  A context owns the buffers of E emails and three streams; every stage is ONE launch series for
// its E emails (emails are the second grid dimension of every kernel), so a proof costs 1 / E of the launches and the thin tails of the
// sums are E lanes wide:
//     stream h   witnesses of the E emails -> A.w | B.w | C.w -> H evaluations -> the H sum
//     stream w   (after the witnesses) one classification pass for the three witness-shaped base sets -> sums a, b1, c
//     stream g   (after the classification) the G2 sum b2
// and the contexts roll: while the host assembles the proofs of a finished context (a dozen group operations each, host arithmetic), the
// other contexts' series keep the device busy; nothing waits for a whole wave of proofs, and nothing depends on the number of hardware
// queues (round 5: one stream per proof, GPU_MAX_HW_QUEUES = 16 or half the rate).
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../../include/zkwg.h"
#include "zkwg_fq.h"

struct ZkProveCtx {
  hipStream_t st_h = nullptr, st_w = nullptr, st_g = nullptr;
  hipEvent_t ev_wit = nullptr, ev_lists = nullptr, ev_w = nullptr, ev_g = nullptr, ev_done = nullptr;
  uint8_t *wit = nullptr, *abc = nullptr, *h = nullptr, *ntt = nullptr, *work_h = nullptr, *work_w = nullptr, *work_g = nullptr, *sums = nullptr;
  uint8_t* lists[3] = {nullptr, nullptr, nullptr};
  uint8_t* host_sums = nullptr;        // pinned: E x 5 x 256
  uint64_t first = 0, count = 0;       // positions [first, first + count) of the call's index list are in flight here
  bool busy = false;
};
struct zkwg_prover {
  zkwg_circuit_t* c;
  int device;
  uint64_t n_rows, n_public, W;
  uint32_t power, E;
  int c_from_ab = 0;                   // 1: the attached system has no C rows (a zkey's section 4): C.w = A.w o B.w (zkwg_prover_create_zkey)
  zkwg_ntt_t* ntt = nullptr;
  zkwg_msm_t *ma = nullptr, *mb1 = nullptr, *mb2 = nullptr, *mc = nullptr, *mh = nullptr;
  uint8_t alpha1[64], beta1[64], beta2[128], delta1[64], delta2[128];
  std::vector<ZkProveCtx> ctx;
};
void zk_abc_c_from_ab_launch(void* d_abc, uint64_t abc_stride, uint64_t n_rows, uint32_t n_emails, hipStream_t st);      // zkwg_kernels_handoff.hip

static void prover_free(zkwg_prover* p) {
  if (!p) return;
  hipSetDevice(p->device);
  for (ZkProveCtx& s : p->ctx) {
    for (hipStream_t st : {s.st_h, s.st_w, s.st_g}) if (st) hipStreamDestroy(st);
    for (hipEvent_t ev : {s.ev_wit, s.ev_lists, s.ev_w, s.ev_g, s.ev_done}) if (ev) hipEventDestroy(ev);
    for (uint8_t* q : {s.wit, s.abc, s.h, s.ntt, s.work_h, s.work_w, s.work_g, s.sums, s.lists[0], s.lists[1], s.lists[2]}) if (q) hipFree(q);
    if (s.host_sums) hipHostFree(s.host_sums);
  }
  zkwg_msm_destroy(p->ma); zkwg_msm_destroy(p->mb1); zkwg_msm_destroy(p->mb2); zkwg_msm_destroy(p->mc); zkwg_msm_destroy(p->mh);
  zkwg_ntt_destroy(p->ntt);
  delete p;
}
// `slots` proofs in flight = contexts x emails per series
static void prover_shape(uint32_t slots, uint32_t& n_ctx, uint32_t& E) {
  n_ctx = slots >= 6 ? 3u : slots >= 2 ? 2u : 1u;
  // (tuning knob, DESIGN.md section 8: ZKWG_PROVER_CONTEXTS = 1 .. 4 rolling contexts instead of the default)
  if (const char* v = getenv("ZKWG_PROVER_CONTEXTS")) { const int k = atoi(v); if (k >= 1 && k <= 4 && (uint32_t)k <= slots) n_ctx = (uint32_t)k; }
  E = (slots + n_ctx - 1) / n_ctx;
  if (E > 32) E = 32;
}

static int prover_new(zkwg_circuit_t* c, int device, uint64_t n_rows, const zkwg_proving_key* key, uint32_t slots, int c_from_ab, zkwg_prover_t** out) {
  const uint64_t W = zkwg_witness_len(c), n = 1ull << key->log2_domain;
  if (key->n_wires != W || key->n_public + 1 >= W || n_rows > n || key->log2_domain > 28) return ZKWG_RC_BAD_CONFIG;
  if (zkwg_abc_bytes(c) != 96 * n_rows) return ZKWG_RC_BAD_CONFIG;
  if (hipSetDevice(device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  zkwg_prover* p = new zkwg_prover();
  p->c = c; p->device = device; p->n_rows = n_rows; p->n_public = key->n_public; p->W = W; p->power = (uint32_t)key->log2_domain; p->c_from_ab = c_from_ab;
  memcpy(p->alpha1, key->alpha1, 64); memcpy(p->beta1, key->beta1, 64); memcpy(p->beta2, key->beta2, 128);
  memcpy(p->delta1, key->delta1, 64); memcpy(p->delta2, key->delta2, 128);
  uint32_t n_ctx;
  prover_shape(slots, n_ctx, p->E);
  const uint64_t E = p->E;
  // What the contexts will need is set aside BEFORE the tables are sized (the tables took what was free and the slots then
  // failed): a table's K copies are only made when they fit beside it, otherwise that plan keeps the classic layout.
  const uint64_t wb = zkwg_witness_bytes(c), ab = zkwg_abc_bytes(c), hb = 32ull << p->power;
  // (work buffers: bounded by the witness-shaped G2 plan at window 13 and the H plan at window 16 -- estimated with their own formulas below)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { free_b = 0; (void)hipGetLastError(); }
  int rc = zkwg_ntt_create(device, p->power, &p->ntt);
  const uint64_t ntt_b = rc == ZKWG_RC_OK ? zkwg_ntt_work_bytes(p->ntt, E) : 0;
  const int ww = W >= (1u << 16) ? 13 : 0, hs0 = n >= (1u << 18) ? 64 : 16;
  const uint64_t est_ctx = E * (wb + ab + hb + zkwg_msm_estimate_work_bytes(1, W, ww, 16) + zkwg_msm_estimate_work_bytes(2, W, ww, 16) +
                                zkwg_msm_estimate_work_bytes(1, n, 0, hs0) + 3 * 8ull * W) + ntt_b + (64ull << 20);
  const uint64_t reserve = n_ctx * est_ctx + (2ull << 30);
  uint64_t budget = free_b > reserve ? free_b - reserve : 1;      // (1: no room for copies -- classic layout everywhere)
  auto plan = [&](int group, const void* bases, uint64_t count, int window, int slice0, zkwg_msm_t** m) {
    if (rc != ZKWG_RC_OK) return;
    rc = zkwg_msm_create_ex(device, group, bases, key->bases_on_device, count, window, slice0, budget, m);
    if (rc == ZKWG_RC_OK) { const uint64_t used = zkwg_msm_table_bytes(*m); budget = budget > used ? budget - used : 1; }
  };
  // witness-shaped sums: a few ten thousand full-size scalars per email -> window 13 (4,096 buckets, ~ 170 entries each), slices of 16;
  // the H sum: 2^power full-size scalars -> window 16, slices of 64
  plan(1, key->h, n, 0, hs0, &p->mh);
  // (tuning knobs, DESIGN.md section 8: ZKWG_PROVER_WITNESS_SLICE = level-0 slice of the witness-shaped sums, ZKWG_PROVER_WITNESS_WINDOW)
  const int ws0 = getenv("ZKWG_PROVER_WITNESS_SLICE") ? std::max(1, std::min(1024, atoi(getenv("ZKWG_PROVER_WITNESS_SLICE")))) : 16;
  const int wwin = getenv("ZKWG_PROVER_WITNESS_WINDOW") ? std::max(2, std::min(16, atoi(getenv("ZKWG_PROVER_WITNESS_WINDOW")))) : ww;
  plan(2, key->b2, W, wwin, ws0, &p->mb2);
  plan(1, key->a, W, wwin, ws0, &p->ma); plan(1, key->b1, W, wwin, ws0, &p->mb1);
  plan(1, key->c, W - key->n_public - 1, wwin, ws0, &p->mc);
  if (rc != ZKWG_RC_OK) { prover_free(p); return rc; }
  const uint64_t work_w = std::max(zkwg_msm_work_bytes(p->ma), std::max(zkwg_msm_work_bytes(p->mb1), zkwg_msm_work_bytes(p->mc)));
  p->ctx.resize(n_ctx);
  for (ZkProveCtx& s : p->ctx) {
    // the witness-shaped sums are chains of short kernels (sort, slices, joins, plane folds) beside stream h's few long ones (transforms, the H sum's
    // slices: 16 k wavefronts each): on higher-priority streams their wavefronts are placed first, so the chains do not queue behind the long kernels
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);           // (numerically lower = higher priority)
    static const int use_prio = getenv("ZKWG_PROVER_PRIO") ? atoi(getenv("ZKWG_PROVER_PRIO")) : 1;
    const int pw = use_prio ? prio_hi : prio_lo;
    bool ok = hipStreamCreateWithPriority(&s.st_h, hipStreamNonBlocking, prio_lo) == hipSuccess && hipStreamCreateWithPriority(&s.st_w, hipStreamNonBlocking, pw) == hipSuccess &&
              hipStreamCreateWithPriority(&s.st_g, hipStreamNonBlocking, pw) == hipSuccess;
    for (hipEvent_t* ev : {&s.ev_wit, &s.ev_lists, &s.ev_w, &s.ev_g, &s.ev_done}) ok = ok && hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.wit, E * wb) == hipSuccess && hipMalloc((void**)&s.abc, E * ab) == hipSuccess && hipMalloc((void**)&s.h, E * hb) == hipSuccess &&
         hipMalloc((void**)&s.ntt, ntt_b) == hipSuccess && hipMalloc((void**)&s.work_h, E * zkwg_msm_work_bytes(p->mh)) == hipSuccess &&
         hipMalloc((void**)&s.work_w, E * work_w) == hipSuccess && hipMalloc((void**)&s.work_g, E * zkwg_msm_work_bytes(p->mb2)) == hipSuccess &&
         hipMalloc((void**)&s.sums, E * 5 * 256) == hipSuccess && hipMalloc((void**)&s.lists[0], zkwg_msm_lists_bytes(p->ma, E)) == hipSuccess &&
         hipMalloc((void**)&s.lists[1], zkwg_msm_lists_bytes(p->mb1, E)) == hipSuccess && hipMalloc((void**)&s.lists[2], zkwg_msm_lists_bytes(p->mc, E)) == hipSuccess &&
         hipHostMalloc((void**)&s.host_sums, E * 5 * 256, hipHostMallocDefault) == hipSuccess;
    if (!ok) { prover_free(p); (void)hipGetLastError(); return ZKWG_RC_OOM; }
  }
  *out = p;
  return ZKWG_RC_OK;
}

// one series for the emails indices[first .. first + count) on context s (nothing synchronised)
static int prover_enqueue(zkwg_prover* p, ZkProveCtx& s, const void* d_in, uint64_t n, const void* d_scratch, const uint64_t* indices, uint64_t first, uint64_t count) {
  zkwg_circuit_t* c = p->c;
  const uint64_t wb = zkwg_witness_bytes(c), ab = zkwg_abc_bytes(c), hb = 32ull << p->power;
  int rc = ZKWG_RC_OK;
  // witnesses and A.w | B.w | C.w: one launch per run of consecutive emails
  for (uint64_t j = 0; j < count && rc == ZKWG_RC_OK;) {
    uint64_t run = 1;
    while (j + run < count && indices[first + j + run] == indices[first + j] + run) ++run;
    if (indices[first + j] + run > n) return ZKWG_RC_BAD_ARG;
    rc = zkwg_expand_device(c, d_in, n, d_scratch, indices[first + j], run, s.wit + j * wb, wb, s.st_h);
    if (rc == ZKWG_RC_OK) rc = zkwg_expand_abc_device(c, d_in, n, d_scratch, indices[first + j], run, 1, s.abc + j * ab, ab, s.st_h);
    j += run;
  }
  if (rc != ZKWG_RC_OK) return rc;
  if (hipEventRecord(s.ev_wit, s.st_h) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  // stream w: classification for the three base sets, then a, b1, c; stream g: b2
  if (hipStreamWaitEvent(s.st_w, s.ev_wit, 0) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  zkwg_msm_t* plans[3] = {p->ma, p->mb1, p->mc};
  const uint64_t firsts[3] = {0, 0, p->n_public + 1};
  void* lists[3] = {s.lists[0], s.lists[1], s.lists[2]};
  rc = zkwg_msm_classify_device(plans, firsts, 3, s.wit, wb, p->W, count, 0, 1, lists, s.st_w);
  if (rc != ZKWG_RC_OK) return rc;
  if (hipEventRecord(s.ev_lists, s.st_w) != hipSuccess || hipStreamWaitEvent(s.st_g, s.ev_lists, 0) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  rc = zkwg_msm_enqueue_lists_device(p->mb2, s.wit, wb, count, 0, s.lists[1], 1, s.work_g, s.sums + 2 * p->E * 256, s.st_g);
  if (rc == ZKWG_RC_OK) rc = zkwg_msm_enqueue_lists_device(p->ma, s.wit, wb, count, 0, s.lists[0], 1, s.work_w, s.sums, s.st_w);
  if (rc == ZKWG_RC_OK) rc = zkwg_msm_enqueue_lists_device(p->mb1, s.wit, wb, count, 0, s.lists[1], 1, s.work_w, s.sums + 1 * p->E * 256, s.st_w);
  if (rc == ZKWG_RC_OK) rc = zkwg_msm_enqueue_lists_device(p->mc, s.wit + 32 * (p->n_public + 1), wb, count, 0, s.lists[2], 1, s.work_w, s.sums + 3 * p->E * 256, s.st_w);
  // stream h: the transforms and the H sum
  if (rc == ZKWG_RC_OK && p->c_from_ab) zk_abc_c_from_ab_launch(s.abc, ab, p->n_rows, (uint32_t)count, s.st_h);
  if (rc == ZKWG_RC_OK) rc = zkwg_h_evaluations_device(p->ntt, s.abc, ab, p->n_rows, count, s.ntt, s.h, hb, s.st_h);
  if (rc == ZKWG_RC_OK) rc = zkwg_msm_enqueue_batch_device(p->mh, s.h, hb, count, 1, 0, s.work_h, s.sums + 4 * p->E * 256, s.st_h);
  if (rc != ZKWG_RC_OK) return rc;
  if (hipEventRecord(s.ev_w, s.st_w) != hipSuccess || hipEventRecord(s.ev_g, s.st_g) != hipSuccess || hipStreamWaitEvent(s.st_h, s.ev_w, 0) != hipSuccess ||
      hipStreamWaitEvent(s.st_h, s.ev_g, 0) != hipSuccess)
    return ZKWG_RC_HIP_ERROR;
  // (sums: five arrays of E accumulators, array k at sums + k E 256)
  if (hipMemcpyAsync(s.host_sums, s.sums, p->E * 5 * 256, hipMemcpyDeviceToHost, s.st_h) != hipSuccess || hipEventRecord(s.ev_done, s.st_h) != hipSuccess)
    return ZKWG_RC_HIP_ERROR;
  s.first = first; s.count = count; s.busy = true;
  return ZKWG_RC_OK;
}
// wait for context s, assemble its proofs
static int prover_finish(zkwg_prover* p, ZkProveCtx& s, const uint8_t* blinding, uint8_t* out_proofs) {
  if (!s.busy) return ZKWG_RC_OK;
  s.busy = false;
  if (hipEventSynchronize(s.ev_done) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  int rc = ZKWG_RC_OK;
  for (uint64_t j = 0; j < s.count && rc == ZKWG_RC_OK; ++j) {
    uint8_t pts[5][128];
    // sum k of email j: array k holds E accumulators of 256 bytes (G1 uses the first 128)
    for (int k = 0; k < 5; ++k) zkwg_msm_finish_host(k == 2 ? 2 : 1, s.host_sums + (uint64_t)k * p->E * 256 + j * (k == 2 ? 256 : 128), 1, pts[k]);
    uint8_t* o = out_proofs + 256 * (s.first + j);
    const uint8_t* bl = blinding + 64 * (s.first + j);
    rc = zkwg_groth16_assemble(pts[0], pts[1], pts[2], pts[3], pts[4], p->alpha1, p->beta1, p->beta2, p->delta1, p->delta2, bl, bl + 32, o, o + 64, o + 192);
  }
  return rc;
}

extern "C" {

int zkwg_prover_create(zkwg_circuit_t* c, int device, const uint8_t* r1cs, uint64_t r1cs_len, uint64_t n_rows, const zkwg_proving_key* key,
                       uint32_t slots, zkwg_prover_t** out) {
  if (!c || !key || !out || !key->a || !key->b1 || !key->b2 || !key->c || !key->h || slots == 0 || slots > 256) return ZKWG_RC_BAD_ARG;
  if (device < 0) return ZKWG_RC_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  int rc = ZKWG_RC_OK;
  if (r1cs) rc = zkwg_circuit_attach_r1cs(c, r1cs, r1cs_len);     // (NULL: the caller attached the system already)
  if (rc != ZKWG_RC_OK) return rc;
  return prover_new(c, device, n_rows, key, slots, 0, out);
}
// The prover from the zkey ALONE -- what `groth16.prove(zkey, wtns)` takes (reference call site: fullProve(input, wasm, zkey),
// packages/helpers/src/chunked-zkey.ts:80-84).  A snarkjs groth16 .zkey carries its own constraint system: section 4 lists the
// coefficients of A and B (matrix, row, wire, value x R^2), the nPublic + 1 rows snarkjs appends to A included, and buildABC1 takes the
// third block as A.w o B.w; sections 5-9 are the bases.  The rows become an in-memory `.r1cs` with empty C combinations that is attached
// to the handle like any other system (zkwg_circuit_attach_r1cs: A.w and B.w then come straight from the compact image), and every
// series computes C.w = A.w o B.w before the transforms (zk_abc_c_from_ab).  Container layout restated from snarkjs [EXT]:
// zkwg/zkey.py has the field-by-field description.
static bool zkey_sections(const uint8_t* z, uint64_t len, uint64_t (&off)[11], uint64_t (&size)[11]) {
  for (int i = 0; i < 11; ++i) off[i] = size[i] = 0;
  if (len < 12 || memcmp(z, "zkey", 4) != 0) return false;
  uint32_t version, nsec;
  memcpy(&version, z + 4, 4); memcpy(&nsec, z + 8, 4);
  if (version != 1) return false;
  uint64_t pos = 12;
  for (uint32_t i = 0; i < nsec; ++i) {
    if (pos + 12 > len) return false;
    uint32_t id; uint64_t sz;
    memcpy(&id, z + pos, 4); memcpy(&sz, z + pos + 4, 8);
    pos += 12;
    if (sz > len - pos) return false;
    if (id >= 1 && id <= 10) { off[id] = pos; size[id] = sz; }
    pos += sz;
  }
  return true;
}
int zkwg_prover_create_zkey(zkwg_circuit_t* c, int device, const uint8_t* zkey, uint64_t zkey_len, uint32_t slots, zkwg_prover_t** out) {
  if (!c || !zkey || !out || slots == 0 || slots > 256) return ZKWG_RC_BAD_ARG;
  if (device < 0) return ZKWG_RC_NO_DEVICE;
  uint64_t off[11], size[11];
  if (!zkey_sections(zkey, zkey_len, off, size)) return ZKWG_RC_BAD_CONFIG;
  for (int need : {1, 2, 4, 5, 6, 7, 8, 9}) if (!off[need]) return ZKWG_RC_BAD_CONFIG;
  uint32_t protocol;
  memcpy(&protocol, zkey + off[1], 4);
  if (size[1] < 4 || protocol != 1) return ZKWG_RC_BAD_CONFIG;                     // groth16
  // header: n8q, q, n8r, r, nVars, nPublic, domainSize, alpha1, beta1, beta2, gamma2, delta1, delta2
  const uint8_t* h = zkey + off[2];
  if (size[2] < 4 + 32 + 4 + 32 + 12 + 64 + 64 + 128 + 128 + 64 + 128) return ZKWG_RC_BAD_CONFIG;
  uint32_t n8q, n8r, n_vars, n_public, domain;
  memcpy(&n8q, h, 4); memcpy(&n8r, h + 36, 4);
  const Fq q = fq_p(); const Fr r = fr_p();
  if (n8q != 32 || n8r != 32 || memcmp(h + 4, q.l, 32) != 0 || memcmp(h + 40, r.l, 32) != 0) return ZKWG_RC_BAD_CONFIG;      // BN254
  memcpy(&n_vars, h + 72, 4); memcpy(&n_public, h + 76, 4); memcpy(&domain, h + 80, 4);
  if (domain == 0 || (domain & (domain - 1)) || n_public + 1 >= n_vars) return ZKWG_RC_BAD_CONFIG;
  uint32_t power = 0;
  while ((1u << power) < domain) ++power;
  if (size[5] != 64ull * n_vars || size[6] != 64ull * n_vars || size[7] != 128ull * n_vars || size[8] != 64ull * (n_vars - n_public - 1) || size[9] != 64ull * domain)
    return ZKWG_RC_BAD_CONFIG;
  zkwg_proving_key key;
  memset(&key, 0, sizeof key);
  key.n_wires = n_vars; key.n_public = n_public; key.log2_domain = power;
  key.a = zkey + off[5]; key.b1 = zkey + off[6]; key.b2 = zkey + off[7]; key.c = zkey + off[8]; key.h = zkey + off[9]; key.bases_on_device = 0;
  const uint8_t* pts = h + 84;
  memcpy(key.alpha1, pts, 64); memcpy(key.beta1, pts + 64, 64); memcpy(key.beta2, pts + 128, 128);
  memcpy(key.delta1, pts + 384, 64); memcpy(key.delta2, pts + 448, 128);                           // (gamma2 sits between beta2 and delta1)
  // section 4 -> rows of A and B
  if (size[4] < 4) return ZKWG_RC_BAD_CONFIG;
  uint32_t n_coef;
  memcpy(&n_coef, zkey + off[4], 4);
  if (size[4] != 4 + 44ull * n_coef) return ZKWG_RC_BAD_CONFIG;
  const uint8_t* cf = zkey + off[4] + 4;
  uint64_t n_rows = 0;
  for (uint32_t i = 0; i < n_coef; ++i) {
    uint32_t m, row, wire;
    memcpy(&m, cf + 44ull * i, 4); memcpy(&row, cf + 44ull * i + 4, 4); memcpy(&wire, cf + 44ull * i + 8, 4);
    if (m > 1 || wire >= n_vars || row >= domain) return ZKWG_RC_BAD_CONFIG;
    n_rows = std::max<uint64_t>(n_rows, (uint64_t)row + 1);
  }
  if (n_rows == 0) return ZKWG_RC_BAD_CONFIG;
  std::vector<uint32_t> cnt(2 * n_rows + 1, 0);                      // terms of A row i at 2 i, of B row i at 2 i + 1
  for (uint32_t i = 0; i < n_coef; ++i) {
    uint32_t m, row;
    memcpy(&m, cf + 44ull * i, 4); memcpy(&row, cf + 44ull * i + 4, 4);
    ++cnt[2ull * row + m];
  }
  // the in-memory .r1cs: per row  A (count, terms) | B (count, terms) | C (0)
  std::vector<uint64_t> at(2 * n_rows);
  uint64_t pos = 0;
  for (uint64_t i = 0; i < n_rows; ++i) {
    at[2 * i] = pos + 4; pos += 4 + 36ull * cnt[2 * i];
    at[2 * i + 1] = pos + 4; pos += 4 + 36ull * cnt[2 * i + 1];
    pos += 4;
  }
  const uint64_t cons_len = pos, hdr_len = 4 + 32 + 16 + 8 + 4, w2l_len = 8ull * n_vars;
  std::vector<uint8_t> file(12 + 3 * 12 + hdr_len + cons_len + w2l_len, 0);
  uint8_t* f = file.data();
  memcpy(f, "r1cs", 4);
  const uint32_t one = 1, three = 3, fs = 32;
  memcpy(f + 4, &one, 4); memcpy(f + 8, &three, 4);
  uint64_t o = 12;
  auto section = [&](uint32_t type, uint64_t sz) { memcpy(f + o, &type, 4); memcpy(f + o + 4, &sz, 8); o += 12; const uint64_t at0 = o; o += sz; return at0; };
  {
    uint8_t* hd = f + section(1, hdr_len);
    memcpy(hd, &fs, 4); memcpy(hd + 4, r.l, 32);
    const uint32_t n_prv = n_vars - 1 - n_public, zero = 0, m32 = (uint32_t)n_rows;
    const uint64_t labels = n_vars;
    memcpy(hd + 36, &n_vars, 4); memcpy(hd + 40, &n_public, 4); memcpy(hd + 44, &zero, 4); memcpy(hd + 48, &n_prv, 4); memcpy(hd + 52, &labels, 8); memcpy(hd + 60, &m32, 4);
  }
  {
    uint8_t* cs = f + section(2, cons_len);
    for (uint64_t i = 0; i < n_rows; ++i) { memcpy(cs + at[2 * i] - 4, &cnt[2 * i], 4); memcpy(cs + at[2 * i + 1] - 4, &cnt[2 * i + 1], 4); }
    for (uint32_t i = 0; i < n_coef; ++i) {
      uint32_t m, row, wire;
      memcpy(&m, cf + 44ull * i, 4); memcpy(&row, cf + 44ull * i + 4, 4); memcpy(&wire, cf + 44ull * i + 8, 4);
      Fr v;
      memcpy(v.l, cf + 44ull * i + 12, 32);
      if (fr_geq(v, r)) return ZKWG_RC_BAD_CONFIG;
      v = fr_from_mont(fr_from_mont(v));                             // stored: coefficient x R^2
      uint8_t* t = cs + at[2ull * row + m];
      memcpy(t, &wire, 4); memcpy(t + 4, v.l, 32);
      at[2ull * row + m] += 36;
    }
  }
  {
    uint8_t* wl = f + section(3, w2l_len);
    for (uint64_t i = 0; i < n_vars; ++i) memcpy(wl + 8 * i, &i, 8);
  }
  if (hipSetDevice(device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  if (zkwg_witness_len(c) != n_vars) return ZKWG_RC_BAD_CONFIG;
  // (a handle that already carries a system of this size keeps it: a second prover over the same handle, e.g. with more slots)
  if (zkwg_abc_bytes(c) == 0) {
    const int rc = zkwg_circuit_attach_r1cs(c, file.data(), file.size());
    if (rc != ZKWG_RC_OK) return rc;
  }
  file.clear(); file.shrink_to_fit();
  return prover_new(c, device, n_rows, &key, slots, 1, out);
}
void zkwg_prover_destroy(zkwg_prover_t* p) { prover_free(p); }
uint32_t zkwg_prover_emails_per_series(const zkwg_prover_t* p) { return p ? p->E : 0; }
uint32_t zkwg_prover_contexts(const zkwg_prover_t* p) { return p ? (uint32_t)p->ctx.size() : 0; }

int zkwg_prover_prove_prepared(zkwg_prover_t* p, const void* d_in, uint64_t n, const void* d_scratch, const uint64_t* indices, uint64_t n_idx,
                               const uint8_t* blinding, uint8_t* out_proofs) {
  if (!p || !d_in || !d_scratch || !indices || !blinding || !out_proofs) return ZKWG_RC_BAD_ARG;
  if (hipSetDevice(p->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  for (uint64_t i = 0; i < n_idx; ++i) if (indices[i] >= n) return ZKWG_RC_BAD_ARG;
  int rc = ZKWG_RC_OK;
  size_t k = 0;
  for (uint64_t first = 0; first < n_idx && rc == ZKWG_RC_OK; first += p->E, ++k) {
    ZkProveCtx& s = p->ctx[k % p->ctx.size()];
    rc = prover_finish(p, s, blinding, out_proofs);               // the series this context ran before (others are still in flight)
    if (rc == ZKWG_RC_OK) rc = prover_enqueue(p, s, d_in, n, d_scratch, indices, first, std::min<uint64_t>(p->E, n_idx - first));
  }
  for (size_t j = 0; j < p->ctx.size(); ++j) {                    // drain, oldest first
    const int r2 = prover_finish(p, p->ctx[(k + j) % p->ctx.size()], blinding, out_proofs);
    if (rc == ZKWG_RC_OK) rc = r2;
  }
  if (rc != ZKWG_RC_OK) { hipDeviceSynchronize(); for (ZkProveCtx& s : p->ctx) s.busy = false; (void)hipGetLastError(); }
  return rc;
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
