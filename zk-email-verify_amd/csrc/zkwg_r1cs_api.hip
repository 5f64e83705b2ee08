// C-ABI of the `.r1cs` loader / device constraint check (include/zkwg.h, zkwg_r1cs_*).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include "../../include/zkwg.h"
#include "zkwg_kernels.h"
#include "zkwg_r1cs.h"

struct zkwg_r1cs {
  ZkR1csHost h;
  int device = -1;
  u64* d_row = nullptr;
  u32* d_wire = nullptr;
  Fr* d_coef = nullptr;
  u8* d_kind = nullptr;
};

extern "C" {

int zkwg_r1cs_load(const uint8_t* bytes, uint64_t len, int device, zkwg_r1cs_t** out) {
  if (!bytes || !out) return ZKWG_RC_BAD_ARG;
  zkwg_r1cs* r = nullptr;
  try {   // no C++ exception may cross the C ABI
    r = new zkwg_r1cs();
    if (!zk_r1cs_parse(bytes, len, r->h)) { delete r; return ZKWG_RC_BAD_CONFIG; }
  } catch (const std::bad_alloc&) {
    delete r;
    return ZKWG_RC_OOM;
  } catch (const std::exception&) {
    delete r;
    return ZKWG_RC_BAD_CONFIG;
  }
  if (device >= 0) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) { delete r; return ZKWG_RC_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { delete r; return ZKWG_RC_HIP_ERROR; }
    r->device = device;
    const ZkR1csHost& h = r->h;
    const size_t nnz = std::max<size_t>(h.wire.size(), 1);
    bool ok = hipMalloc((void**)&r->d_row, h.row_ptr.size() * 8) == hipSuccess &&
              hipMalloc((void**)&r->d_wire, nnz * 4) == hipSuccess &&
              hipMalloc((void**)&r->d_coef, nnz * sizeof(Fr)) == hipSuccess &&
              hipMalloc((void**)&r->d_kind, nnz) == hipSuccess;
    ok = ok && hipMemcpy(r->d_row, h.row_ptr.data(), h.row_ptr.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(r->d_wire, h.wire.data(), h.wire.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(r->d_coef, h.coef.data(), h.coef.size() * sizeof(Fr), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(r->d_kind, h.kind.data(), h.kind.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { hipFree(r->d_row); hipFree(r->d_wire); hipFree(r->d_coef); hipFree(r->d_kind); delete r; return ZKWG_RC_OOM; }
  }
  *out = r;
  return ZKWG_RC_OK;
}

void zkwg_r1cs_destroy(zkwg_r1cs_t* r) {
  if (!r) return;
  if (r->device >= 0) { hipSetDevice(r->device); hipFree(r->d_row); hipFree(r->d_wire); hipFree(r->d_coef); hipFree(r->d_kind); }
  delete r;
}

int zkwg_r1cs_info(const zkwg_r1cs_t* r, uint64_t out[6]) {
  if (!r || !out) return ZKWG_RC_BAD_ARG;
  out[0] = r->h.n_wires; out[1] = r->h.n_pub_out; out[2] = r->h.n_pub_in; out[3] = r->h.n_prv_in;
  out[4] = r->h.n_constraints; out[5] = r->h.n_labels;
  return ZKWG_RC_OK;
}

int zkwg_check_constraints_device(zkwg_r1cs_t* r, const void* d_witness, uint64_t n, uint64_t stride,
                                  void* d_first_bad, void* hip_stream) {
  if (!r || !d_witness || !d_first_bad) return ZKWG_RC_BAD_ARG;
  if (r->device < 0) return ZKWG_RC_NO_DEVICE;
  if (n == 0) return ZKWG_RC_OK;
  if (stride < 32ull * r->h.n_wires || (stride & 15)) return ZKWG_RC_BAD_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  if (hipMemsetAsync(d_first_bad, 0xff, n * 8, st) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  const u32 m = r->h.n_constraints;
  for (u64 lo = 0; m && lo < n; lo += 32768) {   // grid.y is limited to 65535
    const u32 cnt = (u32)std::min<u64>(32768, n - lo);
    hipLaunchKernelGGL(zk_r1cs_check, dim3((m + 255) / 256, cnt), dim3(256), 0, st, r->d_row, r->d_wire, r->d_coef,
                       r->d_kind, m, (const u8*)d_witness + lo * stride, stride, (unsigned long long*)d_first_bad + lo);
  }
  return hipGetLastError() == hipSuccess ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}

int zkwg_r1cs_evaluate_device(zkwg_r1cs_t* r, const void* d_witness, uint64_t n, uint64_t stride, int montgomery,
                              void* d_abc, uint64_t abc_stride, void* hip_stream) {
  if (!r || !d_witness || !d_abc) return ZKWG_RC_BAD_ARG;
  if (r->device < 0) return ZKWG_RC_NO_DEVICE;
  if (n == 0) return ZKWG_RC_OK;
  const u32 m = r->h.n_constraints;
  if (stride < 32ull * r->h.n_wires || (stride & 15) || abc_stride < 96ull * m || (abc_stride & 15)) return ZKWG_RC_BAD_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  for (u64 lo = 0; m && lo < n; lo += 32768) {   // grid.y is limited to 65535
    const u32 cnt = (u32)std::min<u64>(32768, n - lo);
    hipLaunchKernelGGL(zk_r1cs_eval, dim3((u32)((3ull * m + 255) / 256), cnt), dim3(256), 0, st, r->d_row, r->d_wire, r->d_coef,
                       r->d_kind, m, (const u8*)d_witness + lo * stride, stride, (u8*)d_abc + lo * abc_stride, abc_stride, montgomery ? 1 : 0);
  }
  return hipGetLastError() == hipSuccess ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}

int zkwg_check_constraints(zkwg_r1cs_t* r, const uint8_t* witness, uint64_t n, uint64_t stride, uint64_t* first_bad) {
  if (!r || !witness || !first_bad) return ZKWG_RC_BAD_ARG;
  if (r->device < 0) return ZKWG_RC_NO_DEVICE;
  if (hipSetDevice(r->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  const u64 tile = std::max<u64>(1, std::min<u64>(n, (1ull << 31) / std::max<u64>(stride, 1)));   // <= 2 GiB staged at a time
  u8* d_w = nullptr; u64* d_bad = nullptr;
  if (hipMalloc((void**)&d_w, tile * stride) != hipSuccess || hipMalloc((void**)&d_bad, tile * 8) != hipSuccess) {
    hipFree(d_w); hipFree(d_bad);
    return ZKWG_RC_OOM;
  }
  int rc = ZKWG_RC_OK;
  for (u64 lo = 0; lo < n && rc == ZKWG_RC_OK; lo += tile) {
    const u64 cnt = std::min(tile, n - lo);
    if (hipMemcpy(d_w, witness + lo * stride, cnt * stride, hipMemcpyHostToDevice) != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; break; }
    rc = zkwg_check_constraints_device(r, d_w, cnt, stride, d_bad, nullptr);
    if (rc == ZKWG_RC_OK && hipMemcpy(first_bad + lo, d_bad, cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = ZKWG_RC_HIP_ERROR;
  }
  hipFree(d_w); hipFree(d_bad);
  return rc;
}

// ---- prover hand-off: standard <-> Montgomery form of device-resident field elements
int zkwg_convert_montgomery_device(void* d_values, uint64_t n_values, int to_montgomery, void* hip_stream) {
  if (!d_values || ((uintptr_t)d_values & 15)) return ZKWG_RC_BAD_ARG;
  if (n_values == 0) return ZKWG_RC_OK;
  if (n_values > 256ull * 0x7fffffffull) return ZKWG_RC_BAD_ARG;
  hipLaunchKernelGGL(zk_mont_convert, dim3((u32)((n_values + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     (Fr*)d_values, (u64)n_values, to_montgomery ? 1 : 0);
  return hipGetLastError() == hipSuccess ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}

}  // extern "C"
