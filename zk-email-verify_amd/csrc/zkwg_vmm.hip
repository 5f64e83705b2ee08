// Device buffers built from physical chunks with the HIP virtual-memory API: where the output ring of zk_expand lives.
//
// Round 4 found that the rate at which a buffer takes zk_expand's stores depends on the allocation: of seven 29 GB buffers from
// hipMalloc two or three took the same kernel 12-20 % slower than the others, reproducibly per buffer, and
// worked around it by allocating spare candidates and timing the kernel into each (203 GB of transient allocations at set-up).
// tools/chunkbench.hip (profiles/r05/r05_e_chunkbench.txt) shows the cause is not WHERE in HBM the memory lies: 96 physical chunks of
// 1 GiB from hipMemCreate take a 32 KiB-per-workgroup store stream at 6.33-6.57 TB/s each (3.8 % spread), and ranges mapped from the 58
// first, the 58 fastest or the 58 slowest chunks all take it at 7.2-7.3 TB/s.  A ring mapped chunk by chunk has no slow tiles, needs no
// candidates and no transient memory, and works in a process that fills HBM.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/zkwg.h"

namespace {
struct ZkChunked { int device; size_t bytes, chunk; std::vector<hipMemGenericAllocationHandle_t> handles; };
std::mutex g_mu;
std::map<void*, ZkChunked> g_live;
}

extern "C" int zkwg_device_alloc_chunked(int device, uint64_t bytes, uint64_t chunk_bytes, void** out) {
  if (!out || bytes == 0) return ZKWG_RC_BAD_ARG;
  if (device < 0) return ZKWG_RC_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return ZKWG_RC_HIP_ERROR;
  size_t chunk = chunk_bytes ? (size_t)chunk_bytes : ((size_t)1 << 30);
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t n = ((size_t)bytes + chunk - 1) / chunk, total = n * chunk;
  void* base = nullptr;
  if (hipMemAddressReserve(&base, total, 0, nullptr, 0) != hipSuccess) return ZKWG_RC_OOM;
  ZkChunked rec{device, total, chunk, {}};
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  bool ok = true;
  for (size_t k = 0; k < n && ok; ++k) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
    rec.handles.push_back(h);
    if (hipMemMap((char*)base + k * chunk, chunk, 0, h, 0) != hipSuccess) { ok = false; break; }
  }
  if (ok && hipMemSetAccess(base, total, &acc, 1) != hipSuccess) ok = false;
  if (!ok) {
    for (size_t k = 0; k < rec.handles.size(); ++k) { hipMemUnmap((char*)base + k * chunk, chunk); hipMemRelease(rec.handles[k]); }
    hipMemAddressFree(base, total);
    (void)hipGetLastError();
    return ZKWG_RC_OOM;
  }
  { std::lock_guard<std::mutex> g(g_mu); g_live[base] = std::move(rec); }
  *out = base;
  return ZKWG_RC_OK;
}
extern "C" int zkwg_device_free_chunked(void* ptr) {
  if (!ptr) return ZKWG_RC_OK;
  ZkChunked rec;
  {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_live.find(ptr);
    if (it == g_live.end()) return ZKWG_RC_BAD_ARG;
    rec = std::move(it->second);
    g_live.erase(it);
  }
  if (hipSetDevice(rec.device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  hipDeviceSynchronize();
  for (size_t k = 0; k < rec.handles.size(); ++k) { hipMemUnmap((char*)ptr + k * rec.chunk, rec.chunk); hipMemRelease(rec.handles[k]); }
  hipMemAddressFree(ptr, rec.bytes);
  return ZKWG_RC_OK;
}
