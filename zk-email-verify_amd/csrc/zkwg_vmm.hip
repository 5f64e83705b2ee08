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
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/zkwg.h"

namespace {
struct ZkChunked { int device; size_t bytes, chunk; std::vector<hipMemGenericAllocationHandle_t> handles; };
std::mutex g_mu;
std::map<void*, ZkChunked> g_live;
}

// the store shape of zk_expand: 32 KiB per 256-thread workgroup, each XCD one contiguous eighth of the launch
__global__ __launch_bounds__(256) void zk_vmm_probe_fill(uint4* dst, unsigned cpw) {
  unsigned p = blockIdx.x;
  const unsigned per = gridDim.x >> 3;
  if (p < per * 8u) p = (p & 7u) * per + (p >> 3);
  uint4* d = dst + (unsigned long long)p * cpw;
  const uint4 v = make_uint4(0u, 0u, 0u, 0u);
  for (unsigned c = threadIdx.x; c < cpw; c += 256) d[c] = v;
}

// `extra` > 0: that many more physical chunks than needed are created, each is mapped on its own and takes the probe fill above, the
// fastest are kept (mapped in one range, in the order they were created) and the others released -- selection at chunk granularity
// with a transient of `extra` chunks.  rates (GB/s per candidate chunk, creation order; may be NULL) / n_rates report what was seen.
extern "C" int zkwg_device_alloc_chunked_ex(int device, uint64_t bytes, uint64_t chunk_bytes, uint32_t extra, void** out, float* rates, uint32_t cap,
                                            uint32_t* n_rates) {
  if (!out || bytes == 0) return ZKWG_RC_BAD_ARG;
  if (n_rates) *n_rates = 0;
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
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> cand;
  auto release_all = [&]() { for (auto h : cand) hipMemRelease(h); cand.clear(); (void)hipGetLastError(); };
  for (size_t k = 0; k < n + extra; ++k) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    cand.push_back(h);
  }
  if (cand.size() < n) { release_all(); return ZKWG_RC_OOM; }
  std::vector<size_t> keep(n);
  for (size_t k = 0; k < n; ++k) keep[k] = k;
  if (cand.size() > n && chunk >= (1u << 20)) {
    // probe every candidate on its own mapping
    void* va = nullptr;
    std::vector<float> ms(cand.size(), 0.f);
    bool ok = hipMemAddressReserve(&va, chunk, 0, nullptr, 0) == hipSuccess;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned cpw = 2048, np = (unsigned)(chunk / 16 / cpw);
    for (size_t k = 0; k < cand.size() && ok; ++k) {
      ok = hipMemMap(va, chunk, 0, cand[k], 0) == hipSuccess && hipMemSetAccess(va, chunk, &acc, 1) == hipSuccess;
      if (!ok) break;
      hipLaunchKernelGGL(zk_vmm_probe_fill, dim3(np), dim3(256), 0, 0, (uint4*)va, cpw);      // first touch
      hipEventRecord(e0, 0);
      for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(zk_vmm_probe_fill, dim3(np), dim3(256), 0, 0, (uint4*)va, cpw);
      hipEventRecord(e1, 0);
      ok = hipEventSynchronize(e1) == hipSuccess;
      hipEventElapsedTime(&ms[k], e0, e1);
      ms[k] *= 0.25f;
      ok = hipMemUnmap(va, chunk) == hipSuccess && ok;
      if (rates && k < cap) rates[k] = ms[k] > 0.f ? (float)(chunk / (ms[k] * 1e6)) : 0.f;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (va) hipMemAddressFree(va, chunk);
    if (!ok) { release_all(); return ZKWG_RC_HIP_ERROR; }
    if (n_rates) *n_rates = (uint32_t)std::min<size_t>(cand.size(), cap);
    std::vector<size_t> order(cand.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return ms[x] < ms[y]; });
    order.resize(n);
    std::sort(order.begin(), order.end());
    keep = order;
  }
  // release what is not kept, map the rest back to back
  std::vector<char> kept(cand.size(), 0);
  for (size_t k : keep) kept[k] = 1;
  ZkChunked rec{device, total, chunk, {}};
  for (size_t k = 0; k < cand.size(); ++k) { if (kept[k]) rec.handles.push_back(cand[k]); else hipMemRelease(cand[k]); }
  void* base = nullptr;
  if (hipMemAddressReserve(&base, total, 0, nullptr, 0) != hipSuccess) { for (auto h : rec.handles) hipMemRelease(h); return ZKWG_RC_OOM; }
  bool ok = true;
  size_t mapped = 0;
  for (; mapped < n && ok; ++mapped) ok = hipMemMap((char*)base + mapped * chunk, chunk, 0, rec.handles[mapped], 0) == hipSuccess;
  if (ok && hipMemSetAccess(base, total, &acc, 1) != hipSuccess) ok = false;
  if (!ok) {
    for (size_t k = 0; k + (ok ? 0 : 1) <= mapped && k < n; ++k) hipMemUnmap((char*)base + k * chunk, chunk);
    for (auto h : rec.handles) hipMemRelease(h);
    hipMemAddressFree(base, total);
    (void)hipGetLastError();
    return ZKWG_RC_OOM;
  }
  { std::lock_guard<std::mutex> g(g_mu); g_live[base] = std::move(rec); }
  *out = base;
  return ZKWG_RC_OK;
}
extern "C" int zkwg_device_alloc_chunked(int device, uint64_t bytes, uint64_t chunk_bytes, void** out) {
  return zkwg_device_alloc_chunked_ex(device, bytes, chunk_bytes, 0, out, nullptr, 0, nullptr);
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
