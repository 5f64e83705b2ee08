// Which hipMemcpy forms accept memory mapped with the virtual-memory API (csrc/zkwg_vmm.hip)?  (diagnostic for the resident pipeline's ring)
//   hipcc -O3 --offload-arch=gfx950 tools/vmm_probe.hip -o tools/vmm_probe && tools/vmm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define TRY(what, x) do { hipError_t e = (x); printf("%-62s %s\n", what, e == hipSuccess ? "ok" : hipGetErrorString(e)); (void)hipGetLastError(); } while (0)
int main() {
  hipSetDevice(0);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  const size_t chunk = 1ull << 30, n = 3;
  void* base = nullptr;
  TRY("hipMemAddressReserve 3 GiB", hipMemAddressReserve(&base, n * chunk, 0, nullptr, 0));
  hipMemGenericAllocationHandle_t h[3];
  for (size_t k = 0; k < n; ++k) { TRY("hipMemCreate 1 GiB", hipMemCreate(&h[k], chunk, &prop, 0)); TRY("hipMemMap", hipMemMap((char*)base + k * chunk, chunk, 0, h[k], 0)); }
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  TRY("hipMemSetAccess", hipMemSetAccess(base, n * chunk, &acc, 1));
  TRY("hipMemset inside one chunk", hipMemset(base, 7, 1 << 20));
  TRY("hipMemset across a chunk boundary", hipMemset((char*)base + chunk - 4096, 9, 8192));
  std::vector<char> host(1 << 20);
  TRY("hipMemcpy D2H inside one chunk", hipMemcpy(host.data(), base, host.size(), hipMemcpyDeviceToHost));
  printf("   first byte %d\n", host[0]);
  TRY("hipMemcpy D2H across a chunk boundary", hipMemcpy(host.data(), (char*)base + chunk - 4096, 8192, hipMemcpyDeviceToHost));
  printf("   bytes %d %d\n", host[0], host[8191]);
  void* plain = nullptr; hipMalloc(&plain, 1 << 20);
  TRY("hipMemcpy D2D vmm -> hipMalloc", hipMemcpy(plain, base, 1 << 20, hipMemcpyDeviceToDevice));
  TRY("hipMemcpy2DAsync D2D strided vmm -> hipMalloc", hipMemcpy2DAsync(plain, 96, (char*)base + 32, 1 << 16, 96, 8, hipMemcpyDeviceToDevice, 0));
  TRY("hipStreamSynchronize", hipStreamSynchronize(0));
  TRY("hipMemcpy2DAsync D2D strided across the boundary", hipMemcpy2DAsync(plain, 96, (char*)base + chunk - (1 << 17) + 32, 1 << 16, 96, 8, hipMemcpyDeviceToDevice, 0));
  TRY("hipStreamSynchronize", hipStreamSynchronize(0));
  TRY("hipMemcpyAsync D2H (pageable) from vmm", hipMemcpyAsync(host.data(), base, 1 << 20, hipMemcpyDeviceToHost, 0));
  TRY("hipStreamSynchronize", hipStreamSynchronize(0));
  void* pinned = nullptr; hipHostMalloc(&pinned, 1 << 20, 0);
  TRY("hipMemcpyAsync D2H (pinned) from vmm", hipMemcpyAsync(pinned, base, 1 << 20, hipMemcpyDeviceToHost, 0));
  TRY("hipStreamSynchronize", hipStreamSynchronize(0));
  return 0;
}
