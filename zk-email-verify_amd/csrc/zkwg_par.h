// Host-side helper: run fn(chunk, n_chunks) on a few threads (circuit construction from large `.sym` / `.r1cs` files:
// parsing hundreds of megabytes is memory-bound per core).  ZKWG_HOST_THREADS overrides the thread count.
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <thread>
#include <vector>

static inline unsigned zk_host_threads() {
  if (const char* v = getenv("ZKWG_HOST_THREADS")) { const int n = atoi(v); if (n > 0) return (unsigned)std::min(n, 64); }
  unsigned n = std::thread::hardware_concurrency();
  if (n == 0) n = 4;
  return std::min(n, 16u);
}
template <class F>
static inline void zk_parallel_chunks(unsigned n_chunks, F fn) {
  if (n_chunks <= 1) { fn(0u, 1u); return; }
  std::vector<std::thread> th;
  th.reserve(n_chunks - 1);
  for (unsigned i = 1; i < n_chunks; ++i) th.emplace_back([&fn, i, n_chunks] { fn(i, n_chunks); });
  fn(0u, n_chunks);
  for (auto& t : th) t.join();
}
