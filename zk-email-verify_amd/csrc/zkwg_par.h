// Host-side helper: run fn(chunk, n_chunks) on a few threads (circuit construction from large `.sym` / `.r1cs` files:
// parsing hundreds of megabytes is memory-bound per core).  ZKWG_HOST_THREADS overrides the thread count.
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <exception>
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
  // an exception in a worker (bad_alloc on a 0.9 GB .sym) must reach the caller's catch, not std::terminate: every worker catches,
  // every thread is joined, the first exception is rethrown afterwards
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err(n_chunks);
  th.reserve(n_chunks - 1);
  auto guarded = [&fn, &err, n_chunks](unsigned i) { try { fn(i, n_chunks); } catch (...) { err[i] = std::current_exception(); } };
  try {
    for (unsigned i = 1; i < n_chunks; ++i) th.emplace_back(guarded, i);
  } catch (...) { err[0] = std::current_exception(); }     // (thread creation failed: run what was started, then report)
  if (!err[0]) guarded(0u);
  for (auto& t : th) t.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
}
