// A 64-lane wavefront on the host (TEST INFRASTRUCTURE): the lanes are fibers (ucontext) of one thread, run in lane
// order; a cross-lane operation (ballot, readlane, shuffles, the workgroup barrier of a one-wavefront workgroup) is an
// exchange point -- every lane deposits a 64-bit value and yields, and when it resumes all 64 values of that exchange are
// there.  With it csrc/zkwg_rsa_wave.h, the RSA path the device actually runs, compiles and runs under g++ unchanged
// (tests/native/wavetest.cpp), so the shipped algorithm is checked against the oracle without a GPU.
// Valid for convergent code only: every lane must reach the same sequence of exchange points (the same rule the device
// code lives by for its ballots and shuffles).
#pragma once
#include <stdint.h>
#include <ucontext.h>
#include <functional>
#include <vector>

namespace wavesim {
struct Wave {
  ucontext_t main, ctx[64];
  std::vector<char> stacks;
  int cur = 0;
  bool done[64];
  uint64_t buf[2][64];
  uint32_t count[64];      // exchanges each lane has passed
  uint64_t exchanges = 0;
  std::function<void()> body;
};
inline Wave*& current() { static thread_local Wave* w = nullptr; return w; }
inline unsigned lane() { return (unsigned)current()->cur; }
// deposit `mine`, let the other lanes reach the same point, return the 64 deposited values
inline const uint64_t* exchange(uint64_t mine) {
  Wave& w = *current();
  const int l = w.cur;
  const uint32_t n = w.count[l]++;
  w.buf[n & 1][l] = mine;
  if (l == 0) ++w.exchanges;
  swapcontext(&w.ctx[l], &w.main);
  return w.buf[n & 1];
}
inline void trampoline() {
  Wave& w = *current();
  w.body();
  w.done[w.cur] = true;
  swapcontext(&w.ctx[w.cur], &w.main);
}
// run `body` once per lane; returns the number of exchange points
inline uint64_t run(std::function<void()> body, size_t stack_bytes = 1u << 20) {
  Wave w;
  w.body = std::move(body);
  w.stacks.resize(64 * stack_bytes);
  Wave* prev = current();
  current() = &w;
  for (int l = 0; l < 64; ++l) {
    w.done[l] = false; w.count[l] = 0;
    getcontext(&w.ctx[l]);
    w.ctx[l].uc_stack.ss_sp = w.stacks.data() + (size_t)l * stack_bytes;
    w.ctx[l].uc_stack.ss_size = stack_bytes;
    w.ctx[l].uc_link = &w.main;
    makecontext(&w.ctx[l], (void (*)())trampoline, 0);
  }
  for (;;) {
    bool any = false;
    for (int l = 0; l < 64; ++l) {
      if (w.done[l]) continue;
      any = true;
      w.cur = l;
      swapcontext(&w.main, &w.ctx[l]);
    }
    if (!any) break;
  }
  current() = prev;
  return w.exchanges;
}
}  // namespace wavesim

// the HIP spellings csrc/zkwg_rsa_wave.h uses
inline unsigned zk_wavesim_lane() { return wavesim::lane(); }
inline void zk_wavesim_sync() { wavesim::exchange(0); }
inline uint64_t __ballot(bool p) {
  const uint64_t* s = wavesim::exchange(p ? 1u : 0u);
  uint64_t m = 0;
  for (int i = 0; i < 64; ++i) m |= (s[i] & 1u) << i;
  return m;
}
inline int __builtin_amdgcn_readlane(int v, int l) { return (int)(uint32_t)wavesim::exchange((uint32_t)v)[l & 63]; }
inline int __shfl(int v, int src) { return (int)(uint32_t)wavesim::exchange((uint32_t)v)[src & 63]; }
inline int __shfl_up(int v, unsigned d) {
  const unsigned l = wavesim::lane();
  const uint64_t* s = wavesim::exchange((uint32_t)v);
  return l >= d ? (int)(uint32_t)s[l - d] : v;
}
inline int __shfl_down(int v, unsigned d) {
  const unsigned l = wavesim::lane();
  const uint64_t* s = wavesim::exchange((uint32_t)v);
  return l + d < 64 ? (int)(uint32_t)s[l + d] : v;
}
