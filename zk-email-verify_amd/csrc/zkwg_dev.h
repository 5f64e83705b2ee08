// Device-side helpers shared by the gfx950 kernels: witness stores and native SHA-256.
#pragma once
#include <hip/hip_runtime.h>
#include "zkwg_sched.h"

// One witness slot = one 32-byte little-endian field element = two 16-byte chunks.
// All kernels address the witness in uint4 (16-byte) chunks so that a wavefront's
// 64 lanes cover 1 KiB of contiguous HBM per store instruction.
typedef unsigned int zk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void zk_st16(uint4* p, uint4 v) {
  zk_u32x4 x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<zk_u32x4*>(p));
}

__host__ __device__ __forceinline__ uint4 zk_small(u32 v) { return make_uint4(v, 0u, 0u, 0u); }
__host__ __device__ __forceinline__ uint4 zk_zero4() { return make_uint4(0u, 0u, 0u, 0u); }
__device__ __forceinline__ uint4 zk_fr_half(const Fr& a, u32 half) {
  u64 x = a.l[half * 2], y = a.l[half * 2 + 1];
  return make_uint4((u32)x, (u32)(x >> 32), (u32)y, (u32)(y >> 32));
}

// Workgroup-cooperative emit of `nslots` consecutive slots starting at `base`:
// thread t writes chunk t, t+blockDim, ... ; f(slot, half) -> uint4.
template <class F>
__device__ __forceinline__ void zk_emit(uint4* wit, u64 base, u32 nslots, F f) {
  uint4* dst = wit + base * 2;
  for (u32 c = threadIdx.x; c < 2u * nslots; c += blockDim.x) zk_st16(dst + c, f(c >> 1, c & 1u));
}
// Small-integer variant: f(slot) -> u32 (high 28 bytes are zero).
template <class F>
__device__ __forceinline__ void zk_emit_small(uint4* wit, u64 base, u32 nslots, F f) {
  uint4* dst = wit + base * 2;
  for (u32 c = threadIdx.x; c < 2u * nslots; c += blockDim.x) {
    uint4 v = zk_zero4();
    if ((c & 1u) == 0) v.x = f(c >> 1);
    zk_st16(dst + c, v);
  }
}

// ---------------------------------------------------------------- native SHA-256
__device__ __forceinline__ u32 zk_rotr(u32 x, int r) { return __builtin_rotateright32(x, r); }
__device__ __forceinline__ u32 zk_ldbe32(const u8* p) {
  return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
}

static __constant__ u32 ZK_K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ void zk_sha256_iv(u32* h) {
  h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
  h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
}

// 16 big-endian message words of a 64-byte block; 16-byte loads when the block is 16-byte aligned
__device__ __forceinline__ void zk_load_block_be(u32* w, const u8* blk) {
  if ((((uintptr_t)blk) & 15u) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint4 q = *(const uint4*)(blk + 16 * i);
      w[4 * i + 0] = __builtin_bswap32(q.x); w[4 * i + 1] = __builtin_bswap32(q.y);
      w[4 * i + 2] = __builtin_bswap32(q.z); w[4 * i + 3] = __builtin_bswap32(q.w);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = zk_ldbe32(blk + 4 * i);
  }
}

// Plain compression (no trace): st <- compress(st, block bytes)
__device__ inline void zk_sha256_compress(u32* st, const u8* blk) {
  u32 w[16];
  zk_load_block_be(w, blk);
  u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int t = 0; t < 64; ++t) {
    if (t >= 16) {
      u32 x2 = w[(t - 2) & 15], x15 = w[(t - 15) & 15];
      u32 s1 = zk_rotr(x2, 17) ^ zk_rotr(x2, 19) ^ (x2 >> 10);
      u32 s0 = zk_rotr(x15, 7) ^ zk_rotr(x15, 18) ^ (x15 >> 3);
      w[t & 15] = s1 + w[(t - 7) & 15] + s0 + w[(t - 16) & 15];
    }
    u32 t1 = h + (zk_rotr(e, 6) ^ zk_rotr(e, 11) ^ zk_rotr(e, 25)) + ((e & f) ^ (~e & g)) + ZK_K256[t] + w[t & 15];
    u32 t2 = (zk_rotr(a, 2) ^ zk_rotr(a, 13) ^ zk_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
