// Batched BN254-Fr number-theoretic transforms for the step that follows A.w | B.w | C.w in `snarkjs.groth16.prove`
// (reference call site: packages/helpers/src/chunked-zkey.ts:80-84; SURVEY.md 8f4 "hand-off into the prover"): per proof
// three inverse transforms, the coset shift, three forward transforms and a(x) b(x) - c(x) on a domain of 2^20 .. 2^22
// points -- 6 x 2^L x L / 2 butterflies of one Montgomery product each.  Unlike everything else on the path this is
// ARITHMETIC-bound: ~70 M products per EmailVerifier(576,192) proof against ~0.8 GB of HBM traffic; the roofline is the
// issue rate of v_mad_u64_u32 (a Montgomery product = 128 of them: ~77 G products/s per MI355X), not HBM.  No MFMA: the
// products are 254-bit modular integers.
//
// Structure (all values in Montgomery form, in place in a work buffer of n = 2^L elements per polynomial):
//   * decimation in frequency (natural order in, bit-reversed out) for the inverse transforms, decimation in time
//     (bit-reversed in, natural out) for the forward ones: no permutation pass between them, and the coset scaling
//     inc^i / n is a table indexed by bit-reversed position, fused into the last inverse pass;
//   * each transform is a few passes over HBM ("four-step" recursion): COLUMN passes run 2^g-point sub-transforms (g <= 8)
//     on 1024 / 2^g neighbouring columns of a block at a time in LDS, followed (DIF) or preceded (DIT) by the block twiddle
//     w_N^(column * frequency); the last / first pass transforms contiguous 2^10-element rows in LDS;
//   * one workgroup = 1,024 elements = 32 KiB of LDS + its twiddles: 3 workgroups per CU, 2-3 wavefronts per SIMD so that
//     the dependent multiply-add chains of one wavefront hide behind another's; LDS holds the halves of an element in two
//     arrays (conflict-free 16-byte accesses) and two butterfly stages share one pass through it.
#include "zkwg_kernels.h"

#define ZK_NTT_TILE 1024u   // elements per workgroup: a column pass takes 1024 / 2^g neighbouring columns (>= 128 contiguous bytes per row access)

__device__ __forceinline__ u32 zk_bitrev(u32 x, u32 bits) { return bits ? (__brev(x) >> (32u - bits)) : 0u; }
// w^e for the transform's direction: tw[k] = w^k, k < n; the inverse direction reads w^(n - e)
__device__ __forceinline__ Fr zk_ntt_tw(const Fr* __restrict__ tw, u64 n, u64 e, bool inv) {
  e &= n - 1u;
  return tw[inv ? ((n - e) & (n - 1u)) : e];
}

// LDS layout: the two 16-byte halves of an element live in separate arrays (lo[i], hi[i]): a wavefront's ds_read_b128 of
// consecutive elements then covers each bank once -- with 32-byte elements every bank would be hit twice per 16 lanes.
struct ZkLdsFr {
  uint4* lo; uint4* hi;
  __device__ __forceinline__ Fr get(u32 i) const {
    const uint4 a = lo[i], b = hi[i];
    return Fr{{(u64)a.x | ((u64)a.y << 32), (u64)a.z | ((u64)a.w << 32), (u64)b.x | ((u64)b.y << 32), (u64)b.z | ((u64)b.w << 32)}};
  }
  __device__ __forceinline__ void put(u32 i, const Fr& v) const {
    lo[i] = make_uint4((u32)v.l[0], (u32)(v.l[0] >> 32), (u32)v.l[1], (u32)(v.l[1] >> 32));
    hi[i] = make_uint4((u32)v.l[2], (u32)(v.l[2] >> 32), (u32)v.l[3], (u32)(v.l[3] >> 32));
  }
};

// The butterfly stages of 2^g-point sub-transforms over `nel` elements per column, C columns interleaved (element i of
// column cc at i * C + cc), two stages per pass through LDS where possible: a thread loads the four elements of a radix-4
// group, runs both stages in registers (four products, as two radix-2 stages would) and stores them -- 12 LDS accesses per
// four products instead of 20, and half the barriers.
template <bool DIT>
__device__ __forceinline__ void zk_ntt_stages(const ZkLdsFr& y, const ZkLdsFr& twl, u32 G, u32 g, u32 nel, u32 C) {
  u32 st = 0;
  for (; st + 1u < g; st += 2u) {
    const u32 h = DIT ? (1u << st) : (G >> (st + 2u));
    for (u32 b = threadIdx.x; b < (nel / 4u) * C; b += 256u) {
      const u32 cc = b % C, q = b / C;
      const u32 p = q % h, i0 = (q / h) * 4u * h + p;
      const u32 e0 = i0 * C + cc, e1 = (i0 + h) * C + cc, e2 = (i0 + 2u * h) * C + cc, e3 = (i0 + 3u * h) * C + cc;
      const Fr x0 = y.get(e0), x1 = y.get(e1), x2 = y.get(e2), x3 = y.get(e3);
      if (DIT) {
        const Fr w1 = twl.get(p * (G / (2u * h)));
        const Fr t1 = fr_mont_mul(x1, w1), t3 = fr_mont_mul(x3, w1);
        const Fr a0 = fr_add(x0, t1), a1 = fr_sub(x0, t1), a2 = fr_add(x2, t3), a3 = fr_sub(x2, t3);
        const Fr u2 = fr_mont_mul(a2, twl.get(p * (G / (4u * h)))), u3 = fr_mont_mul(a3, twl.get((p + h) * (G / (4u * h))));
        y.put(e0, fr_add(a0, u2)); y.put(e2, fr_sub(a0, u2));
        y.put(e1, fr_add(a1, u3)); y.put(e3, fr_sub(a1, u3));
      } else {
        const Fr a0 = fr_add(x0, x2), a1 = fr_add(x1, x3);
        const Fr a2 = fr_mont_mul(fr_sub(x0, x2), twl.get(p << st)), a3 = fr_mont_mul(fr_sub(x1, x3), twl.get((p + h) << st));
        const Fr w2 = twl.get(p << (st + 1u));
        y.put(e0, fr_add(a0, a1)); y.put(e1, fr_mont_mul(fr_sub(a0, a1), w2));
        y.put(e2, fr_add(a2, a3)); y.put(e3, fr_mont_mul(fr_sub(a2, a3), w2));
      }
    }
    __syncthreads();
  }
  if (st < g) {
    const u32 half = DIT ? (1u << st) : (G >> (st + 1u));
    for (u32 b = threadIdx.x; b < (nel / 2u) * C; b += 256u) {
      const u32 cc = b % C, pi = b / C;
      const u32 i = (pi / half) * 2u * half + (pi % half), j = i + half;
      const u32 k = DIT ? (pi % half) * (G / (2u * half)) : ((pi % half) << st);
      const Fr a = y.get(i * C + cc), bb = y.get(j * C + cc);
      if (DIT) {
        const Fr tt = fr_mont_mul(bb, twl.get(k));
        y.put(i * C + cc, fr_add(a, tt));
        y.put(j * C + cc, fr_sub(a, tt));
      } else {
        y.put(i * C + cc, fr_add(a, bb));
        y.put(j * C + cc, fr_mont_mul(fr_sub(a, bb), twl.get(k)));
      }
    }
    __syncthreads();
  }
}

// One column pass.  Block size N = 2^lb (the sub-problem of this recursion level), sub-transform size G = 2^g over the rows
// r of a column: element index = block * N + r * (N >> g) + c.  DIF: sub-transform, then y *= w_N^(c * bitrev_g(r)).
// DIT: y *= w_N^(c * bitrev_g(r)) first, then the sub-transform.  `src` may differ from `dst` (first inverse pass: reads
// A.w | B.w | C.w, `valid` elements per polynomial, zero beyond) -- polynomial q = blockIdx.y: src + (q / 3) * src_es + (q % 3) * src_ps.
template <bool DIT>
__global__ __launch_bounds__(256) void zk_ntt_col(const Fr* src, u64 src_es, u64 src_ps, u64 valid, Fr* dst,   /* launched in place: src may alias dst */
                                                   const Fr* __restrict__ tw, u32 L, u32 lb, u32 g, u32 inv) {
  extern __shared__ uint4 lds4[];
  const u64 n = 1ull << L;
  const u32 G = 1u << g, C = ZK_NTT_TILE >> g;
  const ZkLdsFr y{lds4, lds4 + G * C};                           // [G][C] elements
  const ZkLdsFr twl{lds4 + 2u * G * C, lds4 + 2u * G * C + G / 2u};   // w_G^k, k < G / 2 (direction applied)
  const u32 cols_per_block = 1u << (lb - g);
  const u64 cid0 = (u64)blockIdx.x * C;
  const u64 block = cid0 >> (lb - g);
  const u32 c0 = (u32)(cid0 & (cols_per_block - 1u));
  const u64 q = blockIdx.y;
  const Fr* s = src + (q / 3u) * src_es + (q % 3u) * src_ps;
  Fr* d = dst + q * n;
  const u64 base = block << lb;
  const bool invb = inv != 0;
  for (u32 k = threadIdx.x; k < G / 2u; k += 256u) twl.put(k, zk_ntt_tw(tw, n, (u64)k << (L - g), invb));
  for (u32 t = threadIdx.x; t < G * C; t += 256u) {
    const u32 r = t / C, cc = t % C;
    const u64 idx = base + ((u64)r << (lb - g)) + c0 + cc;
    Fr v = idx < valid ? s[idx] : fr_zero();
    if (DIT) v = fr_mont_mul(v, zk_ntt_tw(tw, n, ((u64)(c0 + cc) * zk_bitrev(r, g)) << (L - lb), invb));
    y.put(t, v);
  }
  __syncthreads();
  zk_ntt_stages<DIT>(y, twl, G, g, G, C);
  for (u32 t = threadIdx.x; t < G * C; t += 256u) {
    const u32 r = t / C, cc = t % C;
    const u64 idx = base + ((u64)r << (lb - g)) + c0 + cc;
    Fr v = y.get(t);
    if (!DIT) v = fr_mont_mul(v, zk_ntt_tw(tw, n, ((u64)(c0 + cc) * zk_bitrev(r, g)) << (L - lb), invb));
    d[idx] = v;
  }
}

// The row pass: contiguous blocks of G = 2^g elements (g <= 10), 1,024 elements per workgroup.  DIF (inverse direction of the
// pipeline): optional multiplication by scale[position] on the way out (coset shift and 1 / n).  `src` / `valid` as above.
template <bool DIT>
__global__ __launch_bounds__(256) void zk_ntt_row(const Fr* src, u64 src_es, u64 src_ps, u64 valid, Fr* dst,   /* launched in place: src may alias dst */
                                                   const Fr* __restrict__ tw, const Fr* __restrict__ scale, Fr uni, u32 use_uni, u32 L, u32 g, u32 inv) {
  extern __shared__ uint4 lds4[];
  const u64 n = 1ull << L;
  const u32 G = 1u << g;
  const u32 TILE = n < 1024u ? (u32)n : 1024u;
  const ZkLdsFr y{lds4, lds4 + TILE};                                   // [TILE] elements
  const ZkLdsFr twl{lds4 + 2u * TILE, lds4 + 2u * TILE + G / 2u};      // w_G^k, k < G / 2
  const u64 q = blockIdx.y;
  const Fr* s = src + (q / 3u) * src_es + (q % 3u) * src_ps;
  Fr* d = dst + q * n;
  const u64 base = (u64)blockIdx.x * TILE;
  const bool invb = inv != 0;
  for (u32 k = threadIdx.x; k < G / 2u; k += 256u) twl.put(k, zk_ntt_tw(tw, n, (u64)k << (L - g), invb));
  for (u32 t = threadIdx.x; t < TILE; t += 256u) y.put(t, base + t < valid ? s[base + t] : fr_zero());
  __syncthreads();
  zk_ntt_stages<DIT>(y, twl, G, g, TILE, 1u);
  for (u32 t = threadIdx.x; t < TILE; t += 256u) {
    Fr v = y.get(t);
    if (scale) v = fr_mont_mul(v, scale[base + t]);
    else if (use_uni) v = fr_mont_mul(v, uni);     // (stand-alone inverse transform: 1 / n)
    d[base + t] = v;
  }
}

// out[k] = a[k] b[k] - c[k]  (joinABC of groth16_prove.js), polynomials of email e at work + (3 e + {0, 1, 2}) n
__global__ __launch_bounds__(256) void zk_ntt_join(const Fr* __restrict__ work, Fr* __restrict__ out, u64 n, u64 out_es) {
  const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 e = blockIdx.y;
  const Fr* w = work + 3u * e * n;
  out[e * out_es + i] = fr_sub(fr_mont_mul(w[i], w[n + i]), w[2u * n + i]);
}
// in-place bit-reversal permutation of n_polys arrays of 2^L elements (stand-alone transforms only: the pipeline needs none)
__global__ __launch_bounds__(256) void zk_ntt_bitrev(Fr* __restrict__ data, u32 L) {
  const u64 n = 1ull << L;
  const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 j = (u64)zk_bitrev((u32)i, L);
  if (i < j) {
    Fr* p = data + (u64)blockIdx.y * n;
    const Fr a = p[i], b = p[j];
    p[i] = b; p[j] = a;
  }
}

// ---- launch helpers (called from zkwg_ntt_api.hip) ---------------------------------------------------------------
// passes of one transform: column passes (block size 2^lb, 2^g-point sub-transforms), then the row pass (DIF), or the
// reverse (DIT)
extern "C" int zk_ntt_launch(int dit, const Fr* src, u64 src_es, u64 src_ps, u64 valid, Fr* work, const Fr* tw, const Fr* scale, const Fr* uni_host,
                             u32 L, u32 n_polys, u32 inv, hipStream_t st) {
  const Fr uni = uni_host ? *uni_host : fr_zero();
  const u32 use_uni = uni_host ? 1u : 0u;
  const u32 g_row = L < 10u ? L : 10u;
  u32 gs[8], ng = 0;
  {
    const u32 R = L - g_row;
    const u32 np = (R + 7u) / 8u;
    for (u32 i = 0; i < np; ++i) gs[ng++] = R / np + (i < R % np ? 1u : 0u);
  }
  const u64 n = 1ull << L;
  const u32 tile = n < 1024u ? (u32)n : 1024u;
  const size_t row_lds = (tile + (1u << g_row) / 2u) * sizeof(Fr);
  const dim3 rgrid((u32)(n / tile), n_polys);
  if (!dit) {
    const Fr* s = src; u64 es = src_es, ps = src_ps, v = valid;
    u32 lb = L;
    for (u32 i = 0; i < ng; ++i) {
      const u32 g = gs[i];
      const size_t lds = (ZK_NTT_TILE + (1u << g) / 2u) * sizeof(Fr);
      hipLaunchKernelGGL((zk_ntt_col<false>), dim3((u32)(n / ZK_NTT_TILE), n_polys), dim3(256), lds, st, s, es, ps, v, work, tw, L, lb, g, inv);
      s = work; es = 3u * n; ps = n; v = n;
      lb -= g;
    }
    hipLaunchKernelGGL((zk_ntt_row<false>), rgrid, dim3(256), row_lds, st, s, es, ps, v, work, tw, scale, uni, use_uni, L, g_row, inv);
  } else {
    hipLaunchKernelGGL((zk_ntt_row<true>), rgrid, dim3(256), row_lds, st, src, src_es, src_ps, valid, work, tw, (const Fr*)nullptr, uni, 0u, L, g_row, inv);
    u32 lb = g_row;
    for (u32 i = ng; i-- > 0;) {
      const u32 g = gs[i];
      lb += g;
      const size_t lds = (ZK_NTT_TILE + (1u << g) / 2u) * sizeof(Fr);
      hipLaunchKernelGGL((zk_ntt_col<true>), dim3((u32)(n / ZK_NTT_TILE), n_polys), dim3(256), lds, st, (const Fr*)work, 3u * n, n, n, work, tw, L, lb, g, inv);
    }
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int zk_ntt_join_launch(const Fr* work, Fr* out, u64 n, u64 out_es, u32 n_emails, hipStream_t st) {
  hipLaunchKernelGGL(zk_ntt_join, dim3((u32)((n + 255u) / 256u), n_emails), dim3(256), 0, st, work, out, n, out_es);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int zk_ntt_bitrev_launch(Fr* data, u32 L, u32 n_polys, hipStream_t st) {
  hipLaunchKernelGGL(zk_ntt_bitrev, dim3((u32)(((1ull << L) + 255u) / 256u), n_polys), dim3(256), 0, st, data, L);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
