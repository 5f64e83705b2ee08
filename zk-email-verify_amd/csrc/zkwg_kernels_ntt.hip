// Batched BN254-Fr number-theoretic transforms for the step that follows A.w | B.w | C.w in `snarkjs.groth16.prove`
// (reference call site: packages/helpers/src/chunked-zkey.ts:80-84; SURVEY.md 8f4 "hand-off into the prover"): per proof
// three inverse transforms, the coset shift, three forward transforms and a(x) b(x) - c(x) on a domain of 2^20 .. 2^22
// points -- 6 x 2^L x L / 2 butterflies of one Montgomery product each.  ARITHMETIC-bound: ~150 M products per
// EmailVerifier(1024,1536) proof against ~2 GB of HBM traffic; the bound is the issue rate of v_mad_u64_u32.  No MFMA:
// the products are 254-bit modular integers.
//
// Structure:
//   * decimation in frequency (natural order in, bit-reversed out) for the inverse transforms, decimation in time
//     (bit-reversed in, natural out) for the forward ones: no permutation pass between them, and the coset scaling
//     inc^i / n is a table indexed by bit-reversed position, fused into the last inverse pass;
//   * each transform is ceil(L / 7) passes over HBM ("four-step" recursion): COLUMN passes run 2^g-point sub-transforms (g <= 7)
//     on 1024 / 2^g neighbouring columns of a block at a time in LDS, followed (DIF) or preceded (DIT) by the block twiddle
//     w_N^(column * frequency); the last / first pass transforms contiguous 2^g-element rows, 1,024 elements per workgroup;
//   * round 6: VALUES STAY IN 9 x 29-BIT LIMB FORM across the butterfly stages of a pass (zkwg_fr29.h): a product is the 250-instruction
//     product-scanning form without split / pack / conditional subtraction (315 behind the 4 x 64-bit interface), an addition is 9
//     adds, a subtraction adds a multiple of r that dominates the subtrahend (round 5: carry chain + compare + select each).  Data
//     keep the callers' 2^256 form, the twiddle / scale tables are in 2^261 form (one operand in that form is what the 2^-261 of the
//     product needs).  Bounds: a DIF stage pair grows a value 4 x at most (radix-4 group: y0 = x0 + x1 + x2 + x3), so with pass
//     inputs < 5 r the three pairs + one single stage of a 7-stage pass stay below 640 r (the top limb holds what exceeds 2^232:
//     < 2^32 up to 1,352 r), and the pass's closing product with a canonical twiddle returns < (640 / 169 + 1) r < 5 r; a DIT stage adds
//     at most 3 r.  Limbs 0 .. 7 are carry-normalised where a sum feeds a sum (two of a group's four values per stage pair).
//   * the work buffer between passes holds limb form too, planar (16 + 16 + 4 bytes per element in three arrays per polynomial):
//     nothing is reduced or packed until a b - c leaves as canonical words;
//   * one workgroup = 1,024 elements = 36 KiB of LDS + its twiddles: 3 workgroups per CU.
#include "zkwg_kernels.h"
#include "zkwg_fr29.h"

#define ZK_NTT_TILE 1024u   // elements per workgroup: a column pass takes 1024 / 2^g neighbouring columns (>= 128 contiguous bytes per row access)
#define ZK_NTT_GMAX 7u      // butterfly stages per pass (the value bounds above)

__device__ __forceinline__ u32 zk_bitrev(u32 x, u32 bits) { return bits ? (__brev(x) >> (32u - bits)) : 0u; }
// w^e for the transform's direction: tw[k] = w^k (2^261 form), k < n; the inverse direction reads w^(n - e)
__device__ __forceinline__ Fr29 zk_ntt_tw(const Fr* __restrict__ tw, u64 n, u64 e, bool inv) {
  e &= n - 1u;
  return fr29_from_fr(tw[inv ? ((n - e) & (n - 1u)) : e]);
}
__device__ __forceinline__ Fr29 zk_l29(uint4 a, uint4 b, u32 t) { return Fr29{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, t}}; }

// LDS layout: limbs 0-3, limbs 4-7 and the top limb of an element live in separate arrays (lo[i], hi[i], top[i]): a wavefront's
// ds_read_b128 of consecutive elements covers each bank once
struct ZkLds29 {
  uint4* lo; uint4* hi; u32* top;
  __device__ __forceinline__ Fr29 get(u32 i) const { return zk_l29(lo[i], hi[i], top[i]); }
  __device__ __forceinline__ void put(u32 i, const Fr29& v) const {
    lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]); hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]); top[i] = v.l[8];
  }
};
__device__ __forceinline__ ZkLds29 zk_lds29(uint4* base, u32 count) { return ZkLds29{base, base + count, (u32*)(base + 2u * count)}; }
// bytes of LDS for `count` elements (16-byte granules)
__host__ __device__ static inline size_t zk_lds29_bytes(u32 count) { return ((size_t)count * 36 + 15) & ~(size_t)15; }

// a polynomial in HBM: canonical words (the callers' arrays: A.w | B.w | C.w, stand-alone transforms) or the planar limb form of the work buffer
struct ZkNttBuf {
  const void* base;
  u64 es, ps, valid;      // canonical: polynomial q at base + (q / 3) es + (q % 3) ps (Fr units), `valid` elements (zero beyond)
  u32 lazy;               // 1: planar limb form, polynomial q at base + q * 36 n bytes
};
__device__ __forceinline__ Fr29 zk_ntt_load(const ZkNttBuf& b, u64 n, u64 q, u64 idx) {
  if (b.lazy) {
    const u8* p = (const u8*)b.base + q * 36u * n;
    return zk_l29(((const uint4*)p)[idx], ((const uint4*)(p + 16u * n))[idx], ((const u32*)(p + 32u * n))[idx]);
  }
  const Fr* s = (const Fr*)b.base + (q / 3u) * b.es + (q % 3u) * b.ps;
  return idx < b.valid ? fr29_from_fr(s[idx]) : fr29_zero();
}
// V: bound of the value in units of r when the destination is canonical
template <int V>
__device__ __forceinline__ void zk_ntt_store(const ZkNttBuf& b, u64 n, u64 q, u64 idx, const Fr29& v) {
  if (b.lazy) {
    u8* p = (u8*)b.base + q * 36u * n;
    ((uint4*)p)[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    ((uint4*)(p + 16u * n))[idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    ((u32*)(p + 32u * n))[idx] = v.l[8];
  } else {
    ((Fr*)b.base + (q / 3u) * b.es + (q % 3u) * b.ps)[idx] = fr29_to_fr_v<V>(v);
  }
}

// ---- butterflies in limb form.  I: the inputs' value bound in units of r; every input has limbs 0 .. 7 < 2^29 (DIF) -------------------
// DIF radix-4 group (two stages): 4 products, 2 carry normalisations; outputs < 4 I r with normalised limbs
template <int I>
__device__ __forceinline__ void zk_dif4(const Fr29& x0, const Fr29& x1, const Fr29& x2, const Fr29& x3, const Fr29& wa, const Fr29& wb, const Fr29& w2,
                                        Fr29& y0, Fr29& y1, Fr29& y2, Fr29& y3) {
  const Fr29 a0 = fr29_add(x0, x2), a1 = fr29_add(x1, x3);                                  // [2, 2 I]
  const Fr29 a2 = fr29_mul(fr29_sub<I + 1, 1>(x0, x2), wa);                                 // operand [3, 2 I + 1] -> [1, (2 I + 1) / 169 + 1 <= 5]
  const Fr29 a3 = fr29_mul(fr29_sub<I + 1, 1>(x1, x3), wb);
  y0 = fr29_norm(fr29_add(a0, a1));                                                         // [1, 4 I]
  y1 = fr29_mul(fr29_sub<2 * I + 1, 2>(a0, a1), w2);                                        // operand [5, 4 I + 1] -> [1, (4 I + 1) / 169 + 1]
  y2 = fr29_norm(fr29_add(a2, a3));                                                         // [1, 10]
  y3 = fr29_mul(fr29_sub<6, 1>(a2, a3), w2);                                                // [1, 2]
}
template <int I>
__device__ __forceinline__ void zk_dif2(const Fr29& a, const Fr29& b, const Fr29& w, Fr29& y0, Fr29& y1) {
  y0 = fr29_norm(fr29_add(a, b));                                                           // [1, 2 I]
  y1 = fr29_mul(fr29_sub<I + 1, 1>(a, b), w);                                               // [1, (2 I + 1) / 169 + 1]
}
// DIT radix-4 group: x0, x2 normalised by the caller, x1, x3 limbs < 6 2^29; every product is below 2 r, an output gains at most 6 r
__device__ __forceinline__ void zk_dit4(const Fr29& x0, const Fr29& x1, const Fr29& x2, const Fr29& x3, const Fr29& w1, const Fr29& wp, const Fr29& wq,
                                        Fr29& y0, Fr29& y1, Fr29& y2, Fr29& y3) {
  const Fr29 t1 = fr29_mul(x1, w1), t3 = fr29_mul(x3, w1);                                  // [1, 2]
  const Fr29 a0 = fr29_add(x0, t1), a1 = fr29_sub<3, 1>(x0, t1);                            // [2, V + 2], [3, V + 3]
  const Fr29 u2 = fr29_mul(fr29_add(x2, t3), wp), u3 = fr29_mul(fr29_sub<3, 1>(x2, t3), wq);    // operands [2], [3] -> [1, 2]
  y0 = fr29_add(a0, u2); y2 = fr29_sub<3, 1>(a0, u2);                                       // [3, V + 4], [4, V + 5]
  y1 = fr29_add(a1, u3); y3 = fr29_sub<3, 1>(a1, u3);                                       // [4, V + 5], [5, V + 6]
}

// The butterfly stages of 2^g-point sub-transforms over `nel` elements per column, C columns interleaved (element i of
// column cc at i * C + cc), two stages per pass through LDS where possible.
template <bool DIT>
__device__ __forceinline__ void zk_ntt_stages(const ZkLds29& y, const ZkLds29& twl, u32 G, u32 g, u32 nel, u32 C) {
  u32 st = 0;
  for (; st + 1u < g; st += 2u) {
    const u32 h = DIT ? (1u << st) : (G >> (st + 2u));
    for (u32 b = threadIdx.x; b < (nel / 4u) * C; b += 256u) {
      const u32 cc = b % C, q = b / C;
      const u32 p = q % h, i0 = (q / h) * 4u * h + p;
      const u32 e0 = i0 * C + cc, e1 = (i0 + h) * C + cc, e2 = (i0 + 2u * h) * C + cc, e3 = (i0 + 3u * h) * C + cc;
      Fr29 y0, y1, y2, y3;
      if (DIT) {
        zk_dit4(fr29_norm(y.get(e0)), y.get(e1), fr29_norm(y.get(e2)), y.get(e3), twl.get(p * (G / (2u * h))), twl.get(p * (G / (4u * h))),
                twl.get((p + h) * (G / (4u * h))), y0, y1, y2, y3);
        y.put(e0, y0); y.put(e2, y2); y.put(e1, y1); y.put(e3, y3);
      } else {
        const Fr29 x0 = y.get(e0), x1 = y.get(e1), x2 = y.get(e2), x3 = y.get(e3);
        const Fr29 wa = twl.get(p << st), wb = twl.get((p + h) << st), w2 = twl.get(p << (st + 1u));
        // (the inputs' bound: 5 r at the first stage pair of a pass, 4 x more at each following one)
        if (st == 0) zk_dif4<5>(x0, x1, x2, x3, wa, wb, w2, y0, y1, y2, y3);
        else if (st == 2) zk_dif4<20>(x0, x1, x2, x3, wa, wb, w2, y0, y1, y2, y3);
        else zk_dif4<80>(x0, x1, x2, x3, wa, wb, w2, y0, y1, y2, y3);
        y.put(e0, y0); y.put(e1, y1); y.put(e2, y2); y.put(e3, y3);
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
      if (DIT) {
        const Fr29 a = fr29_norm(y.get(i * C + cc)), tt = fr29_mul(y.get(j * C + cc), twl.get(k));
        y.put(i * C + cc, fr29_add(a, tt));
        y.put(j * C + cc, fr29_sub<3, 1>(a, tt));
      } else {
        const Fr29 a = y.get(i * C + cc), bb = y.get(j * C + cc), w = twl.get(k);
        Fr29 y0, y1;
        if (st == 0) zk_dif2<5>(a, bb, w, y0, y1);
        else if (st == 2) zk_dif2<20>(a, bb, w, y0, y1);
        else if (st == 4) zk_dif2<80>(a, bb, w, y0, y1);
        else zk_dif2<320>(a, bb, w, y0, y1);
        y.put(i * C + cc, y0);
        y.put(j * C + cc, y1);
      }
    }
    __syncthreads();
  }
}

// One column pass.  Block size N = 2^lb (the sub-problem of this recursion level), sub-transform size G = 2^g over the rows
// r of a column: element index = block * N + r * (N >> g) + c.  DIF: sub-transform, then y *= w_N^(c * bitrev_g(r)).
// DIT: y *= w_N^(c * bitrev_g(r)) first, then the sub-transform.  Polynomial q = blockIdx.y.
template <bool DIT>
__global__ __launch_bounds__(256) void zk_ntt_col(ZkNttBuf src, ZkNttBuf dst,   /* launched in place: src may alias dst */
                                                   const Fr* __restrict__ tw, u32 L, u32 lb, u32 g, u32 inv) {
  extern __shared__ uint4 lds4[];
  const u64 n = 1ull << L;
  const u32 TILE = n < ZK_NTT_TILE ? (u32)n : ZK_NTT_TILE;      // (domains below 1,024 points: one workgroup, fewer columns)
  const u32 G = 1u << g, C = TILE >> g;
  const ZkLds29 y = zk_lds29(lds4, G * C);                                           // [G][C] elements
  const ZkLds29 twl = zk_lds29(lds4 + zk_lds29_bytes(G * C) / 16u, G / 2u);          // w_G^k, k < G / 2 (direction applied)
  const u32 cols_per_block = 1u << (lb - g);
  const u64 cid0 = (u64)blockIdx.x * C;
  const u64 block = cid0 >> (lb - g);
  const u32 c0 = (u32)(cid0 & (cols_per_block - 1u));
  const u64 q = blockIdx.y;
  const u64 base = block << lb;
  const bool invb = inv != 0;
  for (u32 k = threadIdx.x; k < G / 2u; k += 256u) twl.put(k, zk_ntt_tw(tw, n, (u64)k << (L - g), invb));
  for (u32 t = threadIdx.x; t < G * C; t += 256u) {
    const u32 r = t / C, cc = t % C;
    const u64 idx = base + ((u64)r << (lb - g)) + c0 + cc;
    Fr29 v = zk_ntt_load(src, n, q, idx);
    if (DIT) v = fr29_mul(v, zk_ntt_tw(tw, n, ((u64)(c0 + cc) * zk_bitrev(r, g)) << (L - lb), invb));     // limbs < 6 2^29, value < 30 r -> [1, 2]
    y.put(t, v);
  }
  __syncthreads();
  zk_ntt_stages<DIT>(y, twl, G, g, G, C);
  for (u32 t = threadIdx.x; t < G * C; t += 256u) {
    const u32 r = t / C, cc = t % C;
    const u64 idx = base + ((u64)r << (lb - g)) + c0 + cc;
    Fr29 v = y.get(t);
    if (!DIT) v = fr29_mul(v, zk_ntt_tw(tw, n, ((u64)(c0 + cc) * zk_bitrev(r, g)) << (L - lb), invb));    // < 640 r -> [1, 5]
    zk_ntt_store<32>(dst, n, q, idx, v);
  }
}

// The row pass: contiguous blocks of G = 2^g elements, 1,024 elements per workgroup.  DIF (inverse direction of the
// pipeline): optional multiplication by scale[position] on the way out (coset shift and 1 / n).
template <bool DIT>
__global__ __launch_bounds__(256) void zk_ntt_row(ZkNttBuf src, ZkNttBuf dst,   /* launched in place: src may alias dst */
                                                   const Fr* __restrict__ tw, const Fr* __restrict__ scale, Fr uni, u32 use_uni, u32 L, u32 g, u32 inv) {
  extern __shared__ uint4 lds4[];
  const u64 n = 1ull << L;
  const u32 G = 1u << g;
  const u32 TILE = n < 1024u ? (u32)n : 1024u;
  const ZkLds29 y = zk_lds29(lds4, TILE);                                            // [TILE] elements
  const ZkLds29 twl = zk_lds29(lds4 + zk_lds29_bytes(TILE) / 16u, G / 2u);           // w_G^k, k < G / 2
  const u64 q = blockIdx.y;
  const u64 base = (u64)blockIdx.x * TILE;
  const bool invb = inv != 0;
  for (u32 k = threadIdx.x; k < G / 2u; k += 256u) twl.put(k, zk_ntt_tw(tw, n, (u64)k << (L - g), invb));
  for (u32 t = threadIdx.x; t < TILE; t += 256u) y.put(t, zk_ntt_load(src, n, q, base + t));
  __syncthreads();
  zk_ntt_stages<DIT>(y, twl, G, g, TILE, 1u);
  for (u32 t = threadIdx.x; t < TILE; t += 256u) {
    Fr29 v = y.get(t);
    if (scale) v = fr29_mul(v, fr29_from_fr(scale[base + t]));
    else if (use_uni) v = fr29_mul(v, fr29_from_fr(uni));     // (stand-alone inverse transform: 1 / n)
    zk_ntt_store<32>(dst, n, q, base + t, v);                 // (DIT without a product: < 5 + 21 r)
  }
}

// out[k] = a[k] b[k] - c[k]  (joinABC of groth16_prove.js), polynomials of email e at work + (3 e + {0, 1, 2}) * 36 n bytes (limb form,
// 2^256-form values below 30 r); out: canonical words, 2^256 form.  mul(a, b) carries 2^256 2^256 / 2^261: the constant 2^266 restores it.
__global__ __launch_bounds__(256) void zk_ntt_join(ZkNttBuf work, Fr* __restrict__ out, u64 n, u64 out_es) {
  const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 e = blockIdx.y;
  const Fr29 a = fr29_norm(zk_ntt_load(work, n, 3u * e, i)), b = fr29_norm(zk_ntt_load(work, n, 3u * e + 1u, i)), c = fr29_norm(zk_ntt_load(work, n, 3u * e + 2u, i));
  const Fr29 k266 = Fr29{{0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u}};     // 2^266 mod r
  const Fr29 ab = fr29_mul(fr29_mul(a, b), k266);             // [1, 30 30 / 169 + 1 = 7] -> [1, 2]
  out[e * out_es + i] = fr29_to_fr_v<32>(fr29_sub<31, 1>(ab, c));
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
// reverse (DIT).  src: canonical words (src_lazy = 0: polynomial q at src + (q / 3) src_es + (q % 3) src_ps, `valid` elements) or the
// work buffer itself; `work`: n_polys x 36 n bytes; the LAST pass writes to `out` when it is given (canonical words, polynomial q at
// out + q n: the stand-alone transforms), to the work buffer otherwise.
extern "C" int zk_ntt_launch(int dit, const Fr* src, u64 src_es, u64 src_ps, u64 valid, int src_lazy, void* work, Fr* out, const Fr* tw, const Fr* scale,
                             const Fr* uni_host, u32 L, u32 n_polys, u32 inv, hipStream_t st) {
  const Fr uni = uni_host ? *uni_host : fr_zero();
  const u32 use_uni = uni_host ? 1u : 0u;
  const u64 n = 1ull << L;
  // L stages in ceil(L / GMAX) passes of nearly equal size; the row pass takes the last share (and at most log2 of its tile)
  const u32 np = (L + ZK_NTT_GMAX - 1u) / ZK_NTT_GMAX;
  u32 gs[8];
  for (u32 i = 0; i < np; ++i) gs[i] = L / np + (i < L % np ? 1u : 0u);
  const u32 g_row = gs[np - 1u], ng = np - 1u;
  const u32 tile = n < 1024u ? (u32)n : 1024u;
  const size_t row_lds = zk_lds29_bytes(tile) + zk_lds29_bytes((1u << g_row) / 2u);
  const dim3 rgrid((u32)(n / tile), n_polys);
  const ZkNttBuf W{work, 0, 0, n, 1u};
  ZkNttBuf S = src_lazy ? W : ZkNttBuf{src, src_es, src_ps, valid, 0u};
  // canonical destination addressing is (q / 3) es + (q % 3) ps: consecutive polynomials n apart = es 3 n, ps n
  const ZkNttBuf OUT{out, 3u * n, n, n, 0u};
  if (!dit) {
    u32 lb = L;
    for (u32 i = 0; i < ng; ++i) {
      const u32 g = gs[i];
      const size_t lds = zk_lds29_bytes(tile) + zk_lds29_bytes((1u << g) / 2u);
      hipLaunchKernelGGL((zk_ntt_col<false>), dim3((u32)(n / tile), n_polys), dim3(256), lds, st, S, W, tw, L, lb, g, inv);
      S = W;
      lb -= g;
    }
    hipLaunchKernelGGL((zk_ntt_row<false>), rgrid, dim3(256), row_lds, st, S, out ? OUT : W, tw, scale, uni, use_uni, L, g_row, inv);
  } else {
    hipLaunchKernelGGL((zk_ntt_row<true>), rgrid, dim3(256), row_lds, st, S, (ng == 0 && out) ? OUT : W, tw, (const Fr*)nullptr, uni, 0u, L, g_row, inv);
    u32 lb = g_row;
    for (u32 i = ng; i-- > 0;) {
      const u32 g = gs[i];
      lb += g;
      const size_t lds = zk_lds29_bytes(tile) + zk_lds29_bytes((1u << g) / 2u);
      hipLaunchKernelGGL((zk_ntt_col<true>), dim3((u32)(n / tile), n_polys), dim3(256), lds, st, W, (i == 0 && out) ? OUT : W, tw, L, lb, g, inv);
    }
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int zk_ntt_join_launch(const void* work, Fr* out, u64 n, u64 out_es, u32 n_emails, hipStream_t st) {
  const ZkNttBuf W{work, 0, 0, n, 1u};
  hipLaunchKernelGGL(zk_ntt_join, dim3((u32)((n + 255u) / 256u), n_emails), dim3(256), 0, st, W, out, n, out_es);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int zk_ntt_bitrev_launch(Fr* data, u32 L, u32 n_polys, hipStream_t st) {
  hipLaunchKernelGGL(zk_ntt_bitrev, dim3((u32)(((1ull << L) + 255u) / 256u), n_polys), dim3(256), 0, st, data, L);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
