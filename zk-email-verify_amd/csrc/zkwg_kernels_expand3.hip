// zk_expand (one-shot) -- the streaming expansion kernel, the HBM-write-bound kernel of the path.
//
// Every witness element is a 32-byte little-endian field element, but > 95 % of the EmailVerifier witness is single
// bits and nearly all the rest fits 31 bits.  The prepare kernels leave a compact per-email IMAGE; this kernel
// expands image + input record into the `.wtns` data section (the witness vector of
// `circuit.calculateWitness(input)`, packages/circuits/tests/email-verifier.test.ts:43).
//
// Shape (measured, tools/storebench2.hip / DESIGN.md section 5): HBM takes plain 16-byte stores fastest when every
// workgroup writes ONE small contiguous piece and exits -- 8 KiB per 256-thread workgroup reaches 6.8-6.9 TB/s even
// with two dependent loads in front of the stores, 64 KiB per workgroup 5.8-6.1 -- because the chip then writes one
// dense window that sweeps the buffer in address (= dispatch) order.  So:
//   * one workgroup = 256 K consecutive slots (8 K KiB, K = 1, 2 or 4) of one witness, K slots per thread whose image
//     reads are all in flight together, no loop, no LDS, no barrier;
//   * the workgroup's prologue is one 32-byte table entry (ZkPortionEntry, built on the host): for the > 90 % of
//     pieces that lie inside a single segment it carries the segment's parameters and the piece's offset, so the
//     dependent chain in front of the stores is entry -> image word -> store; pieces that straddle segments walk the
//     segment table (a handful of entries);
//   * a lane decodes its slot into a 32-bit code (zkwg_expand_dec.h: the value, or a reference to 32 bytes held
//     elsewhere); the two lanes that store the slot's halves fetch the code with a wavefront shuffle, so each
//     wavefront store instruction covers 1 KiB of contiguous HBM and every witness byte is written exactly once;
//   * each XCD streams through its own contiguous share of the launch (blockIdx remap).
//
// MONT (prover hand-off): values leave as x * 2^256 mod r: 0 -> 0, 1 -> R, v < 2^16 -> table; references
// resolve to Montgomery-form copies (zk_image_to_mont, the Montgomery inverse table): no field product in this kernel.
// O0 (`circom --O0` / `--O1` builds, zkwg_o0.h): the code of a wire comes from its 8-byte descriptor instead of the
// segment arithmetic -- same store side.
#include "zkwg_expand_dec.h"
#include "zkwg_kernels.h"
#include "zkwg_o0.h"
#include "zkwg_o0_dec.h"

__device__ __forceinline__ uint4 zk_fr_half4(const Fr& m, u32 hf) {
  const u64 x = hf ? m.l[2] : m.l[0], y = hf ? m.l[3] : m.l[1];   // (selects: a runtime index would put the element in scratch memory)
  return make_uint4((u32)x, (u32)(x >> 32), (u32)y, (u32)(y >> 32));
}
// Montgomery form of the rare values no table holds.  Out of line, with the reference sources passed BY VALUE: a reference
// to the ZkRefSrc of the caller would force that struct into scratch memory -- 32 bytes per lane stored on every email
// iteration whether or not the call happens (PMC: 1.39 x the algorithmic bytes written by the Montgomery kernels, r03_pmc_abc)
__device__ __noinline__ uint4 zk_mont_slow(u32 code, u32 hf, const uint4* frv, const uint4* invtab, const u8* rec, const u32* small) {
  if (!(code >> 31)) return zk_fr_half4(fr_to_mont(Fr{{(u64)code, 0, 0, 0}}), hf);   // an immediate >= 2^16
  // RAW / NEG / I64: standard-form value first (zk_ref_half), then one product
  ZkRefSrc R;
  R.frv = frv; R.invtab = invtab; R.rec = rec; R.small = small;
  const uint4 a = zk_ref_half(code, 0u, R), b = zk_ref_half(code, 1u, R);
  const Fr x{{(u64)a.x | ((u64)a.y << 32), (u64)a.z | ((u64)a.w << 32), (u64)b.x | ((u64)b.y << 32), (u64)b.z | ((u64)b.w << 32)}};
  return zk_fr_half4(fr_to_mont(x), hf);
}
// the 16-byte half `hf` of the slot whose code is `code`
template <bool MONT>
__device__ __forceinline__ uint4 zk_slot_half(u32 code, u32 hf, const ZkX3& A, const ZkRefSrc& R) {
  if constexpr (!MONT) {
    return (code >> 31) ? zk_ref_half(code, hf, R) : zk_small(hf ? 0u : code);
  } else {
    if (!(code >> 31)) {
      if (code <= 1u) return code ? zk_fr_half4(fr_R(), hf) : zk_zero4();   // R = 2^256 mod r
      if (code < 65536u) return ((const uint4*)A.rtab)[2u * code + hf];
      return zk_mont_slow(code, hf, R.frv, R.invtab, R.rec, R.small);
    }
    const u32 t = ZK_REF_TYPE(code), p = ZK_REF_PAYLOAD(code);
    if (t == 0u) return R.frv[2u * p + hf];                                       // Montgomery copy of the image's fr
    if (t == 1u) return R.invtab[2u * p + hf];                                    // Montgomery inverse table
    if (t == 2u) return R.frv[2u * (A.img_fr + ((p - A.limb_off) >> 4)) + hf];    // converted limb
    // integers of the image (RAW / NEG / I64: results of integer rows, negative small values): |v| < 2^16 -> the table,
    // r - table[|v|] for a negative one (the A.w / B.w / C.w of bit constraints are full of -1 and -2)
    {
      long long v;
      if (t == 6u) v = -(long long)p;
      else {
        const u32 w = R.small[p];
        if (t == 3u) v = (long long)w; else if (t == 4u) v = (long long)((int)(w << 1) >> 1); else v = (long long)((u64)w | ((u64)R.small[p + 1] << 32));
      }
      const u64 m = v < 0 ? (u64)(-v) : (u64)v;
      if (m < 65536u) {
        const uint4* tb = (const uint4*)A.rtab + 2u * (u32)m;
        if (v >= 0) return tb[hf];
        // r - x, x = m R mod r, 0 < x < r: low half with its borrow, high half minus that borrow
        const uint4 xl = tb[0];
        const u64 x0 = (u64)xl.x | ((u64)xl.y << 32), x1 = (u64)xl.z | ((u64)xl.w << 32);
        const u64 r0 = ZK_P0, r1 = ZK_P1, r2 = ZK_P2, r3 = ZK_P3;
        const u64 d0 = r0 - x0, b0 = r0 < x0 ? 1ull : 0ull;
        const u64 d1 = r1 - x1 - b0, b1 = (r1 < x1 || (r1 == x1 && b0)) ? 1ull : 0ull;
        if (!hf) return make_uint4((u32)d0, (u32)(d0 >> 32), (u32)d1, (u32)(d1 >> 32));
        const uint4 xh = tb[1];
        const u64 x2 = (u64)xh.x | ((u64)xh.y << 32), x3 = (u64)xh.z | ((u64)xh.w << 32);
        const u64 d2 = r2 - x2 - b1, b2 = (r2 < x2 || (r2 == x2 && b1)) ? 1ull : 0ull;
        const u64 d3 = r3 - x3 - b2;
        return make_uint4((u32)d2, (u32)(d2 >> 32), (u32)d3, (u32)(d3 >> 32));
      }
    }
    return zk_mont_slow(code, hf, R.frv, R.invtab, R.rec, R.small);
  }
}

__device__ __forceinline__ u32 zk_x3_unit(u32 xcd_remap) {
  // workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8).  1: each XCD streams through one contiguous eighth of the
  // whole launch; G > 1: through a contiguous run of G pieces inside each group of 8 G pieces (the eight streams then stay
  // within 8 G pieces of each other); 0: plain order
  u32 blk = blockIdx.x;
  if (xcd_remap == 1u) {
    const u32 per = gridDim.x >> 3;
    if (blk < per * 8u) blk = (blk & 7u) * per + (blk >> 3);
  } else if (xcd_remap > 1u) {
    const u32 G = xcd_remap, grp = blk / (8u * G), r = blk - grp * 8u * G;
    if ((grp + 1u) * 8u * G <= gridDim.x) blk = grp * 8u * G + (r & 7u) * G + (r >> 3);
  }
  return blk;
}
__device__ __forceinline__ ZkCtx zk_x3_ctx(const ZkX3& A, u32 e) {
  ZkCtx cx;
  cx.rec = A.in + (u64)e * A.in_stride;
  cx.bits = A.bits + (u64)e * A.img_bits;
  cx.small = A.small + (u64)e * A.img_small;
  cx.half = (int)A.inv_half;
  cx.m_dfa_cm = A.m_dfa_cm; cx.m_dfa_pm = A.m_dfa_pm; cx.m_dfa_st = A.m_dfa_st;
  cx.nd = A.netd;
  return cx;
}
// store side: wavefront w owns the 64 K consecutive slots [64 K w, 64 K (w + 1)) of the piece (K = slots per lane);
// lane l holds the codes of slots 64 k + l (k < K).  Chunk 64 j + l (j < 2 K) of the wavefront's 2 K KiB belongs to
// slot 32 j + (l >> 1): register j / 2, fetched from lane 32 (j & 1) + (l >> 1) with one wavefront shuffle.
template <bool MONT, int K>
__device__ __forceinline__ void zk_x3_store(const ZkX3& A, const ZkCtx& cx, u32 e, u32 el, u64 slot0, u32 nsl, const u32 (&code)[K]) {
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, hf = lane & 1u;
  ZkRefSrc R;
  R.frv = MONT ? (const uint4*)(A.frm + (u64)e * (A.img_fr + ZK_MONT_LIMBS)) : (const uint4*)(A.frv + (u64)e * A.img_fr);
  R.invtab = (const uint4*)(MONT ? A.invtab_m : A.invtab);
  R.rec = cx.rec; R.small = cx.small;
  const u32 w0 = 64u * K * wv;                                        // first slot of this wavefront inside the piece
  uint4* __restrict__ dst = A.wit + (u64)el * A.wit_stride16 + (slot0 + w0) * 2u;
  const u32 left = nsl > w0 ? (nsl - w0) * 2u : 0u;                   // chunks of this wavefront inside the witness
  uint4 v[2 * K];
#pragma unroll
  for (int j = 0; j < 2 * K; ++j) {
    const u32 c = __shfl(code[j >> 1], 32u * (j & 1) + (lane >> 1));
    v[j] = zk_slot_half<MONT>(c, hf, A, R);
  }
#pragma unroll
  for (int j = 0; j < 2 * K; ++j)
    if (64u * j + lane < left) dst[64u * j + lane] = v[j];
}

template <bool MONT, int K>
__device__ __forceinline__ void zk_expand3_body(const ZkX3& A) {
  constexpr u32 SLOTS = 256u * K;
  const u32 unit = zk_x3_unit(A.xcd_remap);
  const u32 p = unit % A.nportions, el = unit / A.nportions;   // piece p of email el (launch-local)
  const u32 e = el + A.e_first;
  const u64 slot0 = (u64)p * SLOTS;
  const u32 nsl = (u32)min((u64)SLOTS, A.W - slot0);
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const ZkCtx cx = zk_x3_ctx(A, e);
  const ZkPortionEntry en = A.ent[p];
  u32 code[K];
  if (en.type < ZSEG_NTYPES) {
    // the whole piece lies inside one segment: its parameters came with the entry
    ZkSeg sg;
    sg.slot = 0; sg.nslots = 0; sg.type = en.type; sg.src = en.src; sg.a = en.a; sg.b = en.b; sg.c = en.c; sg.r0 = 0; sg.pad = en.magic;
    zk_decode_k<K>(sg, en.r_start, 64u * K * wv + lane, nsl, cx, code);
  } else {
    // the piece straddles segments: find each slot's
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const u32 i = 64u * K * wv + 64u * k + lane;
      code[k] = 0u;
      if (i < nsl) {
        const u64 slot = slot0 + i;
        for (u32 si = en.first_seg; si < A.nsegs; ++si) {
          const ZkSeg sg = A.segs[si];
          if (sg.slot > slot) break;
          if (slot < sg.slot + sg.nslots) { code[k] = zk_decode_any(sg, (u32)(slot - sg.slot) + sg.r0, cx); break; }
        }
      }
    }
  }
  zk_x3_store<MONT, K>(A, cx, e, el, slot0, nsl, code);
}
#define ZK_X3_KERNELS(K)                                                                                                                  \
  __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_k##K(ZkX3 A) { zk_expand3_body<false, K>(A); }    \
  __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_mont_k##K(ZkX3 A) { zk_expand3_body<true, K>(A); }
ZK_X3_KERNELS(1)
ZK_X3_KERNELS(2)
ZK_X3_KERNELS(4)
ZK_X3_KERNELS(8)

// Montgomery-form copies of what the references of one email name: its img_fr field elements, then the
// ZK_MONT_LIMBS 128-bit limbs of the record (pubkey, signature, message) -- one product each, once per expansion.
__global__ __launch_bounds__(256) void zk_image_to_mont(ZkX3 A) {
  const u32 per = A.img_fr + ZK_MONT_LIMBS;
  const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
  if (i >= (u64)A.n_count * per) return;
  const u32 e = A.e_first + (u32)(i / per), j = (u32)(i % per);
  Fr x;
  if (j < A.img_fr) x = A.frv[(u64)e * A.img_fr + j];
  else {
    const u64* l = (const u64*)(A.in + (u64)e * A.in_stride + A.limb_off + 16u * (j - A.img_fr));
    x = Fr{{l[0], l[1], 0, 0}};
  }
  A.frm_w[(u64)e * per + j] = fr_to_mont(x);
}

// ---------------------------------------------------------------- numbered circuits (`--O0` / `--O1`), one pass
// What a descriptor reads, resolved once per workgroup: one 64-bit word of `bits`, one byte of the record, two words of
// `small` (index 0 of each array when the kind does not use it).  Per email the lane then issues all its loads back to
// back -- nothing between them depends on a loaded value -- and only afterwards turns them into codes.
struct ZkO0Pre { u32 i64, i8, i32, i32b; };
__device__ __forceinline__ ZkO0Pre zk_o0_pre(uint2 d, const ZkX3& A) {
  ZkO0Pre p{0u, 0u, 0u, 0u};
  switch (d.x >> 28) {
    case ZK_D_BIT64: case ZK_D_BITRUN: p.i64 = d.y; break;
    case ZK_D_BIT8: case ZK_D_BYTE: p.i8 = d.y; break;
    case ZK_D_SMALLRAW: case ZK_D_CODEW: case ZK_D_SMALLN: p.i32 = d.y; break;
    case ZK_D_SMALLS: p.i32 = d.y; p.i32b = d.y + 1u; break;
    case ZK_D_DFA: {
      const u32 dk = (d.x >> 24) & 15u, i = (d.x >> 9) & 0x7ffu;
      p.i32 = A.m_dfa_st + i + (dk == ZDFA_SUB ? 1u : 0u);
      p.i32b = dk == ZDFA_CLS ? A.m_dfa_pm + i : (dk == ZDFA_AND ? A.m_dfa_cm + i : 0u);
      break;
    }
    default: break;
  }
  return p;
}
__device__ __forceinline__ u32 zk_o0_combine(uint2 d, const ZkO0Pre& p, u64 w64, u32 w8, u32 w32, u32 w32b, int half) {
  const u32 a = d.x, b = d.y;
  switch (a >> 28) {
    case ZK_D_IMM: return b;
    case ZK_D_BIT64: return (u32)(w64 >> (a & 63u)) & 1u;
    case ZK_D_BITRUN: return (u32)(w64 >> (a & 63u)) & ((2u << ((a >> 6) & 31u)) - 1u);
    case ZK_D_BIT8: return (w8 >> (a & 7u)) & 1u;
    case ZK_D_BYTE: return w8;
    case ZK_D_SMALLRAW: return zk_raw_code(w32, b);
    case ZK_D_CODEW: return w32;
    case ZK_D_SMALLN: return zk_narrow_code(w32, b);
    case ZK_D_SMALLS: return zk_wide_code(w32, w32b, b);
    case ZK_D_DFA: return zk_dfa_value_w((a >> 24) & 15u, (a >> 20) & 15u, a & 511u, b, w32, w32b, half);
    default: return 0u;
  }
}
#define ZK_O0_BATCH 4
template <bool MONT, int K, int PIPE>
__device__ __forceinline__ void zk_expand3_o0_body(const ZkX3& A, const ZkO0Dev& O) {
  constexpr u32 SLOTS = 256u * K;
  const u32 unit = zk_x3_unit(A.xcd_remap);
  const u32 p = unit % O.nportions, g = unit / O.nportions;     // piece p of the emails [g E, g E + E) of this launch
  const u64 slot0 = (u64)p * SLOTS;
  const u32 nsl = (u32)min((u64)SLOTS, O.W - slot0);
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  // the descriptors of this thread's K wires: loaded once, reused for every email of the group (8 bytes per wire
  // against 32 bytes written per wire and email)
  uint2 d[K];
  u32 aff[K];               // ZK_D_AFF wires: c0 | c1 << 16 applied to the source's value (0x00010000 = identity: every other wire)
  ZkO0Pre pre[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const u32 i = 64u * K * wv + 64u * k + lane;
    d[k] = i < nsl ? O.desc[slot0 + i] : make_uint2(0u, 0u);
    aff[k] = 0x00010000u;
    if ((d[k].x >> 28) == ZK_D_AFF) { const uint4 a = O.aff[d[k].y]; d[k] = make_uint2(a.x, a.y); aff[k] = a.z; }
    pre[k] = zk_o0_pre(d[k], A);
  }
  // The loop over the group's emails is software-pipelined: the image words of email i + 1 are requested before email i is
  // turned into codes and stored (the stores may alias the image as far as the compiler knows, so it would not hoist the
  // loads itself), otherwise every email pays the load latency in front of its stores.
  const u32 el0 = g * O.emails_per_wg, el1 = min((g + 1u) * O.emails_per_wg, A.n_count);
  if constexpr (PIPE >= 2) {
    // The kernel is bound by VALU issue, not by HBM (per email and wavefront ~250 instructions: four loads, a switch over the
    // descriptor kinds that executes every kind present in the wavefront, shuffles, the store-side switch): so pieces whose
    // wires are all of one simple kind -- workgroup-uniform, decided once -- take a short path.  Constants (the C side of every
    // bit constraint, aliases of the constant wire: a fifth of an A.w | B.w | C.w record) need no load at all; single bits and bit
    // fields (the SHA regions: three quarters of an `--O0` witness) need one.
    bool only_imm = true, only_bits = true;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const u32 kd = d[k].x >> 28;
      const bool plain = aff[k] == 0x00010000u;
      only_imm = only_imm && plain && kd == ZK_D_IMM;
      only_bits = only_bits && plain && (kd == ZK_D_IMM || kd == ZK_D_BIT64 || kd == ZK_D_BITRUN);
    }
    const int cls = __syncthreads_and(only_imm ? 1 : 0) ? 0 : (__syncthreads_and(only_bits ? 1 : 0) ? 1 : 2);
    if (cls == 0) {
      u32 code[K];
#pragma unroll
      for (int k = 0; k < K; ++k) code[k] = d[k].y;
      for (u32 el = el0; el < el1; ++el) {
        const u32 e = el + A.e_first;
        zk_x3_store<MONT, K>(A, zk_x3_ctx(A, e), e, el, slot0, nsl, code);
      }
      return;
    }
    if (cls == 1) {
      constexpr int NB = ZK_O0_BATCH;
      u32 sh[K], mk[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const u32 kd = d[k].x >> 28;
        sh[k] = d[k].x & 63u;
        mk[k] = kd == ZK_D_BIT64 ? 1u : (kd == ZK_D_BITRUN ? (2u << ((d[k].x >> 6) & 31u)) - 1u : 0u);   // 0: a constant (IMM)
      }
      for (u32 el = el0; el < el1; el += NB) {
        u64 b64[NB][K];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const u64* bits = A.bits + (u64)(min(el + (u32)q, el1 - 1u) + A.e_first) * A.img_bits;
#pragma unroll
          for (int k = 0; k < K; ++k) b64[q][k] = bits[pre[k].i64];
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          if (el + (u32)q >= el1) break;
          const u32 e = el + q + A.e_first;
          u32 code[K];
#pragma unroll
          for (int k = 0; k < K; ++k) code[k] = mk[k] ? ((u32)(b64[q][k] >> sh[k]) & mk[k]) : d[k].y;
          zk_x3_store<MONT, K>(A, zk_x3_ctx(A, e), e, el + q, slot0, nsl, code);
        }
      }
      return;
    }
    // batches of ZK_O0_BATCH emails: the image words of the whole batch are requested back to back (4 x the loads in flight),
    // then the batch is turned into codes and stored email by email; PIPE == 3 requests the next batch before storing this one
    constexpr int NB = ZK_O0_BATCH;
    u64 b64[NB][K], c64[NB][K]; u32 b8[NB][K], b32[NB][K], b32b[NB][K], c8[NB][K], c32[NB][K], c32b[NB][K];
    auto fetch = [&](u32 first, u64 (&x64)[NB][K], u32 (&x8)[NB][K], u32 (&x32)[NB][K], u32 (&x32b)[NB][K]) {
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const ZkCtx cx = zk_x3_ctx(A, min(first + (u32)q, el1 - 1u) + A.e_first);
#pragma unroll
        for (int k = 0; k < K; ++k) { x64[q][k] = cx.bits[pre[k].i64]; x8[q][k] = cx.rec[pre[k].i8]; x32[q][k] = cx.small[pre[k].i32]; x32b[q][k] = cx.small[pre[k].i32b]; }
      }
    };
    if (el0 < el1) fetch(el0, b64, b8, b32, b32b);
    for (u32 el = el0; el < el1; el += NB) {
      if constexpr (PIPE == 3) fetch(min(el + NB, el1 - 1u), c64, c8, c32, c32b);
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        if (el + (u32)q >= el1) break;
        const u32 e = el + q + A.e_first;
        const ZkCtx cx = zk_x3_ctx(A, e);
        u32 code[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          code[k] = zk_o0_combine(d[k], pre[k], b64[q][k], b8[q][k], b32[q][k], b32b[q][k], cx.half);
          if (aff[k] != 0x00010000u) code[k] = zk_aff_apply(aff[k], code[k], cx);
        }
        zk_x3_store<MONT, K>(A, cx, e, el + q, slot0, nsl, code);
      }
      if constexpr (PIPE == 3) {
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
          for (int k = 0; k < K; ++k) { b64[q][k] = c64[q][k]; b8[q][k] = c8[q][k]; b32[q][k] = c32[q][k]; b32b[q][k] = c32b[q][k]; }
      } else if (el + NB < el1) fetch(el + NB, b64, b8, b32, b32b);
    }
    return;
  }
  u64 w64[K], n64[K]; u32 w8[K], w32[K], w32b[K], n8[K], n32[K], n32b[K];
  if (el0 < el1) {
    const ZkCtx cx = zk_x3_ctx(A, el0 + A.e_first);
#pragma unroll
    for (int k = 0; k < K; ++k) { w64[k] = cx.bits[pre[k].i64]; w8[k] = cx.rec[pre[k].i8]; w32[k] = cx.small[pre[k].i32]; w32b[k] = cx.small[pre[k].i32b]; }
  }
  for (u32 el = el0; el < el1; ++el) {
    const u32 e = el + A.e_first;
    const ZkCtx cx = zk_x3_ctx(A, e);
    if constexpr (PIPE == 1) {
      const ZkCtx nx = zk_x3_ctx(A, min(el + 1u, el1 - 1u) + A.e_first);
#pragma unroll
      for (int k = 0; k < K; ++k) { n64[k] = nx.bits[pre[k].i64]; n8[k] = nx.rec[pre[k].i8]; n32[k] = nx.small[pre[k].i32]; n32b[k] = nx.small[pre[k].i32b]; }
    }
    u32 code[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      code[k] = zk_o0_combine(d[k], pre[k], w64[k], w8[k], w32[k], w32b[k], cx.half);
      if (aff[k] != 0x00010000u) code[k] = zk_aff_apply(aff[k], code[k], cx);
    }
    zk_x3_store<MONT, K>(A, cx, e, el, slot0, nsl, code);
    if constexpr (PIPE == 1) {
#pragma unroll
      for (int k = 0; k < K; ++k) { w64[k] = n64[k]; w8[k] = n8[k]; w32[k] = n32[k]; w32b[k] = n32b[k]; }
    } else if (el + 1u < el1) {
      const ZkCtx nx = zk_x3_ctx(A, el + 1u + A.e_first);
#pragma unroll
      for (int k = 0; k < K; ++k) { w64[k] = nx.bits[pre[k].i64]; w8[k] = nx.rec[pre[k].i8]; w32[k] = nx.small[pre[k].i32]; w32b[k] = nx.small[pre[k].i32b]; }
    }
  }
}
#define ZK_X3_O0_KERNELS(K)                                                                                                                                     \
  __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_o0_k##K(ZkX3 A, ZkO0Dev O) { zk_expand3_o0_body<false, K, 0>(A, O); }      \
  __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_o0_mont_k##K(ZkX3 A, ZkO0Dev O) { zk_expand3_o0_body<true, K, 0>(A, O); } \
  __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_o0p_k##K(ZkX3 A, ZkO0Dev O) { zk_expand3_o0_body<false, K, 1>(A, O); }      \
  __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_o0p_mont_k##K(ZkX3 A, ZkO0Dev O) { zk_expand3_o0_body<true, K, 1>(A, O); }
ZK_X3_O0_KERNELS(1)
ZK_X3_O0_KERNELS(2)
ZK_X3_O0_KERNELS(4)
// K = 1 in batches of 4 emails (PIPE 2: loads of a batch together; 3: and the next batch requested before this one is stored)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_o0b_k1(ZkX3 A, ZkO0Dev O) { zk_expand3_o0_body<false, 1, 2>(A, O); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_o0b_mont_k1(ZkX3 A, ZkO0Dev O) { zk_expand3_o0_body<true, 1, 2>(A, O); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_o0c_k1(ZkX3 A, ZkO0Dev O) { zk_expand3_o0_body<false, 1, 3>(A, O); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand3_o0c_mont_k1(ZkX3 A, ZkO0Dev O) { zk_expand3_o0_body<true, 1, 3>(A, O); }

// wires whose value comes from their segment's own arithmetic (ItemAtIndex selectors, comparators, Base64 ...: 1.5 % of the
// wires of EmailVerifier): their codes, once per email, into small[gen_base + g] -- the streaming kernel then reads them like
// any other small word and carries no segment decoder (126 -> 60 VGPRs)
__global__ __launch_bounds__(256) void zk_o0_generic(ZkX3 A, ZkO0Dev O) {
  const u32 g = blockIdx.x * 256u + threadIdx.x;
  if (g >= O.n_gen) return;
  const u32 e = A.e_first + blockIdx.y;
  const ZkCtx cx = zk_x3_ctx(A, e);
  const ZkSeg sg = A.segs[O.gen_seg[g]];
  A.small_w[(u64)e * A.img_small + O.gen_base + g] = zk_decode_any(sg, O.gen_r[g], cx);
}
// rows whose sources, coefficients and result are small integers: 64-bit integer arithmetic
__device__ __forceinline__ long long zk_small_row_terms(const ZkO0Dev& O, u32 j, const ZkCtx& cx) {
  long long acc = 0;
  for (u64 t = O.s_ptr[j]; t < O.s_ptr[j + 1]; ++t) {
    const uint2 d = O.s_term[t];
    acc += (long long)O.s_coef[t] * zk_code_int(zk_desc_decode(d.x, d.y, cx), cx);
  }
  return acc;
}
// one thread per row that is a group of its own, for ZK_ROW_EMAILS emails: the row's table entries (pointer, terms,
// coefficients) are read once, and per term the image loads of all the emails are issued back to back before any of
// them is used -- the kernel is a chain of dependent loads, so what counts is how many are in flight
// terms t0 + first, t0 + first + step, ... of row j for the emails el0 .. el0 + ZK_ROW_EMAILS - 1 of the launch
__device__ __forceinline__ void zk_row_accumulate(const ZkX3& A, const ZkO0Dev& O, u32 j, u32 el0, u32 first, u32 step, long long acc[ZK_ROW_EMAILS]) {
  const u32 last = A.n_count - 1u;
#pragma unroll
  for (int k = 0; k < ZK_ROW_EMAILS; ++k) acc[k] = 0;
  const u64 t1 = O.s_ptr[j + 1];
  for (u64 t = O.s_ptr[j] + first; t < t1; t += step) {
    const uint2 d = O.s_term[t];
    const long long cf = (long long)O.s_coef[t];
    // (neighbouring rows have the same shape, so the kinds are all but uniform over a wavefront: each kind issues only
    // the loads it needs, for all the emails back to back)
    const u32 kind = d.x >> 28;
    if (kind == ZK_D_BIT64 || kind == ZK_D_BITRUN) {
      u64 w[ZK_ROW_EMAILS];
#pragma unroll
      for (int k = 0; k < ZK_ROW_EMAILS; ++k) w[k] = A.bits[(u64)(A.e_first + min(el0 + (u32)k, last)) * A.img_bits + d.y];
      const u32 mask = kind == ZK_D_BIT64 ? 1u : (2u << ((d.x >> 6) & 31u)) - 1u;
#pragma unroll
      for (int k = 0; k < ZK_ROW_EMAILS; ++k) acc[k] += cf * (long long)((u32)(w[k] >> (d.x & 63u)) & mask);
    } else if (kind == ZK_D_IMM) {
#pragma unroll
      for (int k = 0; k < ZK_ROW_EMAILS; ++k) acc[k] += cf * (long long)d.y;   // (a reference is never small-ranged)
    } else if (kind == ZK_D_BIT8 || kind == ZK_D_BYTE) {
      u32 w[ZK_ROW_EMAILS];
#pragma unroll
      for (int k = 0; k < ZK_ROW_EMAILS; ++k) w[k] = A.in[(u64)(A.e_first + min(el0 + (u32)k, last)) * A.in_stride + d.y];
#pragma unroll
      for (int k = 0; k < ZK_ROW_EMAILS; ++k) acc[k] += cf * (long long)(kind == ZK_D_BYTE ? w[k] : (w[k] >> (d.x & 7u)) & 1u);
    } else {
      const ZkO0Pre pre = zk_o0_pre(d, A);
      u32 w32[ZK_ROW_EMAILS], w32b[ZK_ROW_EMAILS];
#pragma unroll
      for (int k = 0; k < ZK_ROW_EMAILS; ++k) {
        const u32* sm = A.small + (u64)(A.e_first + min(el0 + (u32)k, last)) * A.img_small;
        w32[k] = sm[pre.i32]; w32b[k] = sm[pre.i32b];
      }
#pragma unroll
      for (int k = 0; k < ZK_ROW_EMAILS; ++k) {
        const ZkCtx cx = zk_x3_ctx(A, A.e_first + min(el0 + (u32)k, last));
        acc[k] += cf * zk_code_int(zk_o0_combine(d, pre, 0ull, 0u, w32[k], w32b[k], cx.half), cx);
      }
    }
  }
}
__device__ __forceinline__ void zk_row_put(u32* __restrict__ small_e, u32 where, long long v) {
  if (where >> 31) *(uint2*)(small_e + (where & 0x7fffffffu)) = make_uint2((u32)(u64)v, (u32)((u64)v >> 32));
  else small_e[where] = (u32)(u64)v;   // (range within +-2^30: bit 31 is the sign)
}
__device__ __forceinline__ void zk_row_store(const ZkX3& A, const ZkO0Dev& O, u32 j, u32 el0, const long long acc[ZK_ROW_EMAILS]) {
  const u32 where = O.s_out[j];
#pragma unroll
  for (int k = 0; k < ZK_ROW_EMAILS; ++k)
    if (el0 + (u32)k < A.n_count) zk_row_put(A.small_w + (u64)(A.e_first + el0 + k) * A.img_small, where, acc[k]);
}
__global__ __launch_bounds__(256) void zk_o0_rows_small(ZkX3 A, ZkO0Dev O) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= O.n_small_single) return;
  const u32 j = O.s_single[i], el0 = blockIdx.y * ZK_ROW_EMAILS;
  long long acc[ZK_ROW_EMAILS];
  zk_row_accumulate(A, O, j, el0, 0u, 1u, acc);
  zk_row_store(A, O, j, el0, acc);
}
// rows of more than ZK_ROW_LONG terms (up to 1,633 in EmailVerifier: the sums over a whole header): one wavefront per
// row, the lanes stride over the terms, a butterfly adds the partial sums
__global__ __launch_bounds__(256) void zk_o0_rows_small_long(ZkX3 A, ZkO0Dev O) {
  const u32 i = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (i >= O.n_small_long) return;
  const u32 j = O.s_long[i], el0 = blockIdx.y * ZK_ROW_EMAILS;
  long long acc[ZK_ROW_EMAILS];
  zk_row_accumulate(A, O, j, el0, lane, 64u, acc);
#pragma unroll
  for (int k = 0; k < ZK_ROW_EMAILS; ++k)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const u32 lo = __shfl_xor((u32)(u64)acc[k], d), hi = __shfl_xor((u32)((u64)acc[k] >> 32), d);
      acc[k] += (long long)((u64)lo | ((u64)hi << 32));
    }
  if (lane == 0) zk_row_store(A, O, j, el0, acc);
}
// one wavefront per chain (zkwg_o0.h: every row continues the sum of the row before it -- the running sums of
// MultiOR / CalculateTotal / popcount chains): lane = row, each lane sums the terms its row adds, an inclusive prefix
// sum over the wavefront (shuffles) turns them into the rows' values, 64 rows per round
__global__ __launch_bounds__(64) void zk_o0_chains_small(ZkX3 A, ZkO0Dev O) {
  const uint2 ch = O.s_chains[blockIdx.x];
  const u32 e = A.e_first + blockIdx.y, lane = threadIdx.x;
  const ZkCtx cx = zk_x3_ctx(A, e);
  u32* out = A.small_w + (u64)e * A.img_small;
  long long carry = 0;
  // the first row carries the whole sum so far (up to 1,625 terms in EmailVerifier's constraint system): when it is long
  // the lanes share its terms instead of leaving them to lane 0
  const u64 h0 = O.s_ptr[ch.x], h1 = O.s_ptr[ch.x + 1];
  const bool coop = h1 - h0 > 32u;
  long long head = 0;
  if (coop) {
    for (u64 t = h0 + lane; t < h1; t += 64u) {
      const uint2 d = O.s_term[t];
      head += (long long)O.s_coef[t] * zk_code_int(zk_desc_decode(d.x, d.y, cx), cx);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const u32 lo = __shfl_xor((u32)(u64)head, d), hi = __shfl_xor((u32)((u64)head >> 32), d);
      head += (long long)((u64)lo | ((u64)hi << 32));
    }
  }
  for (u32 base = 0; base < ch.y; base += 64u) {
    const bool live = base + lane < ch.y;
    const u32 j = ch.x + base + lane;
    long long v = !live ? 0 : (coop && base + lane == 0u) ? head : zk_small_row_terms(O, j, cx);
#pragma unroll
    for (u32 d = 1; d < 64u; d <<= 1) {
      const u32 lo = __shfl_up((u32)(u64)v, d), hi = __shfl_up((u32)((u64)v >> 32), d);
      if (lane >= d) v += (long long)((u64)lo | ((u64)hi << 32));
    }
    v += carry;
    if (live) zk_row_put(out, O.s_out[j], v);
    carry = (long long)((u64)(u32)__shfl((u32)(u64)v, 63) | ((u64)(u32)__shfl((u32)((u64)v >> 32), 63) << 32));
  }
}
// the other rows: arithmetic mod r (zk_linear_row's), sources decoded from the image.
// One wavefront = one row for 64 emails (lane = email).  Everything the row's table holds -- term pointer, descriptor, kind,
// coefficient -- is wave-uniform: loaded once per wavefront through the scalar path, every branch on it is uniform, and the
// coefficient sits in scalar registers where the multiplier reads it.  What differs per lane is the operand, fetched from
// that email's image.  Products with a generic coefficient are accumulated unreduced (17 x 32-bit limbs, fr_wide_mac:
// 64 multiplier issues instead of the 128 of a Montgomery product) and reduced once per 16 terms; coefficients of +-1 and
// bit operands are plain modular additions.  Rows that read the same operands (the 33 evaluation points of one FpMul, the
// rows of one interpolation matrix) are neighbours in the table: each XCD takes one contiguous range of rows, so they
// share its L2.  (Round 3: 8 lanes per row chasing dependent loads, 128 bytes of table per term and lane: 15.7 ms per
// 1,024 emails of EmailVerifier(576,192)'s constraint system, 84 k products per email.)
__global__ __launch_bounds__(64) void zk_o0_rows_fr(ZkX3 A, ZkO0Dev O) {
  u32 blk = blockIdx.x;
  { const u32 per = gridDim.x >> 3; if (blk < per * 8u) blk = (blk & 7u) * per + (blk >> 3); }
  const u32 j = __builtin_amdgcn_readfirstlane(blk);
  const u32 lane = threadIdx.x, el = blockIdx.y * 64u + lane, last = A.n_count - 1u;
  const u32 e = A.e_first + min(el, last);
  const ZkCtx cx = zk_x3_ctx(A, e);
  ZkRefSrc R;
  R.frv = (const uint4*)(A.frv + (u64)e * A.img_fr);
  R.invtab = (const uint4*)A.invtab;
  R.rec = cx.rec; R.small = cx.small;
  FrWide w;
  fr_wide_zero(w);
  u32 nw = 0;
  Fr s = fr_zero();
  const u64 t0 = O.f_ptr[j], t1 = O.f_ptr[j + 1];
  for (u64 t = t0; t < t1; ++t) {
    const uint2 d = O.f_term[t];
    const u32 kd = O.f_kind[t], dk = d.x >> 28;
    const u32 code = zk_desc_decode(d.x, d.y, cx);
    if (kd == ZK_COEF_ONE) { s = fr_add(s, zk_code_value(code, R)); continue; }
    if (kd == ZK_COEF_MINUS_ONE) { s = fr_sub(s, zk_code_value(code, R)); continue; }
    if (dk == ZK_D_BIT64 || dk == ZK_D_BIT8) {
      // a bit times a coefficient (the powers of two of Bits2Num ...): add the coefficient or not
      const Fr cf = O.f_coef[t];
      if (code) s = fr_add(s, cf);
      continue;
    }
    const Fr cfm = O.f_coefm[t];
    fr_wide_mac(w, zk_code_value(code, R), cfm);
    if (++nw == 16u) { s = fr_add(s, fr_wide_redc(w)); fr_wide_zero(w); nw = 0; }
  }
  if (nw) s = fr_add(s, fr_wide_redc(w));
  if (el <= last) A.frv_w[(u64)e * A.img_fr + O.fr_base + j] = s;
}
