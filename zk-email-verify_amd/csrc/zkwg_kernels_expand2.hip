// zk_expand (LDS-staged) -- the streaming expansion kernel, the HBM-write-bound kernel of the path.
//
// Every witness element is a 32-byte little-endian field element, but > 95 % of the EmailVerifier witness
// is single bits and nearly all the rest fits 31 bits.  The prepare kernels leave a compact per-email IMAGE;
// this kernel expands image + input record into the `.wtns` data section (the witness vector of
// `circuit.calculateWitness(input)`, packages/circuits/tests/email-verifier.test.ts:43),
// guided by the circuit's static segment table (zkwg_sched.h).
//
// One workgroup = one contiguous portion (s.portion slots, 64 KiB of output at 2048) of one witness, in two
// phases that never mix loads with stores:
//   decode : every slot of the portion becomes a 4-byte code in LDS (zkwg_expand_dec.h: the value itself, or a
//            reference to 32 bytes held elsewhere) -- one slot per thread per pass, all image reads happen here;
//   store  : a fill-shaped loop: each wavefront store instruction covers 1 KiB of contiguous HBM, the stores of
//            a thread are issued back to back with no load or wait between them (a portion whose segments are all
//            immediate-valued takes the branch-free path; the others fetch their references 4 chunks at a time).
// Consecutive workgroups write consecutive portions, each XCD streams through its own contiguous share of the
// launch, every witness byte is written exactly once and never read.
//
// MONT = true (prover hand-off): the same two phases, the store phase writes x * 2^256 mod r:
// 0 -> 0, 1 -> R, v < 2^16 -> table of v * R; references resolve to Montgomery-form copies (the image's field
// elements and the record's limbs converted once per batch by zk_image_to_mont, the Montgomery inverse table), so
// the store loop carries no field multiplication.
#include "zkwg_expand_dec.h"
#include "zkwg_kernels.h"

#define ZK_X2_THREADS 256u

template <class DEC>
__device__ __forceinline__ void zk_fill_codes(u32* __restrict__ code, const ZkSeg& sg, const ZkCtx& cx, u32 r0, u32 n, u32 tid) {
  const DEC dec(sg, cx);
#pragma unroll 4
  for (u32 i = tid; i < n; i += ZK_X2_THREADS) code[i] = dec(r0 + i);
}

__device__ __forceinline__ void zk_decode_segment(u32* __restrict__ code, const ZkSeg& sg, const ZkCtx& cx, u32 r0, u32 n, u32 tid) {
  switch (sg.type) {
#define ZK_X(T, D) case T: zk_fill_codes<D>(code, sg, cx, r0, n, tid); break;
    ZK_FOR_SEG_TYPES(ZK_X)
#undef ZK_X
    default:   // ZSEG_HOLE: nothing produces these slots
      for (u32 i = tid; i < n; i += ZK_X2_THREADS) code[i] = 0u;
      break;
  }
}

// ---------------------------------------------------------------- store phase, standard form
// chunk c = 16 bytes; even chunks carry an immediate slot's value, odd chunks its (zero) high half
template <int ITERS>
__device__ __forceinline__ void zk_store_pure_full(uint4* __restrict__ dst, const u32* __restrict__ code, u32 tid) {
  u32 v[ITERS];
#pragma unroll
  for (int k = 0; k < ITERS; ++k) v[k] = code[(tid >> 1) + (ZK_X2_THREADS / 2u) * (u32)k];
  const bool lo = !(tid & 1u);
#pragma unroll
  for (int k = 0; k < ITERS; ++k) dst[tid + ZK_X2_THREADS * (u32)k] = zk_small(lo ? v[k] : 0u);
}
__device__ __forceinline__ void zk_store_pure(uint4* __restrict__ dst, const u32* __restrict__ code, u32 nch, u32 tid) {
  if (nch == 16u * ZK_X2_THREADS) { zk_store_pure_full<16>(dst, code, tid); return; }
  if (nch == 8u * ZK_X2_THREADS) { zk_store_pure_full<8>(dst, code, tid); return; }
  if (nch == 4u * ZK_X2_THREADS) { zk_store_pure_full<4>(dst, code, tid); return; }
  const bool lo = !(tid & 1u);
  for (u32 c = tid; c < nch; c += ZK_X2_THREADS) dst[c] = zk_small(lo ? code[c >> 1] : 0u);
}
// portions that hold references: the 16 bytes of a reference come through one unconditional load per chunk
// (immediate chunks read the all-zero inverse-table entry of d = 0), 4 chunks in flight per thread
__device__ __forceinline__ void zk_store_general(uint4* __restrict__ dst, const u32* __restrict__ code, u32 nch, u32 tid,
                                                 const ZkRefSrc& R, const uint4* __restrict__ zero16) {
  const u32 hf = tid & 1u;
  for (u32 c0 = tid; c0 < nch; c0 += 4u * ZK_X2_THREADS) {
    u32 cd[4];
    bool any_ref = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 c = c0 + ZK_X2_THREADS * (u32)k;
      cd[k] = c < nch ? code[c >> 1] : 0u;
      any_ref = any_ref || (cd[k] >> 31);
    }
    if (__builtin_amdgcn_ballot_w64(any_ref) == 0ull) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 c = c0 + ZK_X2_THREADS * (u32)k;
        if (c < nch) dst[c] = zk_small(hf ? 0u : cd[k]);
      }
      continue;
    }
    uint4 ld[4];
    bool rare = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 t = ZK_REF_TYPE(cd[k]), p = ZK_REF_PAYLOAD(cd[k]);
      const uint4* a = zero16;
      if (cd[k] >> 31) {
        if (t == 0u) a = R.frv + 2u * p + hf;
        else if (t == 1u) a = R.invtab + 2u * p + hf;
        else if (t == 2u) a = hf ? zero16 : (const uint4*)(R.rec + p);
        else rare = true;
      }
      ld[k] = *a;
    }
    const bool any_rare = __builtin_amdgcn_ballot_w64(rare) != 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 c = c0 + ZK_X2_THREADS * (u32)k;
      uint4 v = (cd[k] >> 31) ? ld[k] : zk_small(hf ? 0u : cd[k]);
      if (any_rare && (cd[k] >> 31) && ZK_REF_TYPE(cd[k]) >= 3u) v = zk_ref_half(cd[k], hf, R);
      if (c < nch) dst[c] = v;
    }
  }
}

// ---------------------------------------------------------------- store phase, Montgomery form
__device__ __forceinline__ uint4 zk_fr_half4(const Fr& m, u32 hf) {
  const u64 x = m.l[2 * hf], y = m.l[2 * hf + 1];
  return make_uint4((u32)x, (u32)(x >> 32), (u32)y, (u32)(y >> 32));
}
// v * R mod r of an immediate: 0, R, the table below 2^16, one product otherwise (rare: nothing in EmailVerifier)
__device__ __noinline__ uint4 zk_mont_imm_slow(u32 v, u32 hf, const uint4* __restrict__ rtab) {
  if (v < 65536u) return rtab[2u * v + hf];
  return zk_fr_half4(fr_to_mont(Fr{{(u64)v, 0, 0, 0}}), hf);
}
__device__ __forceinline__ uint4 zk_mont_imm(u32 v, u32 hf, const uint4* __restrict__ rtab) {
  if (v > 1u) return zk_mont_imm_slow(v, hf, rtab);
  return v ? zk_fr_half4(fr_R(), hf) : zk_zero4();   // R = 2^256 mod r
}
struct ZkMontSrc {
  ZkRefSrc R;                        // frv = Montgomery copies (image field elements, then the record's limbs), invtab = Montgomery table
  const uint4* __restrict__ rtab;    // v * R for v < 65536
  u32 limb_base, limb_off;           // index of the first converted limb inside R.frv / record offset of limb 0
};
__device__ __noinline__ uint4 zk_mont_ref_rare(u32 code, u32 hf, const ZkMontSrc& M) {
  // RAW: a 32-bit value with bit 31 set; NEG: r - m
  const u32 w = M.R.small[ZK_REF_PAYLOAD(code)];
  if (ZK_REF_TYPE(code) == 3u) return zk_fr_half4(fr_to_mont(Fr{{(u64)w, 0, 0, 0}}), hf);
  const u32 m = (u32)(-((int)(w << 1) >> 1));
  return zk_fr_half4(fr_neg(fr_to_mont(Fr{{(u64)m, 0, 0, 0}})), hf);
}
__device__ __forceinline__ void zk_store_mont(uint4* __restrict__ dst, const u32* __restrict__ code, u32 nch, u32 tid,
                                              const ZkMontSrc& M, bool pure) {
  const u32 hf = tid & 1u;
  for (u32 c = tid; c < nch; c += ZK_X2_THREADS) {
    const u32 cd = code[c >> 1];
    uint4 v;
    if (pure || !(cd >> 31)) v = zk_mont_imm(cd, hf, M.rtab);
    else {
      const u32 t = ZK_REF_TYPE(cd), p = ZK_REF_PAYLOAD(cd);
      if (t == 0u) v = M.R.frv[2u * p + hf];
      else if (t == 1u) v = M.R.invtab[2u * p + hf];
      else if (t == 2u) v = M.R.frv[2u * (M.limb_base + ((p - M.limb_off) >> 4)) + hf];
      else v = zk_mont_ref_rare(cd, hf, M);
    }
    dst[c] = v;
  }
}

// store phase of one portion of one email (the codes of its slots are in LDS)
template <bool MONT>
__device__ __forceinline__ void zk_store_phase(const ZkSched& s, const ZkBufs& B, const ZkCtx& cx, u32 e, uint4* __restrict__ dst,
                                               const u32* __restrict__ code, u32 nch, u32 tid, bool pure) {
  if constexpr (MONT) {
    ZkMontSrc M;
    M.R.frv = (const uint4*)(B.frm + (u64)e * (s.img_fr + ZK_MONT_LIMBS));
    M.R.invtab = (const uint4*)B.invtab_m;
    M.R.rec = cx.rec; M.R.small = cx.small;
    M.rtab = (const uint4*)B.rtab;
    M.limb_base = s.img_fr; M.limb_off = s.in_off[3];   // ZKWG_IN_PUBKEY: pubkey, signature, message limbs are contiguous
    zk_store_mont(dst, code, nch, tid, M, pure);
  } else if (pure) {
    zk_store_pure(dst, code, nch, tid);
  } else {
    ZkRefSrc R;
    R.frv = (const uint4*)(B.frv + (u64)e * s.img_fr);
    R.invtab = (const uint4*)B.invtab;
    R.rec = cx.rec; R.small = cx.small;
    zk_store_general(dst, code, nch, tid, R, (const uint4*)(B.invtab + s.inv_half));   // (d = 0)^-1 := 0: 32 zero bytes
  }
}

// XCD-aware workgroup -> unit mapping.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8); give every
// XCD a contiguous run of units: xcd_remap = 1: one run per XCD over the whole launch; K > 1: runs of K workgroups
// inside groups of 8 K.
__device__ __forceinline__ u32 zk_xcd_unit(u32 xcd_remap) {
  u32 blk = blockIdx.x;
  if (xcd_remap == 1u) {
    const u32 per = gridDim.x >> 3;
    if (blk < per * 8u) blk = (blk & 7u) * per + (blk >> 3);
  } else if (xcd_remap > 1u) {
    const u32 K = xcd_remap, G = 8u * K;
    const u32 g = blk / G, r = blk - g * G;
    if ((g + 1u) * G <= gridDim.x) blk = g * G + (r & 7u) * K + (r >> 3);
  }
  return blk;
}
__device__ __forceinline__ ZkCtx zk_email_ctx(const ZkSched& s, const ZkBufs& B, u32 e) {
  ZkCtx cx;
  cx.rec = B.in + (u64)e * s.in_stride;
  cx.bits = B.bits + (u64)e * s.img_bits;
  cx.small = B.small + (u64)e * s.img_small;
  cx.half = (int)s.inv_half;
  cx.m_dfa_cm = s.m_dfa_cm; cx.m_dfa_pm = s.m_dfa_pm;
  return cx;
}

// ---------------------------------------------------------------- the kernel
template <bool MONT>
__device__ __forceinline__ void zk_expand2_body(const ZkSched& s, const ZkBufs& B, u32* __restrict__ code) {
  const u32 blk = zk_xcd_unit(B.xcd_remap);
  // workgroup (p, g): portion p of the emails [g*E, g*E+E) of this launch
  const u32 p = blk % s.nportions;
  const u32 g = blk / s.nportions;
  const u32 E = B.emails_per_wg;
  const u32 el0 = g * E;                                   // first email (launch-local index)
  const u32 el1 = min(el0 + E, B.n_emails - B.e_first);    // one past the last
  if (el0 >= el1) return;
  const u64 slot0 = (u64)p * s.portion;
  const u64 slot1 = min(s.W, slot0 + s.portion);
  const u32 nch = (u32)(slot1 - slot0) * 2u;
  const u32 tid = threadIdx.x;
  const u32 si0 = B.first_seg[p];
  const bool pure = (B.pflags[p] & 1u) != 0u;
  for (u32 el = el0; el < el1; ++el) {
    const u32 e = el + B.e_first;                // email index inside the prepared batch
    const ZkCtx cx = zk_email_ctx(s, B, e);
    for (u32 si = si0; si < s.nsegs; ++si) {
      const ZkSeg sg = B.segs[si];
      if (sg.slot >= slot1) break;
      const u64 lo = max(sg.slot, slot0);
      const u64 hi = min(sg.slot + sg.nslots, slot1);
      zk_decode_segment(code + (u32)(lo - slot0), sg, cx, (u32)(lo - sg.slot) + sg.r0, (u32)(hi - lo), tid);
    }
    __syncthreads();
    zk_store_phase<MONT>(s, B, cx, e, B.wit + (u64)el * B.wit_stride16 + slot0 * 2, code, nch, tid, pure);
    if (el + 1 < el1) __syncthreads();
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void zk_expand2(ZkSched s, ZkBufs B) {
  extern __shared__ u32 zk_x2_code[];
  zk_expand2_body<false>(s, B, zk_x2_code);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void zk_expand2_mont(ZkSched s, ZkBufs B) {
  extern __shared__ u32 zk_x2_code[];
  zk_expand2_body<true>(s, B, zk_x2_code);
}

