// Decoding of the per-wire descriptors of a numbered layout (zkwg_o0.h) into slot codes, and of codes into integers / field
// elements: shared by the device kernels (zkwg_kernels_expand3.hip) and the host evaluation of the same tables
// (zkwg_abc_host in zkwg_api.hip: layout-only handles, tests).
#pragma once
#include "zkwg_expand_dec.h"
#include "zkwg_o0.h"

// codes of integer row results: a small non-negative value is an immediate, a small negative one the load-free r - m
// (every bit constraint has a B side of 0 / -1: a reference into the image would put a dependent load in front of the store)
ZK_DEC __forceinline__ u32 zk_narrow_code(u32 w, u32 b) {
  if (!(w >> 31)) return w;
  return w > 0xf0000000u ? (ZK_REF_MINUS | (0u - w)) : (ZK_REF_NEG | b);
}
ZK_DEC __forceinline__ u32 zk_wide_code(u32 lo, u32 hi, u32 b) {
  if (hi == 0u && !(lo >> 31)) return lo;
  if (hi == 0xffffffffu && lo > 0xf0000000u) return ZK_REF_MINUS | (0u - lo);
  return ZK_REF_I64 | b;
}
// the code of one wire from its descriptor (zkwg_o0.h)
ZK_DEC __forceinline__ u32 zk_desc_decode(u32 a, u32 b, const ZkCtx& cx) {
  switch (a >> 28) {
    case ZK_D_IMM: return b;
    case ZK_D_BIT64: return (u32)(cx.bits[b] >> (a & 63u)) & 1u;
    case ZK_D_BITRUN: return (u32)(cx.bits[b] >> (a & 63u)) & ((2u << ((a >> 6) & 31u)) - 1u);
    case ZK_D_BIT8: return (u32)(cx.rec[b] >> (a & 7u)) & 1u;
    case ZK_D_BYTE: return cx.rec[b];
    case ZK_D_SMALLRAW: return zk_raw_code(cx.small[b], b);
    case ZK_D_CODEW: return cx.small[b];
    case ZK_D_SMALLN: return zk_narrow_code(cx.small[b], b);
    case ZK_D_SMALLS: return zk_wide_code(cx.small[b], cx.small[b + 1], b);
    case ZK_D_DFA: return zk_dfa_value((a >> 24) & 15u, (a >> 9) & 0x7ffu, (a >> 20) & 15u, a & 511u, b, cx.small + cx.m_dfa_st, cx.small + cx.m_dfa_cm, cx.small + cx.m_dfa_pm, cx.half);
    default: return 0u;   // (no wire keeps kind GENERIC: zk_o0_build turns them into CODEW)
  }
}
// the value of a code as a signed integer (small rows: every source is small-ranged by construction)
ZK_DEC __forceinline__ long long zk_code_int(u32 code, const ZkCtx& cx) {
  if (!(code >> 31)) return (long long)code;
  const u32 p = ZK_REF_PAYLOAD(code);
  if (ZK_REF_TYPE(code) == 6u) return -(long long)p;                               // MINUS
  const u32 w = cx.small[p];
  switch (ZK_REF_TYPE(code)) {
    case 3: return (long long)w;                                                   // RAW
    case 4: return (long long)((int)(w << 1) >> 1);                                // NEG
    default: return (long long)((u64)w | ((u64)cx.small[p + 1] << 32));            // I64 (no other reference is small-ranged)
  }
}
// the value of a code as a field element (standard form)
ZK_DEC __forceinline__ Fr zk_code_value(u32 code, const ZkRefSrc& R) {
  if (!(code >> 31)) return Fr{{(u64)code, 0, 0, 0}};
  const uint4 a = zk_ref_half(code, 0u, R), b = zk_ref_half(code, 1u, R);
  return Fr{{(u64)a.x | ((u64)a.y << 32), (u64)a.z | ((u64)a.w << 32), (u64)b.x | ((u64)b.y << 32), (u64)b.z | ((u64)b.w << 32)}};
}

// ZK_D_AFF: c0 + c1 * source (zkwg_o0.h), `code` = the source's code
ZK_DEC __forceinline__ u32 zk_aff_apply(u32 packed, u32 code, const ZkCtx& cx) {
  const long long sv = (code >> 31) ? zk_code_int(code, cx) : (long long)code;
  const long long v = (long long)(short)(packed & 0xffffu) + (long long)(short)(packed >> 16) * sv;
  return v >= 0 ? (u32)v : (ZK_REF_MINUS | (u32)(-v));
}
// the code of wire (a, b) of a descriptor table whose affine entries are `aff` (host evaluation; the streaming kernel resolves
// the entry once per workgroup instead)
ZK_DEC __forceinline__ u32 zk_wire_code(u32 a, u32 b, const u32* aff, const ZkCtx& cx) {
  if ((a >> 28) != ZK_D_AFF) return zk_desc_decode(a, b, cx);
  const u32* e = aff + 4ull * b;
  return zk_aff_apply(e[2], zk_desc_decode(e[0], e[1], cx), cx);
}
