// Slot decoders of zk_expand: how one witness slot derives from the compact image / input record.
//
// A decoder maps (segment, logical index r) to a 32-bit CODE:
//   bit 31 clear : the slot's value is the code itself (a non-negative integer below 2^31) -- > 95 % of the
//                  EmailVerifier witness (single bits, bytes, counters);
//   bit 31 set   : a reference, type in bits 30..28, payload in bits 27..0 -- the 32 bytes are fetched by the
//                  store loop (genuine field elements, table inverses, 128-bit limbs ...).
// zk_expand runs the decoders into LDS (4 bytes per slot) and then streams the portion out with a
// fill-shaped store loop; the numbered-circuit (`--O0`) expansion calls the same decoders per wire through
// a descriptor table (zkwg_o0.h).  Segment semantics: zkwg_sched.h (enum ZkSegType).
#pragma once
#include "zkwg_dev.h"
#include "zkwg_bh_dfa.h"
#include "zkwg_net_core.h"

// the decoders also run on the host (zkwg_expand_host: witnesses expanded in host memory from a downloaded image)
#define ZK_DEC __host__ __device__

#define ZK_REF 0x80000000u
#define ZK_REF_FRV (ZK_REF | (0u << 28))   // frv[payload]            (32 bytes)
#define ZK_REF_INV (ZK_REF | (1u << 28))   // invtab[payload]         (d^-1, payload = d + inv_half)
#define ZK_REF_LIMB (ZK_REF | (2u << 28))  // 16-byte limb at rec + payload, high half zero
#define ZK_REF_RAW (ZK_REF | (3u << 28))   // small[payload] as a raw 32-bit value (bit 31 may be set)
#define ZK_REF_NEG (ZK_REF | (4u << 28))   // small[payload] is a signed word w, d = (i32)(w << 1) >> 1 < 0: the slot is r + d
#define ZK_REF_I64 (ZK_REF | (5u << 28))   // (small[payload], small[payload + 1]) is a signed 64-bit integer v: the slot is v mod r
#define ZK_REF_MINUS (ZK_REF | (6u << 28)) // no load: the slot is r - payload, 0 < payload < 2^28 (small negative results: the -1 / -2 of bit constraints)
#define ZK_REF_TYPE(code) (((code) >> 28) & 7u)
#define ZK_REF_PAYLOAD(code) ((code) & 0x0fffffffu)

struct ZkCtx {
  const u8* __restrict__ rec;
  const u64* __restrict__ bits;
  const u32* __restrict__ small;
  int half;           // inverse table covers [-half, half]
  u32 m_dfa_cm, m_dfa_pm, m_dfa_st;   // small[] offsets of the DFA class masks, primitive masks and per-position words
  const ZkNetDec* nd;                 // loaded regex template (ZSEG_NETP); NULL otherwise
};

ZK_DEC __forceinline__ u32 zk_udiv(u32 r, u32 d, u32 magic) { return magic ? (u32)(((u64)r * magic) >> 32) : r / d; }
ZK_DEC __forceinline__ u32 zk_inv_code(int d, int half) { const int c = d < -half ? -half : (d > half ? half : d); return ZK_REF_INV | (u32)(c + half); }
ZK_DEC __forceinline__ u32 zk_raw_code(u32 v, u32 idx) { return (v >> 31) ? (ZK_REF_RAW | idx) : v; }
ZK_DEC __forceinline__ u32 zk_minu(u32 a, u32 b) { return a < b ? a : b; }

struct ZkDecSmall {
  const u32* __restrict__ p; u32 base;
  ZK_DEC ZkDecSmall(const ZkSeg& sg, const ZkCtx& cx) : p(cx.small), base(sg.src) {}
  ZK_DEC u32 operator()(u32 r) const { return zk_raw_code(p[base + r], base + r); }
};
struct ZkDecFr {
  u32 base;
  ZK_DEC ZkDecFr(const ZkSeg& sg, const ZkCtx&) : base(sg.src) {}
  ZK_DEC u32 operator()(u32 r) const { return ZK_REF_FRV | (base + r); }
};
struct ZkDecBits {
  const u64* __restrict__ p; u32 a, b, magic;
  ZK_DEC ZkDecBits(const ZkSeg& sg, const ZkCtx& cx) : p(cx.bits + sg.src), a(sg.a), b(sg.b), magic(sg.pad) {}
  ZK_DEC u32 operator()(u32 r) const {
    const u32 g = zk_udiv(r, a, magic), bit = r - g * a;
    return (u32)(p[g * b + (bit >> 6)] >> (bit & 63)) & 1u;
  }
};
// Sha256compression periods: PER slots over WORDS words, the last word takes the tail
template <u32 PER, u32 WORDS>
struct ZkDecSha {
  const u64* __restrict__ p;
  ZK_DEC ZkDecSha(const ZkSeg& sg, const ZkCtx& cx) : p(cx.bits + sg.src) {}
  ZK_DEC u32 operator()(u32 r) const {
    const u32 i = r / PER, q = r - i * PER;
    const u32 sub = zk_minu(q >> 5, WORDS - 1u);
    return (u32)(p[i * WORDS + sub] >> (q - sub * 32u)) & 1u;
  }
};
struct ZkDecIsz {
  const u32* __restrict__ p; int half;
  ZK_DEC ZkDecIsz(const ZkSeg& sg, const ZkCtx& cx) : p(cx.small + sg.src), half(cx.half) {}
  ZK_DEC u32 operator()(u32 r) const {
    const int d = (int)p[r >> 1];
    return (r & 1u) ? zk_inv_code(d, half) : (u32)(d == 0);
  }
};
struct ZkDecSel {
  // 256 x ItemAtIndex(NB): per output bit k: nums[NB], then NB x (isz.out, isz.inv)
  const u32* __restrict__ dig; u32 NB, per, magic; int idx, half;
  ZK_DEC ZkDecSel(const ZkSeg& sg, const ZkCtx& cx)
      : dig(cx.small + sg.b), NB(sg.a), per(3u * sg.a), magic(sg.pad), idx((int)cx.small[sg.src]), half(cx.half) {}
  ZK_DEC u32 operator()(u32 r) const {
    const u32 k = zk_udiv(r, per, magic), q = r - k * per;
    if (q < NB) return ((int)q == idx) ? ((dig[k >> 5] >> (31u - (k & 31u))) & 1u) : 0u;
    const u32 t = q - NB, j = t >> 1;
    return (t & 1u) ? zk_inv_code(idx - (int)j, half) : (u32)((int)j == idx);
  }
};
struct ZkDecIn8 {
  const u8* __restrict__ p;
  ZK_DEC ZkDecIn8(const ZkSeg& sg, const ZkCtx& cx) : p(cx.rec + sg.src) {}
  ZK_DEC u32 operator()(u32 r) const { return p[r]; }
};
struct ZkDecIn8Mask {
  const u8* __restrict__ p; const u8* __restrict__ m;
  ZK_DEC ZkDecIn8Mask(const ZkSeg& sg, const ZkCtx& cx) : p(cx.rec + sg.src), m(cx.rec + sg.a) {}
  ZK_DEC u32 operator()(u32 r) const { return (u32)p[r] * (u32)m[r]; }
};
struct ZkDecIn8Bits {
  const u8* __restrict__ p;
  ZK_DEC ZkDecIn8Bits(const ZkSeg& sg, const ZkCtx& cx) : p(cx.rec + sg.src) {}
  ZK_DEC u32 operator()(u32 r) const { return (u32)(p[r >> 3] >> (r & 7u)) & 1u; }
};
struct ZkDecLimb {
  u32 base;
  ZK_DEC ZkDecLimb(const ZkSeg& sg, const ZkCtx&) : base(sg.src) {}
  ZK_DEC u32 operator()(u32 r) const { return ZK_REF_LIMB | (base + 16u * r); }
};
struct ZkDecLtBits {
  long long base; u32 per, magic;
  ZK_DEC ZkDecLtBits(const ZkSeg& sg, const ZkCtx& cx)
      : base((long long)(int)cx.small[sg.src] + (1ll << sg.a)), per(sg.a + 1u), magic(sg.pad) {}
  ZK_DEC u32 operator()(u32 r) const {
    const u32 i = zk_udiv(r, per, magic), bit = r - i * per;
    return (u32)((u64)(base - (long long)i) >> bit) & 1u;
  }
};
struct ZkDecRegSel {
  // SelectRegexReveal (utils/regex.circom:31-37): per index i: IsEqual(i,start) (out,inv),
  // IsZero(in[i]) (out,inv), [i>0: IsZero(in[i-1]) (out,inv)], GreaterThan(bl)(i, start+43) bits
  const u32* __restrict__ rev; u32 bl, per, N, magic; int start, half;
  ZK_DEC ZkDecRegSel(const ZkSeg& sg, const ZkCtx& cx)
      : rev(cx.small + sg.b), bl(sg.a), per(6u + sg.a + 1u), N(sg.c), magic(sg.pad), start((int)cx.small[sg.src]), half(cx.half) {}
  ZK_DEC u32 operator()(u32 r) const {
    u32 i, q;
    if (r < per - 2u) {  // index 0 has no "previous" IsZero
      i = 0; q = r < 4u ? r : r + 2u;
    } else {
      const u32 rr = r - (per - 2u), t = zk_udiv(rr, per, magic);
      i = 1u + t; q = rr - t * per;
    }
    if (q < 6u) {
      int d;
      if (q < 2u) d = start - (int)i;                       // isz.in = in[1] - in[0] = startIndex - i
      else if (q < 4u) d = (int)rev[i < N ? i : N - 1u];
      else d = (int)rev[i ? i - 1u : 0u];
      return (q & 1u) ? zk_inv_code(d, half) : (u32)(d == 0);
    }
    const long long val = (long long)start + 43 + (1ll << bl) - (long long)i;
    return (u32)((u64)val >> (q - 6u)) & 1u;
  }
};
struct ZkDecVShift {
  const u32* __restrict__ small; u32 N, b, shift, magic;
  ZK_DEC ZkDecVShift(const ZkSeg& sg, const ZkCtx& cx) : small(cx.small), N(sg.a), b(sg.b), shift(cx.small[sg.src]), magic(sg.pad) {}
  ZK_DEC u32 operator()(u32 r) const {
    const u32 j = zk_udiv(r, N, magic), i = r - j * N;
    const u32 sh = shift & ((2u << j) - 1u);
    const u32 idx = b + (i + sh) % N;
    return zk_raw_code(small[idx], idx);
  }
};
template <bool FULL>   // FULL: Base64Lookup internals (68 slots per char), else the 6 value bits
struct ZkDecB64 {
  const u32* __restrict__ chars; int half;
  ZK_DEC ZkDecB64(const ZkSeg& sg, const ZkCtx& cx) : chars(cx.small + sg.src), half(cx.half) {}
  ZK_DEC u32 operator()(u32 r) const {
    constexpr u32 per = FULL ? 68u : 6u;
    const u32 g = r / per, q = r - g * per;
    const int ch = (int)chars[g];
    // lib/base64.circom:71-128
    const u32 rAZ = (ch >= 65 && ch <= 90), raz = (ch >= 97 && ch <= 122), r09 = (ch >= 48 && ch <= 57);
    const u32 sAZ = rAZ * (u32)(ch - 65);
    const u32 saz = sAZ + raz * (u32)(ch - 71);
    const u32 s09 = saz + r09 * (u32)(ch + 4);
    const u32 spl = s09 + (ch == 43) * (u32)(ch + 19);
    const u32 ssl = spl + (ch == 47) * (u32)(ch + 16);
    if (!FULL) return (ssl >> q) & 1u;
    if (q < 8u) {
      const u32 mids[8] = {rAZ, sAZ, raz, saz, r09, s09, spl, ssl};
      return mids[q];
    }
    if (q < 62u) {
      const u32 k = (q - 8u) / 9u, bit = (q - 8u) - k * 9u;
      // le_Z: in+256-91, ge_A: 64+256-in, le_z: in+256-123, ge_a: 96+256-in, le_9: in+256-58, ge_0: 47+256-in
      const int vals[6] = {ch + 256 - 91, 64 + 256 - ch, ch + 256 - 123, 96 + 256 - ch, ch + 256 - 58, 47 + 256 - ch};
      return ((u32)vals[k] >> bit) & 1u;
    }
    const u32 k = (q - 62u) >> 1;
    const int d = ch - (k == 0 ? 43 : (k == 1 ? 47 : 61));
    return ((q - 62u) & 1u) ? zk_inv_code(d, half) : (u32)(d == 0);
  }
};
// BodyHashRegex DFA circuit arrays (zkwg_layout.h zk_walk_bh_regex): one entry per position i of
// in[] = [255, header...].  zk_misc_ev left one word per position: in | st<<8 | nx<<16 | st_next<<24
// (nx = the transition out of a non-zero state, 255 = none) plus the class / primitive truth masks.
// value of a DFA-array element (kind, sub-slot q) from the position word w0 (of position i, i + 1 for ZDFA_SUB) and
// the position's mask word mw (primitive masks for ZDFA_CLS, class masks for ZDFA_AND; unused otherwise)
ZK_DEC __forceinline__ u32 zk_dfa_value_w(u32 kind, u32 q, u32 pb, u32 pc, u32 w0, u32 mw, int half) {
  const u32 b = w0 & 255u, st = (w0 >> 8) & 255u, nx = (w0 >> 16) & 255u, sn = w0 >> 24;
  switch (kind) {
    case ZDFA_EQ: {
      const int d = (int)pb - (int)b;          // isz.in = in[1] - in[0] = ch - in[i]
      return q ? zk_inv_code(d, half) : (u32)(d == 0);
    }
    case ZDFA_LT: return ((pc ? pb + b : pb - b) >> q) & 1u;
    case ZDFA_RNG: return (u32)(b >= pb && b <= pc);
    case ZDFA_CLS: return (u32)(q == 0) ^ (u32)(__builtin_popcount(mw & pc) != 0);
    case ZDFA_AND: return (u32)(pb ? (st == pb) : (nx == 255u)) & ((mw >> pc) & 1u);
    case ZDFA_TMP: return (u32)(q == 0) ^ (u32)(nx == pb);
    case ZDFA_FZE: return (u32)(q == 0) ^ (u32)(nx != 255u);
    case ZDFA_ST: return (u32)(q == 0) ^ (u32)(sn == pb);
    case ZDFA_SUB: {
      // message index i: transition st[i+1] -> st[i+2] = (st, sn) of word i+1
      const u32 key = st | (sn << 8);
      bool hit = false;
#pragma unroll
      for (u32 k = 0; k < ZK_DFA_NPUBLIC; ++k) hit = hit || key == (ZK_DFA_PUBLIC[k][0] | ((u32)ZK_DFA_PUBLIC[k][1] << 8));
      return (u32)(q == 0) ^ (u32)hit;
    }
    default: return 0u;
  }
}
ZK_DEC __forceinline__ u32 zk_dfa_value(u32 kind, u32 i, u32 q, u32 pb, u32 pc, const u32* __restrict__ pos, const u32* __restrict__ cmask,
                                        const u32* __restrict__ pmask, int half) {
  const u32 w0 = pos[i + (kind == ZDFA_SUB ? 1u : 0u)];
  const u32 mw = kind == ZDFA_CLS ? pmask[i] : (kind == ZDFA_AND ? cmask[i] : 0u);
  return zk_dfa_value_w(kind, q, pb, pc, w0, mw, half);
}
// (position, sub-slot) of element r of a DFA array of the given kind
ZK_DEC __forceinline__ void zk_dfa_index(u32 kind, u32 r, u32& i, u32& q) {
  if (kind == ZDFA_LT) { i = r / 9u; q = r - i * 9u; }
  else if (kind == ZDFA_RNG || kind == ZDFA_AND) { i = r; q = 0; }
  else { i = r >> 1; q = r & 1u; }
}
struct ZkDecDfa {
  const u32* __restrict__ pos; const u32* __restrict__ cmask; const u32* __restrict__ pmask;
  u32 kind, pb, pc; int half;
  ZK_DEC ZkDecDfa(const ZkSeg& sg, const ZkCtx& cx)
      : pos(cx.small + sg.src), cmask(cx.small + cx.m_dfa_cm), pmask(cx.small + cx.m_dfa_pm), kind(sg.a), pb(sg.b), pc(sg.c), half(cx.half) {}
  ZK_DEC u32 operator()(u32 r) const {
    u32 i, q;
    zk_dfa_index(kind, r, i, q);
    return zk_dfa_value(kind, i, q, pb, pc, pos, cmask, pmask, half);
  }
};
// RemoveSoftLineBreaks arrays derived from the emailBody bytes alone
// (helpers/remove-soft-line-breaks.circom:47-91): "=\r\n" at j  <=>  isSoftBreak[j]
struct ZkDecRslb {
  const u8* __restrict__ enc; u32 kind, b, c; int half;
  ZK_DEC ZkDecRslb(const ZkSeg& sg, const ZkCtx& cx) : enc(cx.rec + sg.src), kind(sg.a), b(sg.b), c(sg.c), half(cx.half) {}
  ZK_DEC u32 operator()(u32 r) const {
    if (kind == ZRS_EQ) {
      const int d = (int)c - (int)enc[(r >> 1) + b];   // isz.in = in[1] - in[0]
      return (r & 1u) ? zk_inv_code(d, half) : (u32)(d == 0);
    }
    if (kind == ZRS_TSB) return (u32)(enc[r] == 61u) & (u32)(enc[r + 1] == 13u);
    if (kind == ZRS_SB) return (u32)(enc[r] == 61u) & (u32)(enc[r + 1] == 13u) & (u32)(enc[r + 2] == 10u);
    // processed[r]: zero inside a soft break starting at r, r-1 or r-2 (starts only below M-2)
    bool z = false;
#pragma unroll
    for (u32 k = 0; k < 3; ++k)
      if (r >= k && r - k + 2 < b) { const u32 j = r - k; z = z || (enc[j] == 61u && enc[j + 1] == 13u && enc[j + 2] == 10u); }
    return z ? 0u : (u32)enc[r];
  }
};
// One periodic run of a loaded regex template's region (zkwg_circom.h finish_region, zkwg_net_core.h ZkNetDec).  Slot r is
// element q = r % period of period i = r / period: the run's descriptor q says where the slot's stored word is -- a column of a
// transposed table, addressed through the position word of position (descriptor's) + i, or, for the few signals the evaluator
// computes, the image.  A stored word is a 31-bit signed integer or the inverse of one (bit 31) from the table; a negative
// integer -m is the field element r - m.  Branch-free: every lane forms one address; neighbouring lanes are neighbouring slots
// of one position, i.e. neighbouring columns of one table row.
struct ZkDecNetP {
  const u32* __restrict__ small; const ZkNetDec* __restrict__ D; u32 pd0, P, magic, pos0, c; int half;
  // (the decode parameters stay behind the pointer: copied into the functor they cost the store kernel 30 scalar registers it does not
  // have -- 88 bytes of scratch per lane in every zk_expand3 kernel, loaded template or not)
  ZK_DEC ZkDecNetP(const ZkSeg& sg, const ZkCtx& cx) : small(cx.small), D(cx.nd), pd0(sg.src), P(sg.a), magic(sg.pad), pos0(sg.b), c(sg.c), half(cx.half) {}
  ZK_DEC u32 operator()(u32 r) const {
    const u32 i = zk_udiv(r, P, magic), q = r - i * P;
    // descriptor and position word are asked for together: nearly every descriptor of a run sits at the run's own position
    // (offset 0), so the dependent chain in front of the store is {descriptor, position word} -> table word
    const u32 pwi = D->m_net_pw + zk_minu(pos0 + i, D->n_in);
    const uint2 d = ((const uint2*)D->pd)[pd0 + q];
    u32 pw = small[pwi];
    if (d.y) pw = small[pwi + d.y];
    const bool tab = (d.x >> 30) != ZKNP_EVAL;
    const u32 at = D->m_net + c + r;
    const u32* __restrict__ src = tab ? D->tab + zk_netp_addr(*D, d.x, pw) : small + at;
    const u32 w = *src;
    const int v = (int)(w << 1) >> 1;
    if (w & 0x80000000u) return zk_inv_code(v, half);
    return v >= 0 ? (u32)v : (tab ? (ZK_REF_MINUS | (u32)(-v)) : (ZK_REF_NEG | at));
  }
};

// A long run of the region with a dense table (zkwg_circom.h finish_region): position word -> table word, the chain the built-in DFA
// segments have (entry -> position word -> value); consecutive lanes read consecutive words of one row.
struct ZkDecNetQ {
  const u32* __restrict__ small; const ZkNetDec* __restrict__ D; u32 dense, P, magic, pos0; int half;
  ZK_DEC ZkDecNetQ(const ZkSeg& sg, const ZkCtx& cx) : small(cx.small), D(cx.nd), dense(sg.src), P(sg.a), magic(sg.pad), pos0(sg.b), half(cx.half) {}
  ZK_DEC u32 operator()(u32 r) const {
    const u32 i = zk_udiv(r, P, magic), q = r - i * P;
    const u32 w = zk_netq_word(*D, dense, P, pos0, i, q, small);
    const int v = (int)(w << 1) >> 1;
    if (w & 0x80000000u) return zk_inv_code(v, half);
    return v >= 0 ? (u32)v : (ZK_REF_MINUS | (u32)(-v));
  }
};

// every segment type with its decoder: ZK_FOR_SEG_TYPES(X) expands X(type, Decoder) once per type
#define ZK_FOR_SEG_TYPES(X)                                                                                           \
  X(ZSEG_SMALL, ZkDecSmall) X(ZSEG_FR, ZkDecFr) X(ZSEG_BITS, ZkDecBits)                                                \
  X(ZSEG_SHA_SP, ZkDecSha<ZK_SP_SLOTS ZK_COMMA 5>) X(ZSEG_SHA_T1, ZkDecSha<ZK_T1_SLOTS ZK_COMMA 4>)                    \
  X(ZSEG_SHA_T2, ZkDecSha<ZK_T2_SLOTS ZK_COMMA 5>) X(ZSEG_ISZ, ZkDecIsz) X(ZSEG_SEL, ZkDecSel) X(ZSEG_IN8, ZkDecIn8)   \
  X(ZSEG_IN8MASK, ZkDecIn8Mask) X(ZSEG_IN8BITS, ZkDecIn8Bits) X(ZSEG_LIMB, ZkDecLimb) X(ZSEG_LTBITS, ZkDecLtBits)      \
  X(ZSEG_REGSEL, ZkDecRegSel) X(ZSEG_VSHIFT, ZkDecVShift) X(ZSEG_B64BITS, ZkDecB64<false>) X(ZSEG_B64, ZkDecB64<true>) \
  X(ZSEG_DFA, ZkDecDfa) X(ZSEG_RSLB, ZkDecRslb) X(ZSEG_NETP, ZkDecNetP) X(ZSEG_NETQ, ZkDecNetQ)
#define ZK_COMMA ,

// one slot of any segment (pieces that straddle segments, the numbered-circuit expansion and the linear-row kernels
// reach slots one by one)
ZK_DEC inline u32 zk_decode_any(const ZkSeg& sg, u32 r, const ZkCtx& cx) {
  switch (sg.type) {
#define ZK_X(T, D) case T: return D(sg, cx)(r);
    ZK_FOR_SEG_TYPES(ZK_X)
#undef ZK_X
    default: return 0u;   // ZSEG_HOLE: nothing produces these slots
  }
}
// K slots r0 + i0, r0 + i0 + 64, ... of ONE segment (slots at or beyond `n` give 0): the type switch is taken once and the
// K decodes are straight-line code, so their image reads are in flight together
template <int K>
__device__ __forceinline__ void zk_decode_k(const ZkSeg& sg, u32 r0, u32 i0, u32 n, const ZkCtx& cx, u32 (&code)[K]) {
  switch (sg.type) {
#define ZK_X(T, D) case T: { const D dec(sg, cx); _Pragma("unroll") for (int k = 0; k < K; ++k) { const u32 i = i0 + 64u * (u32)k; code[k] = i < n ? dec(r0 + i) : 0u; } break; }
    ZK_FOR_SEG_TYPES(ZK_X)
#undef ZK_X
    default:
#pragma unroll
      for (int k = 0; k < K; ++k) code[k] = 0u;
      break;
  }
}

// ---------------------------------------------------------------- store side
// the 16-byte half `hf` of the slot a reference code names
struct ZkRefSrc {
  const uint4* __restrict__ frv;     // the email's field elements (Montgomery output: their Montgomery copies)
  const uint4* __restrict__ invtab;  // inverse table (Montgomery output: the Montgomery-form copy)
  const u8* __restrict__ rec;
  const u32* __restrict__ small;
};
ZK_DEC __forceinline__ uint4 zk_ref_half(u32 code, u32 hf, const ZkRefSrc& R) {
  const u32 p = ZK_REF_PAYLOAD(code);
  switch (ZK_REF_TYPE(code)) {
    case 0: return R.frv[2u * p + hf];
    case 1: return R.invtab[2u * p + hf];
    case 2: return hf ? zk_zero4() : *(const uint4*)(R.rec + p);
    case 3: return hf ? zk_zero4() : zk_small(R.small[p]);
    case 4: {
      const u32 w = R.small[p];
      const u32 m = (u32)(-((int)(w << 1) >> 1));
      return hf ? make_uint4(0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u)
                : make_uint4(0xf0000001u - m, 0x43e1f593u, 0x79b97091u, 0x2833e848u);
    }
    case 6:
      return hf ? make_uint4(0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u)
                : make_uint4(0xf0000001u - p, 0x43e1f593u, 0x79b97091u, 0x2833e848u);
    default: {
      const long long v = (long long)((u64)R.small[p] | ((u64)R.small[p + 1] << 32));
      if (v >= 0) return hf ? zk_zero4() : make_uint4((u32)v, (u32)((u64)v >> 32), 0u, 0u);
      // r - m, m = -v < 2^63: a borrow out of the low limb stops in the next one (r's second limb is not 0)
      const u64 m = (u64)(-v), l0 = 0x43e1f593f0000001ull - m, l1 = 0x2833e84879b97091ull - (m > 0x43e1f593f0000001ull ? 1ull : 0ull);
      return hf ? make_uint4(0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u)
                : make_uint4((u32)l0, (u32)(l0 >> 32), (u32)l1, (u32)(l1 >> 32));
    }
  }
}
