// Gate-list evaluation of a loaded regex template (zkwg_circom.h): one gate record.
//
// Shared by the device kernel zk_net_eval and the host mirror used by the CPU tests.  Values are small
// signed integers; a kept signal is stored in the image as a 31-bit two's-complement word, the inverse
// hint of IsZero as the integer it inverts with bit 31 set (zk_expand's ZSEG_NETP looks the inverse up).
// Every operand is a word of the evaluator's LDS image [gate values | message bytes | 0 | scratch]; the
// loader assigned the value words by liveness, the image in global memory is write-only.
//
// Record (16 words):  [0] op | k << 4 | wide << 9        [1] destination slot (output index for ZKN_OUT)
//                     [2] aux (ZKN_NEZ factor)           [3] LDS word of the value (the scratch word if nobody reads it)
//                     [4..6] constants c0A, c0B, c0C     [7] 0
//                     [8..15] operand terms: narrow = 8 x (LDS byte offset | coefficient << 18), wide = 4 x (LDS word, coefficient)
// Term roles:  narrow 0,1 -> A   2,3 -> B   4..7 -> C        wide 0 -> A   1 -> B   2,3 -> C
//   ZKN_QUAD / ZKN_ASSERT:  A * B + C  (A = c0A + its terms ...)          every other op:  L = c0A + all terms
//   ZKN_NEZ: aux * (L != 0) + c0C     ZKN_BIT: (L >> k) & 1     ZKN_INV0: the inverse of L (0 for 0)
// Unused terms point at the zero word with coefficient 0.
#pragma once
#include "zkwg_fr.h"

#define ZKN_VAL_INVERSE 0x80000000u
enum ZkNetOp : u32 { ZKN_NOP = 0, ZKN_QUAD = 1, ZKN_INV0 = 2, ZKN_BIT = 3, ZKN_NEZ = 4, ZKN_LIN = 5, ZKN_ASSERT = 6, ZKN_OUT = 7 };

ZK_HD int zk_net_decode(u32 w) { return (int)(w << 1) >> 1; }   // 31-bit two's complement

// A byte-local kept signal (a function of one message byte: the comparators of the regex circuit; zkwg_circom.h localize) is not a
// gate of the list, and neither is a kept signal of a collapsed recurrence (zkwg_circom.h chain_pass): a function of the chain state
// entering its position (zk_net_scan) and the position's symbol -- its byte, and for the backward chain also the forward state there.
// Both are served from tables; how zk_expand reaches them: ZkNetDec below.
struct ZkNetChains {
  u32 n_in;
  u32 f_end, f_smax, f_mw;            // forward chain: bytes [0, f_end)
  u32 b_end, b_smax, b_mw, b_fdim;    // backward chain: bytes [n_in - b_end, n_in); b_fdim = f_smax (or 1)
  const u8 *f_cls, *f_delta, *b_cls, *b_delta;
  const u32 *f_mask, *b_mask;
};
ZK_HD u32 zk_net_fwd_row(const ZkNetChains& K, u32 pos, const u8* fstate, const u8* msg) {
  return ((u32)K.f_cls[pos] * K.f_smax + fstate[pos]) * 256u + msg[pos];
}
ZK_HD u32 zk_net_bwd_row(const ZkNetChains& K, u32 pos, const u8* fstate, const u8* bstate, const u8* msg) {
  const u32 f = pos < K.f_end ? (u32)fstate[pos] : 0u;
  return (((u32)K.b_cls[pos] * K.b_smax + bstate[pos]) * K.b_fdim + f) * 256u + msg[pos];
}
// The bodies of zk_net_scan and zk_net_eval's prologue (zkwg_kernels_net.hip), shared with the host mirror of
// the CPU tests (tests/native/hosttest.cpp runs exactly this code).
//
// zk_net_scan: the state entering every position, one byte each, packed four to a word
// (fstate bytes [0 .. f_end], bstate bytes [n_in - b_end .. n_in - 1]).
ZK_HD void zk_net_scan_email(const ZkNetChains& K, const u8* msg, u32* fout, u32* bout) {
  u32 st = 0;
  for (u32 i = 0; i <= K.f_end && K.f_end; i += 4) {
    u32 b[4], packed = 0;
    for (u32 k = 0; k < 4; ++k) b[k] = i + k < K.f_end ? (u32)msg[i + k] : 0u;
    for (u32 k = 0; k < 4; ++k) {
      packed |= st << (8u * k);
      if (i + k < K.f_end) st = K.f_delta[((u32)K.f_cls[i + k] * K.f_smax + st) * 256u + b[k]];
    }
    fout[i >> 2] = packed;
  }
  if (!K.b_end) return;
  const u8* fstate = (const u8*)fout;     // (this lane's own stores: program order)
  const u32 lo = K.n_in - K.b_end;
  u32 packed = 0;
  st = 0;
  for (u32 t = 0; t < K.b_end; ++t) {
    const u32 p = K.n_in - 1u - t;
    packed |= st << (8u * (p & 3u));
    const u32 f = p < K.f_end ? (u32)fstate[p] : 0u;
    st = K.b_delta[(((u32)K.b_cls[p] * K.b_smax + st) * K.b_fdim + f) * 256u + msg[p]];
    if ((p & 3u) == 0u || p == lo) { bout[p >> 2] = packed; packed = 0; }
  }
}
// zk_net_eval's prologue: the mask words of position i (mw[0 .. MW + f_mw + b_mw)): the byte-local frontier bits, then the chains'
ZK_HD void zk_net_mask_words(const ZkNetChains& K, u32 MW, const u32* mask_tab, u32 i, const u8* msg, const u8* fstate, const u8* bstate, int* mw) {
  const u32 b = msg[i];
  for (u32 m = 0; m < MW; ++m) mw[m] = (int)mask_tab[b * MW + m];
  if (K.f_mw) {
    const bool in = i < K.f_end;
    const u32 row = in ? zk_net_fwd_row(K, i, fstate, msg) * K.f_mw : 0u;
    for (u32 m = 0; m < K.f_mw; ++m) mw[MW + m] = in ? (int)K.f_mask[row + m] : 0;
  }
  if (K.b_mw) {
    const bool in = i + K.b_end >= K.n_in;
    const u32 row = in ? zk_net_bwd_row(K, i, fstate, bstate, msg) * K.b_mw : 0u;
    for (u32 m = 0; m < K.b_mw; ++m) mw[MW + K.f_mw + m] = in ? (int)K.b_mask[row + m] : 0;
  }
}
// ---- the region as zk_expand reads it (zkwg_circom.h finish_region; ZSEG_NETP, zkwg_expand_dec.h ZkDecNetP) -------------------
// zk_net_eval's prologue leaves one POSITION WORD per message byte in the image: byte | fstate << 8 | bstate << 16 (fstate = 0 beyond
// the forward chain).  A table-served slot -- byte-local, forward or backward chain -- is then one load of its run's period descriptor,
// one of the position word and one of a transposed table row; nothing writes such a slot into the image any more (zk_net_fill, rounds
// 3-4, wrote 4 bytes per slot and email that zk_expand read back).
struct ZkNetDec {
  const u32* pd;        // 2 words per descriptor: [type << 30 | column] [position of period 0]
  const u32* tab;       // transposed tables L | F | B
  u32 offF, offB, nL, nF, nB, b_fdim;
  u32 m_net, m_net_pw;  // image (small) offsets: evaluated words, position words
  u32 n_in;             // message bytes (= position words)
};
enum ZkNetPType : u32 { ZKNP_EVAL = 0, ZKNP_LOCAL = 1, ZKNP_FWD = 2, ZKNP_BWD = 3 };
ZK_HD u32 zk_net_pos_word(const ZkNetChains& K, u32 pos, const u8* msg, const u8* fstate, const u8* bstate) {
  const u32 f = (K.f_end && pos < K.f_end) ? (u32)fstate[pos] : 0u;
  const u32 b = (K.b_end && pos + K.b_end >= K.n_in) ? (u32)bstate[pos] : 0u;
  return (u32)msg[pos] | (f << 8) | (b << 16);
}
// table address of a table-served slot (type != ZKNP_EVAL) from its descriptor word and the position word
ZK_HD u32 zk_netp_addr(const ZkNetDec& D, u32 d0, u32 pw) {
  const u32 t = d0 >> 30, col = d0 & 0x3fffffffu;
  const u32 byte = pw & 255u, fs = (pw >> 8) & 255u, bs = (pw >> 16) & 255u;
  const u32 row = t == ZKNP_LOCAL ? byte : (t == ZKNP_FWD ? fs * 256u + byte : (bs * D.b_fdim + fs) * 256u + byte);
  const u32 n = t == ZKNP_LOCAL ? D.nL : (t == ZKNP_FWD ? D.nF : D.nB);
  const u32 base = t == ZKNP_LOCAL ? 0u : (t == ZKNP_FWD ? D.offF : D.offB);
  return base + row * n + col;
}
// element (period i, slot q) of a run with a dense table at tab[dense ...] (zkwg_circom.h finish_region): rows = (forward state, byte)
ZK_HD u32 zk_netq_word(const ZkNetDec& D, u32 dense, u32 period, u32 pos0, u32 i, u32 q, const u32* small) {
  const u32 pw = small[D.m_net_pw + pos0 + i];
  return D.tab[dense + (((pw >> 8) & 255u) * 256u + (pw & 255u)) * period + q];
}
// stored word of region slot `slot` = element (period i, descriptor (d0, d1)) of a run whose descriptors are relative to position
// pos0; small = the email's image
ZK_HD u32 zk_netp_word(const ZkNetDec& D, u32 d0, u32 d1, u32 pos0, u32 i, u32 slot, const u32* small) {
  if ((d0 >> 30) == ZKNP_EVAL) return small[D.m_net + slot];
  return D.tab[zk_netp_addr(D, d0, small[D.m_net_pw + pos0 + i + d1])];
}
ZK_HD bool zk_net_desc_is_chain(u32 d) { return (d >> 30) == 3u; }   // (either chain)

// General path, exact in 64 bits: every record type.  `lds_r` / `lds`: the evaluator's LDS image for reads /
// writes (the same memory on the device, where the 64 lanes of a step read before any of them writes; the
// sequential host mirror reads a snapshot taken at the start of the step); `img`: the region in the email's image.
// Returns false when an assertion of the template fails or a value leaves its range.
ZK_HD bool zk_net_record(const u32* r, const int* lds_r, int* lds, u32* img, u32* out_match, u32* out_reveal, long long inv_limit) {
  const u32 op = r[0] & 15u;
  const bool wide = (r[0] >> 9) & 1u;
  long long p[8];
#pragma unroll
  for (u32 t = 0; t < 8; ++t) {
    u32 idx;
    int cf;
    if (wide) { idx = t < 4 ? r[8 + 2 * t] : 0u; cf = t < 4 ? (int)r[9 + 2 * t] : 0; }
    else { idx = (r[8 + t] & 0x3ffffu) >> 2; cf = (int)r[8 + t] >> 18; }
    p[t] = (long long)cf * lds_r[idx];
  }
  long long a, b, c;
  if (wide) { a = p[0]; b = p[1]; c = p[2] + p[3]; }
  else { a = p[0] + p[1]; b = p[2] + p[3]; c = (p[4] + p[5]) + (p[6] + p[7]); }
  const long long L = (int)r[4] + a + b + c;
  a += (int)r[4]; b += (int)r[5]; c += (int)r[6];
  const u32 dst = r[1];
  long long v = L;
  bool ok = true;
  if (op == ZKN_QUAD || op == ZKN_ASSERT) {
    ok = a > -(1ll << 31) && a < (1ll << 31) && b > -(1ll << 31) && b < (1ll << 31);
    v = a * b + c;
  } else if (op == ZKN_NEZ) v = (L != 0 ? (long long)(int)r[2] : 0ll) + (int)r[6];
  else if (op == ZKN_BIT) { ok = L >= 0; v = (L >> ((r[0] >> 4) & 31u)) & 1; }   // a negative operand is a 254-bit field element: Num2Bits rejects it
  if (op == ZKN_NOP) return true;
  if (op == ZKN_ASSERT) return ok && v == 0;
  if (op == ZKN_OUT) {
    if (dst == 0) *out_match = (u32)L; else out_reveal[dst - 1] = (u32)L;
    return true;
  }
  if (op == ZKN_INV0) {
    img[dst] = L == 0 ? 0u : (ZKN_VAL_INVERSE | ((u32)L & 0x7fffffffu));
    return L >= -inv_limit && L <= inv_limit;   // the inverse table of zk_expand covers [-inv_half, inv_half]
  }
  ok = ok && v > -(1ll << 30) && v < (1ll << 30);
  img[dst] = (u32)v & 0x7fffffffu;
  lds[r[3]] = (int)v;
  return ok;
}

// 32-bit path for the steps whose records the loader proved exact in 32 bits (value intervals): narrow terms,
// ops QUAD / NEZ / BIT / LIN / INV0, no range checks left to make; all candidate values are computed and
// selected, so the lanes of a step do not diverge.  Same arithmetic as zk_net_record.
// (products of 24-bit operands: the loader bounds coefficients by 2^13, operand values by 2^20 and the factors
// of a product by 2^23.)  One straight-line body for every op -- v = X * Y + Z with the three picked by selects --
// so that the compiler emits no branch: a branch around the stores made its s_waitcnt bookkeeping fall back to
// vmcnt(0) in every step, which serialised the record prefetch of zk_net_eval.
ZK_HD int zk_mul24(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __mul24(a, b);
#else
  return a * b;
#endif
}
ZK_HD void zk_net_record32(const u32* r, const int* lds_r, int* lds, u32* img) {
  const u32 op = r[0] & 15u;
  int p[8];
#pragma unroll
  for (u32 t = 0; t < 8; ++t) p[t] = zk_mul24((int)r[8 + t] >> 18, *(const int*)((const char*)lds_r + (r[8 + t] & 0x3ffffu)));
  const int sa = p[0] + p[1], sb = p[2] + p[3], sc = (p[4] + p[5]) + (p[6] + p[7]);
  const int L = (int)r[4] + sa + sb + sc;
  const bool quad = op == ZKN_QUAD, nez = op == ZKN_NEZ, bit = op == ZKN_BIT;
  const int X = quad ? sa + (int)r[4] : (nez ? (int)(L != 0) : (bit ? (L >> ((r[0] >> 4) & 31u)) & 1 : L));
  const int Y = quad ? sb + (int)r[5] : (nez ? (int)r[2] : 1);
  const int Z = quad ? sc + (int)r[6] : (nez ? (int)r[6] : 0);
  const int v = zk_mul24(X, Y) + Z;
  const u32 inv = L == 0 ? 0u : (ZKN_VAL_INVERSE | ((u32)L & 0x7fffffffu));
  img[r[1]] = op == ZKN_INV0 ? inv : ((u32)v & 0x7fffffffu);
  lds[r[3]] = v;   // (the scratch word for ZKN_INV0 and for values nobody reads)
}
