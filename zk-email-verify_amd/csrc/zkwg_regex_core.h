// BodyHashRegex scan (interface level).
//
// [EXT] @zk-email/zk-regex-circom 2.3.2 `body_hash_regex.circom` is instantiated at
// packages/circuits/email-verifier.circom:126 with the regex
//     (\r\n|^)dkim-signature:([a-z]+=[^;]+; )+bh=[a-zA-Z0-9+/=]+;
// (zk-regex feeds byte 255 in front of the message to stand for `^`).  This function
// runs the equivalent NFA as a bit-set over the header and produces the circuit's
// interface signals: the match flag `out` and reveal0[i] = msg[i] inside the public
// part (the bh value) of every match, else 0.
#pragma once
#include "zkwg_fr.h"

ZK_HD bool zk_is_b64(u32 c) {
  return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '+' || c == '/' || c == '=';
}

// msg[0..n) -> rev[0..n) (u32 each); returns the number of matches.
ZK_HD u32 zk_bh_regex_scan(const u8* msg, u32 n, u32* rev) {
  const char* pat = "dkim-signature:";  // 15 chars; bit (2+k) = k chars matched
  enum { B_START = 0, B_CR = 1, B_LS = 2, B_NAME = 18, B_EQ = 19, B_VAL = 20, B_SC = 21, B_SP = 22,
         B_B1 = 23, B_B2 = 24, B_B3 = 25, B_B4 = 26, B_ACC = 27 };
  u32 st = 1u << B_START;
  u32 bh_start = 0, matches = 0;
  for (u32 pos = 0; pos <= n; ++pos) {  // pos 0 is the 255 start marker, msg[i] at pos i+1
    const u32 c = pos == 0 ? 255u : msg[pos - 1];
    const bool az = (c >= 'a' && c <= 'z');
    u32 nx = 1u << B_START;
    if (c == '\r') nx |= 1u << B_CR;
    if (((st >> B_CR) & 1u) && c == '\n') nx |= 1u << B_LS;
    if (c == 255u) nx |= 1u << B_LS;
    for (u32 k = 1; k <= 15; ++k)
      if (((st >> (2 + k - 1)) & 1u) && c == (u32)(u8)pat[k - 1]) nx |= 1u << (2 + k);
    if (az && (((st >> 17) | (st >> B_SP) | (st >> B_NAME)) & 1u)) nx |= 1u << B_NAME;
    if (((st >> B_NAME) & 1u) && c == '=') nx |= 1u << B_EQ;
    if ((((st >> B_EQ) | (st >> B_VAL)) & 1u) && c != ';') nx |= 1u << B_VAL;
    if (((st >> B_VAL) & 1u) && c == ';') nx |= 1u << B_SC;
    if (((st >> B_SC) & 1u) && c == ' ') nx |= 1u << B_SP;
    if (((st >> B_SP) & 1u) && c == 'b') nx |= 1u << B_B1;
    if (((st >> B_B1) & 1u) && c == 'h') nx |= 1u << B_B2;
    if (((st >> B_B2) & 1u) && c == '=') { nx |= 1u << B_B3; bh_start = pos + 1; }
    if ((((st >> B_B3) | (st >> B_B4)) & 1u) && zk_is_b64(c)) nx |= 1u << B_B4;
    if (((st >> B_B4) & 1u) && c == ';') {
      ++matches;
      for (u32 q = bh_start; q < pos; ++q) rev[q - 1] = msg[q - 1];
    }
    st = nx;
  }
  return matches;
}
