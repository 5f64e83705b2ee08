// Witness schedule: the compact "kept-v1" layout as segment offsets.
// Built once per circuit on the host (zkwg_sched.cpp) and passed by value to the
// kernels.  All `s_*` members are witness slot indices (field-element units)
// relative to the start of one email's witness.
#pragma once
#include <stdint.h>
#include "zkwg_fr.h"

// Kept signals of one circomlib Sha256compression instance, in layout order:
//   sigmaPlus[48] x (sigma1.xor3.out32, .mid32, sigma0.xor3.out32, .mid32, sum.out34) = 48 x 162
//   t1[64]        x (ch.out32, bigsigma1.xor3.out32, .mid32, sum.out35)               = 64 x 131
//   t2[64]        x (bigsigma0.xor3.out32, .mid32, maj.out32, .mid32, sum.out33)      = 64 x 161
//   suma[64] x 33, sume[64] x 33, fsum[8] x 33
#define ZK_COMP_SLOTS 30952u
#define ZK_SP_SLOTS 162u
#define ZK_T1_SLOTS 131u
#define ZK_T2_SLOTS 161u
#define ZK_SEC_SP_END (48u * ZK_SP_SLOTS)                   // 7776
#define ZK_SEC_T1_END (ZK_SEC_SP_END + 64u * ZK_T1_SLOTS)   // 16160
#define ZK_SEC_T2_END (ZK_SEC_T1_END + 64u * ZK_T2_SLOTS)   // 26464
// trace groups (one u64 per 32..35-bit group)
#define ZK_G_SP 0u
#define ZK_G_T1 240u
#define ZK_G_T2 496u
#define ZK_G_SUMA 816u
#define ZK_G_SUME 880u
#define ZK_G_FSUM 944u
#define ZK_TRACE_GROUPS 952u

struct ZkShaFrame {      // one Sha256Bytes / Sha256BytesPartial instance
  u32 max_bytes;         // maxByteLength
  u32 nblocks;           // maxBits / 512
  u32 lenbits;           // log2Ceil(maxBits)
  u32 partial;           // 1 = Sha256BytesPartial (preHash input)
  u32 in_data;           // byte offset of the data bytes in an input record
  u32 in_len;            // byte offset of the u32 length
  u32 in_pre;            // byte offset of precomputedSHA[32] (partial only)
  u32 hstate_base;       // first chaining-state index (units of 8 x u32) in scratch
  u64 s_inBlockIndex;    // 1 slot
  u64 s_lenbits;         // lenbits+1 slots (LessEqThan -> LessThan -> Num2Bits)
  u64 s_comp;            // nblocks x ZK_COMP_SLOTS
  u64 s_sel;             // 256 x (nblocks nums + nblocks x (isz.out, isz.inv))
  u64 s_bytes;           // max_bytes x 8
  u64 s_states;          // 32 x 8 (partial only)
};

struct ZkSched {
  u32 main_kind;
  u32 n, k;
  u32 ignore_body;
  u32 nframes;           // SHA frames (0: header, 1: body)
  u32 total_blocks;      // sum of nblocks
  u32 hstates_per_email; // sum of (nblocks+1)
  u32 in_stride;         // bytes per input record
  u32 in_off[9];         // enum zkwg_input_field -> byte offset
  u32 n_public;
  u64 W;                 // witness length in field elements
  u64 inv_table_len;     // entries in the small-inverse table
  ZkShaFrame fr[2];
  // main-component I/O slots
  u64 s_out;             // first output slot (always 1)
  u64 s_pub_in;          // first public input slot
  u64 s_prv_in;          // first private input slot
};
